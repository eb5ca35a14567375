#!/bin/bash
# round 3, GPU session 7: what clocks does k_search_fast run at?  (the run under rocprofv3 was 4.7 % faster than the plain run of the
# same session: profilers pin the performance level)  perf level auto vs high, sclk sampled while the kernel runs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/s7
O=gpurun_out/s7
rocm-smi -p -c --showpower 2>&1 | grep -v "^=\|^$" | head -30 > $O/smi_before.txt
sample() { while true; do rocm-smi -c --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.5; done; }
run() { timeout 600 python scripts/perf_search.py --n 10000000 --nq 262144 --L 3 --rescore 196 --reps 6 --configs VS_FAST=1 --graph-cache /tmp/g 2>&1 | grep -E "search |index ready"; }
echo "# perf level auto" | tee $O/clocks.txt
sample > $O/samples_auto.txt & SP=$!
run | tee -a $O/clocks.txt
kill $SP
rocm-smi --setperflevel high > $O/setperf.txt 2>&1
rocm-smi -p 2>&1 | grep -i -E "perf|level" | head -5 >> $O/setperf.txt
echo "# perf level high" | tee -a $O/clocks.txt
sample > $O/samples_high.txt & SP=$!
run | tee -a $O/clocks.txt
kill $SP
rocm-smi --setperflevel auto >> $O/setperf.txt 2>&1
echo "# perf level auto again" | tee -a $O/clocks.txt
run | tee -a $O/clocks.txt
rm -f /tmp/g.*
sort $O/samples_auto.txt | uniq -c | sort -rn | head -8 > $O/samples_auto_hist.txt
sort $O/samples_high.txt | uniq -c | sort -rn | head -8 > $O/samples_high_hist.txt
cat $O/setperf.txt; head -4 $O/samples_auto_hist.txt; head -4 $O/samples_high_hist.txt
