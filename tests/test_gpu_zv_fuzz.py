"""A fixed handful of the cases scripts/fuzz_emu.py draws (random geometry x random scan parameters x four of the library's
kernels / regimes each, `plain` storage every fifth case, the amgettuple mirror on top, page-path round trips every seventh): everything must equal the oracle.
The open-ended run is `python scripts/fuzz_emu.py --seconds N` (interpreter) or `--gpu` (device)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


@pytest.mark.parametrize("case_seed", [7000001, 7000002, 7000003, 7000004, 7000005, 7000006, 7000007, 7000008, 7000009, 7000014, 7000013, 7000020, 7000025, 7000018,
                                       777000331])  # (the last one: scans that exhaust a 900-node graph — the case that caught the epoch-tagged tables on hardware)
def test_fuzz_case(gpu_ctx, oracle, case_seed):
    import fuzz_emu
    saved = {v: os.environ.get(v) for v in fuzz_emu.TUNING}
    try:
        fuzz_emu.one_case(gpu_ctx, oracle, case_seed, verbose=False)
    finally:
        for v, x in saved.items():
            if x is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = x
