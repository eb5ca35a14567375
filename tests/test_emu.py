"""Kernel SOURCE parity without a GPU: the `-m gpu` parity tests re-run in a child process against
tests/emu/libvsgpu_emu.so — the unmodified product sources (pgvectorscale_amd/csrc/*.hip) compiled for the host on top
of a wave64 lockstep interpreter (tests/emu/fake/hip/hip_runtime.h, tests/emu/emu_rt.cpp: one fiber per GPU thread,
rendezvous at every cross-lane operation).  The interpreter is test infrastructure: it checks what the kernels COMPUTE
(heap mechanics, dedup, visited list, DPP / ballot logic, accumulation order) against the oracle when no MI355X is at
hand; it says nothing about speed, registers or memory-model races, and the product never loads it.  The GPU tier
(`-m gpu`, real hardware through the real libvsgpu.so) remains the parity gate.

Default: a subset that finishes in well under a minute on 8 cores.  VS_EMU_FULL=1: the whole `-m gpu` suite (about 3 minutes on 8 cores).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")

SUBSET = ("test_hip_path_reproduces_golden or test_index_from_pages_searches_like_the_oracle or "
          "(test_every_regime_is_exact and labels_deleted) or (test_search_rows_match_oracle and (cosine_768 or tiny_L)) or "
          "(test_sbq_stream_bit_exact and (three_bits or big_R)) or (test_rerank_matches_reference_order and 100) or "
          "test_scan_topk or test_train_and_quantize_corpus_bit_exact or test_deleted_label_null_and_exhaustive or "
          "test_concurrent_scans_are_batched_and_exact or test_client_processes_share_launches or test_pages_decoded_on_the_device or (test_plain_storage_rows_match_the_oracle and 100) or "
          "(test_rows_and_stats_one_row_at_a_time and (l2_window or labels_deleted)) or test_scan_that_outgrows_its_capacities or "
          "(test_wide_keys_host_batch and 1) or test_label_masks_for_any_label_values or test_backends_with_different_snapshots or "
          "test_snapshot_masks_across_processes or test_index_from_the_relation_alone or (test_every_regime_is_exact and tableless_slotmap_tight) or "
          "(test_register_capped_variants and 6_virgin) or test_amgettuple_mirror_on_a_broker or "
          "test_backend_processes_stream_past_the_first_rows or test_two_row_gather_full_neighbor_lists or "
          "test_autotune_holds_every_variant_to_the_defaults_rows or test_a_variant_whose_rows_differ_is_disqualified or "
          "test_small_scans_try_the_table_less_regime or test_set_variant_by_name or "
          "test_a_client_that_rewrites_its_request_after_posting_cannot_move_the_dispatcher or test_replica_on_a_second_context_outlives_its_source or (test_multi_search_batch_returns_the_single_device_rows and 33) or "
          "test_comm_world_of_one_gathers_and_replicates or test_entry_points_that_move_the_arrays_refuse_while_a_view_is_alive or "
          "test_placement_never_changes_a_row or test_pooled_scans_hand_out_the_oracles_rows_and_stats or test_a_round_carries_the_listed_scans_that_are_streamed_ahead or (test_every_regime_is_exact and tableless_q16_tight) or "
          "test_handles_return_their_device_memory or test_deep_scans_stream_on_lanes or test_shm_server_with_lanes_gives_its_memory_back or test_staging_ring_round_trip or test_row_wise_staging_pads_and_copies_every_row")


@pytest.fixture(scope="module")
def emu_lib():
    if os.environ.get("VS_EMU"):
        pytest.skip("already inside the emulated run")
    r = subprocess.run(["make", "-C", EMU_DIR, "-j8", "-s"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(EMU_DIR, "libvsgpu_emu.so")


def test_gpu_parity_tests_pass_on_the_wave64_interpreter(emu_lib):
    env = dict(os.environ, VS_EMU="1")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"]
    if not os.environ.get("VS_EMU_FULL"):
        cmd += ["-k", SUBSET]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=3000)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
