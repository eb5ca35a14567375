"""The C-ABI library loads and exports every symbol include/vsgpu.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "vsgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libvsgpu.so missing: run __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/vsgpu.h but not exported"
    # and the Python binding table covers the header exactly
    assert sorted(_lib.SYMBOLS) == declared
    P.load()


def test_no_cpu_fallback_without_device():
    """Product path must fail loudly when there is no HIP device (this container) instead of computing on the CPU."""
    import pytest
    import pgvectorscale_amd as P
    try:
        ctx = P.Context(0)
    except P.VsError as e:
        assert e.code == -2 and "no CPU fallback" in str(e)
    else:  # on a GPU box the context simply works
        ctx.close()
        pytest.skip("HIP device present")


def test_product_package_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pgvectorscale_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "vs_oracle" not in txt, f


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """include/vsgpu.h is what a Rust / C host binds: it must compile as C99 with no C++ in it, and a plain C program must link
    against libvsgpu.so and get VS_ERR_HIP (not a crash, not a CPU fallback) from the first entry point that needs a device when
    there is none — on a GPU box the same program creates a context and shards a batch with vs_shard_range."""
    import subprocess
    import pgvectorscale_amd
    hdr = os.path.join(ROOT, "include", "vsgpu.h")
    r = subprocess.run(["gcc", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "vsgpu.h"
int main(void) {
    vs_ctx* ctx = NULL;
    uint32_t b = 0, e = 0;
    if (vs_shard_range(11, 2, 1, &b, &e) != VS_OK || b != 6 || e != 11) { printf("shard arithmetic\n"); return 2; }
    if (vs_shard_range(4, 2, 2, &b, &e) != VS_ERR_INVALID) { printf("bad rank accepted\n"); return 2; }
    int rc = vs_ctx_create(0, &ctx);
    if (rc == VS_OK) { printf("device: ok\n"); vs_ctx_destroy(ctx); return 0; }
    if (rc != VS_ERR_HIP || ctx != NULL || strstr(vs_last_error(), "no CPU fallback") == NULL) { printf("unexpected: %d %s\n", rc, vs_last_error()); return 3; }
    printf("no device: VS_ERR_HIP\n");
    return 0;
}
''')
    libdir = os.path.dirname(pgvectorscale_amd._lib.LIB_PATH)
    exe = tmp_path / "host"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-l:libvsgpu.so",
                        f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "VS_ERR_HIP" in r.stdout or "device: ok" in r.stdout


def test_options_go_through_the_abi_and_the_environment_is_a_snapshot(monkeypatch):
    """vs_set_option / vs_get_option (round 6: the VS_* tuning switches of DESIGN.md section 10 are looked up in a table, not with getenv
    on the launch path): an option set through the ABI wins over the environment variable of the same name; the environment is seen
    through a snapshot that follows later changes of the process's VS_* variables; unsetting falls back; bad names are refused"""
    import pgvectorscale_amd as P
    monkeypatch.delenv("VS_TEST_OPTION_X", raising=False)
    assert P.get_option("VS_TEST_OPTION_X") is None
    monkeypatch.setenv("VS_TEST_OPTION_X", "7")
    assert P.get_option("VS_TEST_OPTION_X") == "7"       # the snapshot follows the environment ...
    monkeypatch.setenv("VS_TEST_OPTION_X", "8")
    assert P.get_option("VS_TEST_OPTION_X") == "8"
    P.set_option("VS_TEST_OPTION_X", 3)
    assert P.get_option("VS_TEST_OPTION_X") == "3"       # ... and an option set through the ABI wins
    monkeypatch.setenv("VS_TEST_OPTION_X", "9")
    assert P.get_option("VS_TEST_OPTION_X") == "3"
    P.set_option("VS_TEST_OPTION_X", None)
    assert P.get_option("VS_TEST_OPTION_X") == "9"
    monkeypatch.delenv("VS_TEST_OPTION_X")
    assert P.get_option("VS_TEST_OPTION_X") is None
    with pytest.raises(P.VsError):
        P.set_option("PATH", "x")


def test_no_getenv_on_the_launch_path():
    """the product sources read their options through vs_opt_get (vs_options.cpp) only"""
    csrc = os.path.join(ROOT, "pgvectorscale_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".cpp", ".h")) and f not in ("vs_options.cpp", "vs_shm_lat.cpp"):
            assert "getenv(" not in open(os.path.join(csrc, f)).read(), f
