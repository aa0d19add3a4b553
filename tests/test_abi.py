"""The C-ABI library loads and exports every symbol include/gemma_hip.h declares (no GPU needed);
without a GPU every compute entry point must fail loudly -- there is no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    from gemma_amd import build
    return build.build()


def test_header_symbols_exported():
    so = _built()
    hdr = open(os.path.join(ROOT, "include", "gemma_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(gemma_hip_[A-Za-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    lib = C.CDLL(so)
    for name in declared:
        assert hasattr(lib, name), name
    from gemma_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared
    assert _lib.lib().gemma_hip_abi_version() == 4


def test_struct_layouts_match_header():
    from gemma_amd import _lib
    assert C.sizeof(_lib.SumStat) == 64  # SUMSTAT: 8 doubles, src/param.h:54-66
    assert C.sizeof(_lib.LmmCfg) == 72


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked tests")
    from gemma_amd import _lib, api
    rc = _lib.lib().gemma_hip_init(-1, 0)
    assert rc == _lib.ENODEV
    A = np.ones((4, 4))
    with pytest.raises(_lib.GemmaHipError) as e:
        api.fast_dgemm("N", "N", 1.0, A, A, 0.0, np.zeros((4, 4)))
    assert e.value.code == _lib.ENODEV
    with pytest.raises(_lib.GemmaHipError):
        api.CalcKin(np.zeros((2, 4)), _lib.GENO_F64_SNP_MAJOR, 4)


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "gemma_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.replace("(oracle", "").lower() or "oracle" not in txt, (dp, f)
    for f in os.listdir(os.path.join(ROOT, "include")):  # the public headers: C ABI and the C++ host layer
        txt = open(os.path.join(ROOT, "include", f), errors="ignore").read().lower()
        assert "oracle" not in txt and "orc_" not in txt, f


def test_shard_range_covers_everything():
    from gemma_amd.dist import shard_range
    for p in (0, 1, 7, 1000, 1000001):
        for w in (1, 2, 4, 8):
            got = [shard_range(p, r, w) for r in range(w)]
            assert got[0][0] == 0 and got[-1][1] == p
            for a, b in zip(got, got[1:]):
                assert a[1] == b[0]


def test_cpp_host_mirror_compiles():
    """include/gemma_host.hpp (the C++ mirror of fast_dgemm / PlinkKin / class LMM ...) and its test driver
    build against the library with plain g++ (no GPU, no GSL)."""
    import subprocess
    import tempfile
    so = _built()
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "host_mirror_driver.cpp"),
                               "-L" + os.path.dirname(so), "-lgemma_hip", "-pthread", "-Wl,-rpath," + os.path.dirname(so),
                               "-o", os.path.join(td, "drv")])
        r = subprocess.run([os.path.join(td, "drv")], capture_output=True)
        assert r.returncode == 2  # usage error, before any GPU call


def test_file_driver_on_the_product_library_fails_loudly_without_gpu():
    """tests/cpp/gemma_file_driver.cpp linked against the PRODUCT library (not the test double): without a GPU the run
    stops at gemma_hip_init with the no-device message -- nothing is computed on the CPU; sharded ranks report it to the parent."""
    import subprocess
    import tempfile
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_workflow_files.py")
    so = _built()
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "drv")
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "gemma_file_driver.cpp"), "-L" + os.path.dirname(so),
                               "-lgemma_hip", "-Wl,-rpath," + os.path.dirname(so), "-lz", "-pthread", "-o", exe])
        P = os.path.join(ROOT, "tests", "golden", "text", "P")
        r = subprocess.run([exe, "-bfile", P, "-gk", "-outdir", td], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr and not os.path.exists(os.path.join(td, "result.cXX.txt"))
        r = subprocess.run([exe, "-bfile", P, "-k", "none", "-lmm", "1", "-gpus", "2", "-outdir", td], capture_output=True, text=True)
        assert r.returncode == 7 and r.stderr.count("no CPU fallback") == 2
