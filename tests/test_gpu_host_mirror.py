"""The C++ host mirror (include/gemma_host.hpp) end to end on the GPU box: a PLINK .bed with missing
calls and non-phenotyped individuals -> PlinkKin -> CenterMatrix -> EigenDecomp_Zeroed -> CalcUtX ->
null model -> LMM::AnalyzePlink -> WriteFiles, compared with the oracle's restatement of the same
BatchRun sequence at the precision of the .assoc.txt format (6 significant digits)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    exe = os.path.join(tmp, "host_mirror_driver")
    from gemma_amd import build
    build.build()
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_mirror_driver.cpp"),
                           "-L" + os.path.join(ROOT, "gemma_amd"), "-lgemma_hip", "-pthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "gemma_amd"), "-o", exe])
    return exe


@pytest.mark.parametrize("mode", [1, 4])
def test_cpp_host_mirror_plink_end_to_end(tmp_path, oracle, mode):
    rng = np.random.default_rng(42)
    ni_total, ns = 403, 600
    codes = rng.choice([0, 1, 2, 3], size=(ns, ni_total), p=[0.28, 0.02, 0.42, 0.28]).astype(np.uint8)
    nb = (ni_total + 3) // 4
    pad = np.zeros((ns, nb * 4), dtype=np.uint8)
    pad[:, :ni_total] = codes
    raw = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
    prefix = str(tmp_path / "syn")
    with open(prefix + ".bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        f.write(raw.tobytes())
    ind = (rng.random(ni_total) > 0.15).astype(np.int32)
    G_all = oracle.bed_decode(raw, ni_total)
    y_all = np.where(np.isnan(G_all[:3]), 0, G_all[:3]).T @ np.array([0.4, -0.3, 0.2]) + rng.standard_normal(ni_total)
    with open(prefix + ".pheno", "w") as f:
        for i in range(ni_total):
            f.write("NA\n" if ind[i] == 0 else "%.17g\n" % y_all[i])
    exe = _build(str(tmp_path))
    out = subprocess.run([exe, prefix, str(ni_total), str(ns), prefix + ".pheno", str(mode), str(tmp_path), "res"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    # oracle: the same two-run sequence, K through the 10-significant-digit cXX.txt
    Kexact = oracle.calc_kin(G_all, 1)
    K = oracle.round10(Kexact)
    Kfile = np.loadtxt(tmp_path / "res.cXX.txt")
    assert Kfile.shape == (ni_total, ni_total)
    # the file holds 10 significant digits: an entry that sits on a rounding boundary may print one unit of the tenth digit
    # away from the rounded restatement (a 1e-15 difference in the sum decides it), never more
    ulp10 = 10.0 ** (np.floor(np.log10(np.maximum(np.abs(Kexact), 1e-300))) - 9)
    err = np.abs(Kfile - Kexact)
    worst = np.unravel_index(np.argmax(err / ulp10), err.shape)
    assert (err <= 0.5 * ulp10 + 1e-13 * np.abs(Kexact).max()).all(), (worst, Kfile[worst], Kexact[worst])
    assert (np.abs(Kfile - K) > 0.01 * ulp10).mean() < 2e-3  # such flips: a few dozen of 162k entries
    W = np.ones((int(ind.sum()), 1))
    st, null, _ = oracle.run_lmm(mode, G_all, ind, np.ones(ns, dtype=np.int32), y_all, W, K)
    lines = open(tmp_path / "res.assoc.txt").read().strip().split("\n")
    hdr = lines[0].split("\t")
    assert len(lines) == ns + 1
    cols = {"l_remle": "lambda_remle", "l_mle": "lambda_mle"}
    tab = np.array([[float(x) for x in ln.split("\t")[7:]] for ln in lines[1:]])
    for j, name in enumerate(hdr[7:]):
        ref = st[cols.get(name, name)]
        got = tab[:, j]
        tol = 1e-3 if name.startswith("l_") else 2e-6  # 6 printed digits + 1e-6 parity
        ok = np.isclose(got, ref, rtol=tol, atol=1e-300, equal_nan=True)
        assert ok.mean() > 0.999 and np.isclose(got, ref, rtol=1e-3, atol=1e-300, equal_nan=True).all(), name
    toks = out.stdout.split()
    assert float(toks[toks.index("l_remle_null") + 1]) == pytest.approx(null["l_remle_null"], rel=1e-3)
    assert float(toks[toks.index("pve") + 1]) == pytest.approx(null["pve"], rel=1e-3)
