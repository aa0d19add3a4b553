"""The reference ITSELF bound to libgemma_hip.so: oracle/_ref_hip/gemma is GEMMA's own source --
CLI, PARAM, readers, QC, null model, writers -- with the four waists of INTEGRATION.md compiled in under
-DGEMMA_WITH_HIP (oracle/hip_patch/apply_patch.py inserts the calls; `make -C oracle ref_hip` builds it in this container,
the binary travels to the GPU box): fast_cblas_dgemm -> gemma_hip_dgemm (kinship accumulation, CalcUtX), CenterMatrix ->
gemma_hip_center, EigenDecomp_Zeroed -> gemma_hip_eigh, LMM::Analyze's batch_compute -> gemma_hip_lmm_setup / lmm_batch /
lmm_finish.  BXD through its own command lines must land on the unpatched reference's outputs (tests/golden/ref_bxd.npz,
written by oracle/_ref/gemma)."""
import os
import subprocess

import numpy as np
import pytest

import filecases as fc

pytestmark = pytest.mark.gpu

EXE = os.path.join(fc.ROOT, "oracle", "_ref_hip", "gemma")


def _run(cwd, *args):
    r = subprocess.run([EXE] + [str(a) for a in args], cwd=str(cwd), capture_output=True)  # its progress bar is not UTF-8
    assert r.returncode == 0, "bound reference failed: %s\n%s\n%s" % (
        " ".join(map(str, args)), r.stdout[-2000:].decode(errors="replace"), r.stderr[-2000:].decode(errors="replace"))
    return r


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref_hip/gemma was not built (needs /root/reference at build time)")
def test_bound_reference_bxd_gk_and_lmm4(tmp_path):
    import gzip
    T = fc.TXT
    plain = {}
    for name in ("bxd_trait", "bxd_cvt", "bxd_anno"):  # the reference reads these three as plain text (only -g through zlib)
        plain[name] = str(tmp_path / (name + ".txt"))
        with gzip.open(os.path.join(T, name + ".txt.gz"), "rb") as f, open(plain[name], "wb") as g:
            g.write(f.read())
    base = ["-g", os.path.join(T, "bxd_mean_genotypes.txt.gz"), "-p", plain["bxd_trait"], "-c", plain["bxd_cvt"],
            "-a", plain["bxd_anno"]]
    full = np.load(os.path.join(fc.ROOT, "tests", "golden", "ref_bxd.npz"))
    _run(tmp_path, *base, "-gk", "-o", "BXD")                       # kinship: Xlarge Xlarge^T on the device
    cxx = tmp_path / "output" / "BXD.cXX.txt"
    K = np.loadtxt(cxx)
    assert K.shape == full["cXX"].shape and np.abs(K - full["cXX"]).max() <= 2e-10  # one unit of the 10th printed digit
    for m in (4, 1):
        _run(tmp_path, *base, "-k", cxx, "-lmm", m, "-maf", "0.1", "-o", "L%d" % m)  # centre, eigh, CalcUtX, per-SNP loop
        got = tmp_path / "output" / ("L%d.assoc.txt" % m)
        fc.compare_assoc(str(got), os.path.join(T, "L%d.assoc.head.txt" % m), n_ref_rows=7317)
        hdr, rows = fc.read_assoc(str(got))
        assert [r[1] for r in rows] == list(full["rs"])
        for j, name in enumerate(hdr[7:], start=7):
            g = np.array([float(r[j]) for r in rows])
            want = full["lmm%d_%s" % (m, name)]
            rel = np.abs(g - want) / np.maximum(np.abs(want), 1e-300)
            rel[np.isnan(g) & np.isnan(want)] = 0.0
            flips = np.isnan(g) != np.isnan(want)
            assert flips.mean() <= 1e-3
            if name in fc.LAM_COLS:
                assert (rel[~flips] <= 1e-3).all() and (rel[~flips] <= fc.STAT_TOL).mean() >= 0.98, (m, name)
            else:
                assert (rel[~flips] <= fc.STAT_TOL).all(), (m, name, np.nanmax(rel[~flips]))
    log = open(tmp_path / "output" / "L4.log.txt").read()
    assert "number of analyzed individuals = 67" in log and "number of analyzed SNPs" in log
