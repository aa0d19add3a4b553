"""File in, file out on the MI355X: gemma's own command lines (-g/-p/-c/-a or -bfile; -gk; -k ... -lmm; -eigen; -d/-u)
through tests/cpp/gemma_file_driver.cpp over the HIP library -- first-pass QC, kinship, the 10-digit hand-off, centring,
eigendecomposition, null model and the per-SNP loop all on the device, text parsed by the host thread pool -- against the
files the reference binary wrote for the same inputs (tests/golden/text/).  CPU twin: tests/test_file_driver_cpu.py.

Order: the first five (BXD, PLINK, -loco, multivariate PLINK, -lm) ran green on the MI355X in round 1's last GPU session; the
ones after them were written once that round's GPU minutes were spent -- same driver, same C ABI calls as
tests/test_gpu_reference.py makes through the Python mirror, host side verified by the CPU twin -- and come last so that a
surprise there cannot hide anything in front of them under `pytest -x`."""
import os

import pytest

import filecases as fc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    from gemma_amd import build
    build.build()
    return fc.build_driver(tmp_path_factory.mktemp("drv"), os.path.join(fc.ROOT, "gemma_amd"))


def test_bxd_bimbam_files_to_reference_outputs(driver, tmp_path):
    fc.bxd_bimbam_workflow(driver, tmp_path)


def test_plink_files_to_reference_outputs(driver, tmp_path):
    fc.plink_workflow(driver, tmp_path)


def test_loco_bimbam_files_to_reference_outputs(driver, tmp_path):
    fc.loco_workflow(driver, tmp_path, chrs=(2, 4), modes=(1, 4))


def test_mvlmm_plink_files_to_reference_outputs(driver, tmp_path):
    fc.mvlmm_workflow(driver, tmp_path, modes=(1, 2, 3, 4))


def test_mvlmm_gxe_plink_files_to_reference_outputs(driver, tmp_path):
    """-gxe with two phenotypes, files to .assoc.txt, against the reference's own run (tests/golden/ref_mv_wide.npz)"""
    fc.mvlmm_workflow(driver, tmp_path, modes=(1, 2, 3, 4), gxe=True)


def test_lm_files_to_reference_outputs_and_golden_checksum(driver, tmp_path):
    fc.lm_workflow(driver, tmp_path)


def test_gxe_plink_files_to_reference_outputs(driver, tmp_path):
    fc.gxe_workflow(driver, tmp_path, modes=(1, 4))


def test_gene_expression_file_to_reference_outputs(driver, tmp_path):
    fc.gene_workflow(driver, tmp_path, modes=(1, 4))


def test_snps_notsnp_km2_to_reference_outputs(driver, tmp_path):
    fc.selection_options_workflow(driver, tmp_path)


def test_mvlmm_crt_option_reference_outputs(driver, tmp_path):
    """gemma ... -lmm m -n 1 2 -crt: PCRT's corrected p values (MvNr::crt_factors on the device) against the reference's"""
    fc.mvlmm_workflow(driver, tmp_path, modes=(1, 2, 4), crt=True)


def test_mvlmm_crt_three_traits_reference_outputs(driver, tmp_path):
    """-crt with three traits and missing phenotypes: the 52 SNPs whose p_wald the reference corrects"""
    fc.mvlmm3_workflow(driver, tmp_path, modes=(1,), crt=True)


def test_mvlmm_bimbam_text_to_reference_outputs(driver, tmp_path):
    fc.mvlmm_workflow(driver, tmp_path, modes=(1, 3), bimbam=True)


def test_hwe_filter_to_reference_outputs(driver, tmp_path):
    fc.hwe_workflow(driver, tmp_path)


def test_mvlmm_three_traits_missing_phenotypes(driver, tmp_path):
    fc.mvlmm3_workflow(driver, tmp_path, modes=(1, 3))


def test_standardised_kinship_from_text(driver, tmp_path):
    fc.standardised_kinship_workflow(driver, tmp_path)


def test_two_ranks_on_one_device_kinship_allreduce_eigen_broadcast(driver, tmp_path):
    """The N > 1 protocol of the C++ host with the REAL kernels: two ranks on device 0 (-samegpu: the library's
    shared-memory test transport stands in for RCCL, which refuses two ranks on one device), SNP-sharded kinship with the
    all-reduce, rank-0 eigensolver, one broadcast of the kept (U, eval), SNP-sharded association with the carry seeded."""
    fc.sharded_inproc_workflow(driver, tmp_path, world=2, samegpu=True)


def test_two_ranks_rank0_eigensolver_and_broadcast(driver, tmp_path, monkeypatch):
    """The same with GEMMA_HIP_EIGH_SHARD=0: rank 0 alone decomposes and (U, eval) travel in ONE broadcast (rounds 2-3); the
    default since round 4 is the collective eigensolver above (every rank decomposes, back-transformations shared out)."""
    monkeypatch.setenv("GEMMA_HIP_EIGH_SHARD", "0")
    fc.sharded_inproc_workflow(driver, tmp_path, world=2, samegpu=True)


def test_rccl_entry_points_single_rank(driver):
    """ncclGetUniqueId / ncclCommInitRank / ncclBroadcast / ncclAllReduce through the library on the one device there is:
    a communicator of one rank over the real librccl (dlopen), broadcast and all-reduce are then the identity."""
    import ctypes as C
    import torch
    from gemma_amd import api, _lib as L
    api.init(0)
    lib = L.lib()
    ident = C.create_string_buffer(L.COMM_ID_BYTES)
    import os as _os
    _os.environ.pop("GEMMA_HIP_COMM", None)
    L.check(lib.gemma_hip_comm_unique_id(ident), "comm_unique_id")
    assert any(b != 0 for b in ident.raw)
    L.check(lib.gemma_hip_comm_init(ident, 0, 1), "comm_init")
    x = torch.arange(1000, dtype=torch.float64, device="cuda")
    L.check(lib.gemma_hip_comm_bcast_d(C.c_void_p(x.data_ptr()), x.numel() * 8, 0, None), "bcast")
    L.check(lib.gemma_hip_comm_allreduce_sum_d(C.c_void_p(x.data_ptr()), x.numel(), None), "allreduce")
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float64))
    r, w, t = C.c_int(), C.c_int(), C.c_int()
    lib.gemma_hip_comm_info(C.byref(r), C.byref(w), C.byref(t))
    assert (r.value, w.value) == (0, 1)
    L.check(lib.gemma_hip_comm_finalize(), "comm_finalize")


# BASELINE config 1.  The genotype file of the reference's example (mouse_hs1940.geno.txt.gz, 14 MB) is a blob that neither
# this repository nor the reference checkout here carries; where it is present (GEMMA_EXAMPLE_DIR, or an example/ directory
# next to the repository / under the reference tree) the published rows of example/demo.txt:27-32 and the null-model pve
# of :40-41 are the check.  Absent -> skipped, never faked.
_DEMO_ROWS = [  # rs, beta, se, l_remle, p_wald (example/demo.txt:28-32)
    ("rs3683945", -7.788665e-02, 6.193502e-02, 4.317993e+00, 2.087616e-01),
    ("rs3707673", -6.654282e-02, 6.210234e-02, 4.316144e+00, 2.841271e-01),
    ("rs6269442", -5.344241e-02, 5.377464e-02, 4.323611e+00, 3.204804e-01),
    ("rs6336442", -6.770154e-02, 6.209267e-02, 4.315713e+00, 2.757541e-01),
    ("rs13475700", -5.659089e-02, 7.175374e-02, 4.340145e+00, 4.304306e-01),
]


def _mouse_dir():
    cands = [os.environ.get("GEMMA_EXAMPLE_DIR"), os.path.join(fc.ROOT, "example"), os.path.join(fc.ROOT, "tests", "golden", "example"),
             "/root/reference/example", "/data/gemma/example"]
    for d in cands:
        if d and all(os.path.exists(os.path.join(d, f)) for f in
                     ("mouse_hs1940.geno.txt.gz", "mouse_hs1940.pheno.txt", "mouse_hs1940.anno.txt")):
            return d
    return None


def test_mouse_hs1940_demo_rows(driver, tmp_path):
    d = _mouse_dir()
    if d is None:
        pytest.skip("mouse_hs1940.geno.txt.gz is not available on this machine (blob outside the repository)")
    g, p, a = (os.path.join(d, "mouse_hs1940." + s) for s in ("geno.txt.gz", "pheno.txt", "anno.txt"))
    fc.drive(driver, "-g", g, "-p", p, "-n", 1, "-a", a, "-gk", 1, "-outdir", tmp_path, "-o", "mouse")
    kv = fc.drive(driver, "-g", g, "-p", p, "-n", 1, "-a", a, "-k", tmp_path / "mouse.cXX.txt", "-lmm", 1, "-outdir", tmp_path,
                  "-o", "mouse_lmm")
    hdr, rows = fc.read_assoc(tmp_path / "mouse_lmm.assoc.txt")
    col = {h: i for i, h in enumerate(hdr)}
    by_rs = {r[col["rs"]]: r for r in rows[:50]}
    for rs, beta, se, lam, pw in _DEMO_ROWS:
        r = by_rs[rs]
        for name, ref, tol in (("beta", beta, 2e-6), ("se", se, 2e-6), ("l_remle", lam, 1e-3), ("p_wald", pw, 2e-6)):
            assert float(r[col[name]]) == pytest.approx(ref, rel=tol), (rs, name)
    if "pve" in kv:
        assert float(kv["pve"]) == pytest.approx(0.608801, rel=2e-6)
