"""The C++ host layer end to end on the CPU: tests/cpp/gemma_file_driver.cpp (readers, first pass, feeders, hand-off
files, writers of include/gemma_io_host.hpp + include/gemma_host.hpp) linked against tests/cpp/abi_double.cpp -- the
C ABI implemented over the oracle, a TEST DOUBLE built into a temporary directory -- and checked against
the files the reference binary wrote for the same inputs (tests/golden/text/).  What this pins is the host side: which
individuals and SNPs are selected, what is fed in which order and layout, and every byte of the output formats.  The
device side of the same workflows is tests/test_gpu_workflow_files.py."""
import os
import subprocess

import pytest

import filecases as fc

ROOT = fc.ROOT


@pytest.fixture(scope="module")
def driver(tmp_path_factory, oracle):
    oracle.lib()  # builds oracle/libgemma_oracle.so
    tmp = tmp_path_factory.mktemp("double")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "abi_double.cpp"), "-L" + os.path.join(ROOT, "oracle"),
                           "-lgemma_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-ldl",
                           "-o", str(tmp / "libgemma_hip.so")])
    return fc.build_driver(tmp, str(tmp))


@pytest.mark.parametrize("block", [None, "97"])
def test_bxd_bimbam_files_to_reference_outputs(driver, tmp_path, monkeypatch, block):
    """block = 97: every feeder (first pass, kinship, association) runs many blocks through the two-slot prefetcher"""
    if block:
        monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", block)
        monkeypatch.setenv("GEMMA_HIP_IO_THREADS", "3")
    fc.bxd_bimbam_workflow(driver, tmp_path, modes=(1, 2, 3, 4, 9) if block is None else (1, 4, 9))


@pytest.mark.parametrize("block", [None, "61"])
def test_plink_files_to_reference_outputs(driver, tmp_path, monkeypatch, block):
    if block:
        monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", block)
    fc.plink_workflow(driver, tmp_path)


def test_loco_bimbam_files_to_reference_outputs(driver, tmp_path, monkeypatch):
    """n = 1008 here: the double takes LAPACK's dsyev from the OpenBLAS inside scipy for the eigenproblem"""
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if libs:
        monkeypatch.setenv("GEMMA_DOUBLE_LAPACK", libs[0])
    monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", "700")
    fc.loco_workflow(driver, tmp_path, chrs=(2,), modes=(1,))


def test_mvlmm_plink_files_to_reference_outputs(driver, tmp_path, monkeypatch):
    """class MVLMM of the C++ host layer (null block, per-SNP blocks, WriteFiles) over the double's oracle-backed mvLMM"""
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if libs:
        monkeypatch.setenv("GEMMA_DOUBLE_LAPACK", libs[0])
    monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", "300")
    fc.mvlmm_workflow(driver, tmp_path, modes=(1, 4))


def test_mvlmm_gxe_plink_files_to_reference_outputs(driver, tmp_path, monkeypatch):
    """-gxe with two phenotypes through the host layer (MVLMM::AnalyzePlinkGXE: null fit on (W, env), gemma_mvlmm_opt.gxe) against
    the reference's own -gxe run"""
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if libs:
        monkeypatch.setenv("GEMMA_DOUBLE_LAPACK", libs[0])
    monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", "60")
    fc.mvlmm_workflow(driver, tmp_path, modes=(4,), gxe=True)


def test_mvlmm_crt_option_reference_outputs(driver, tmp_path, monkeypatch):
    """-crt through the host layer (PARAM::crt -> gemma_mvlmm_opt.crt) against the reference run with -crt"""
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if libs:
        monkeypatch.setenv("GEMMA_DOUBLE_LAPACK", libs[0])
    monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", "300")
    fc.mvlmm_workflow(driver, tmp_path, modes=(4,), crt=True)


def test_mvlmm_crt_three_traits_reference_outputs(driver, tmp_path, monkeypatch):
    """-crt with three traits and missing phenotypes: 52 SNPs reach MphNR and get PCRT's corrected p_wald in the reference"""
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if libs:
        monkeypatch.setenv("GEMMA_DOUBLE_LAPACK", libs[0])
    monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", "700")
    fc.mvlmm3_workflow(driver, tmp_path, modes=(1,), crt=True)


def test_mvlmm_three_traits_missing_phenotypes(driver, tmp_path, monkeypatch):
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if libs:
        monkeypatch.setenv("GEMMA_DOUBLE_LAPACK", libs[0])
    fc.mvlmm3_workflow(driver, tmp_path, modes=(1,))


def test_mvlmm_bimbam_text_to_reference_outputs(driver, tmp_path, monkeypatch):
    """MVLMM::AnalyzeBimbam from the text file (threaded reader, one block ahead)"""
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if libs:
        monkeypatch.setenv("GEMMA_DOUBLE_LAPACK", libs[0])
    monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", "300")
    fc.mvlmm_workflow(driver, tmp_path, modes=(1,), bimbam=True)


def test_lm_files_to_reference_outputs_and_golden_checksum(driver, tmp_path, monkeypatch):
    monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", "1000")
    fc.lm_workflow(driver, tmp_path)


def test_gxe_plink_files_to_reference_outputs(driver, tmp_path, monkeypatch):
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if libs:
        monkeypatch.setenv("GEMMA_DOUBLE_LAPACK", libs[0])
    monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", "800")
    fc.gxe_workflow(driver, tmp_path, modes=(1,))


def test_gene_expression_file_to_reference_outputs(driver, tmp_path, monkeypatch):
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if libs:
        monkeypatch.setenv("GEMMA_DOUBLE_LAPACK", libs[0])
    monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", "16")  # 40 genes in three blocks
    fc.gene_workflow(driver, tmp_path, modes=(1, 4))


def test_snps_notsnp_km2_to_reference_outputs(driver, tmp_path):
    fc.selection_options_workflow(driver, tmp_path)


def test_hwe_filter_to_reference_outputs(driver, tmp_path):
    fc.hwe_workflow(driver, tmp_path)


def test_standardised_kinship_from_text(driver, tmp_path):
    fc.standardised_kinship_workflow(driver, tmp_path)


def test_driver_reports_reader_errors(driver, tmp_path):
    """Where the reference's readers return false the driver stops (exit code 3) instead of analysing garbage."""
    bad = tmp_path / "short.txt"
    bad.write_text("rs1, A, G, 0, 1\n")
    r = subprocess.run([driver, "-g", str(bad), "-p", os.path.join(fc.TXT, "bxd_trait.txt.gz"), "-gk", "-outdir", str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode == 3 and "not enough genotypes" in r.stdout
    r = subprocess.run([driver, "-bfile", str(tmp_path / "nothing"), "-gk", "-outdir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 3 and "error opening .bim file" in r.stdout


def test_driver_edge_cases(driver, tmp_path):
    """Empty and ragged inputs end where the reference's readers end: no analysed individual, a truncated .bed, a kinship
    file of the wrong size; an input without a single SNP line is an empty (not a crashing) first pass."""
    import numpy as np
    out = str(tmp_path)
    geno = os.path.join(fc.TXT, "bxd_mean_genotypes_head.txt")
    import gzip
    ni = len(gzip.open(os.path.join(fc.TXT, "bxd_trait.txt.gz"), "rt").read().split())
    allna = tmp_path / "na.txt"
    allna.write_text("NA\n" * ni)
    r = subprocess.run([driver, "-g", geno, "-p", str(allna), "-gk", "-outdir", out], capture_output=True, text=True)
    assert r.returncode == 3 and "number of analyzed individuals equals 0" in r.stdout
    # truncated .bed: the first pass reports it instead of analysing a short block
    pre = str(tmp_path / "T")
    for ext in (".bim", ".fam"):
        open(pre + ext, "w").write(open(os.path.join(fc.TXT, "P" + ext)).read())
    raw = open(os.path.join(fc.TXT, "P.bed"), "rb").read()
    open(pre + ".bed", "wb").write(raw[: len(raw) // 2])
    r = subprocess.run([driver, "-bfile", pre, "-gk", "-outdir", out], capture_output=True, text=True)
    assert r.returncode == 3 and "truncated" in r.stdout
    # kinship file with one row too few / one column too many for the .fam
    K = np.eye(240)
    np.savetxt(tmp_path / "short.cXX.txt", K[:239], fmt="%.10g", delimiter="\t")
    r = subprocess.run([driver, "-bfile", os.path.join(fc.TXT, "P"), "-k", str(tmp_path / "short.cXX.txt"), "-lmm", "1",
                        "-outdir", out], capture_output=True, text=True)
    assert r.returncode == 5 and "rows in the kinship file" in r.stdout
    np.savetxt(tmp_path / "wide.cXX.txt", np.eye(240, 241), fmt="%.10g", delimiter="\t")
    r = subprocess.run([driver, "-bfile", os.path.join(fc.TXT, "P"), "-k", str(tmp_path / "wide.cXX.txt"), "-lmm", "1",
                        "-outdir", out], capture_output=True, text=True)
    assert r.returncode == 5 and "columns in the kinship file" in r.stdout
    # no SNP at all: an empty genotype file passes the first pass with zero SNPs
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    r = subprocess.run([driver, "-g", str(empty), "-p", os.path.join(fc.TXT, "bxd_trait.txt.gz"), "-lm", "1", "-outdir", out,
                        "-o", "none"], capture_output=True, text=True)
    assert r.returncode == 0 and "ns_total=0 ns_test=0" in r.stdout and "snps=0" in r.stdout
    assert len(open(tmp_path / "none.assoc.txt").read().strip().split("\n")) == 1  # the header alone


def test_feeders_are_race_free_under_thread_sanitizer(driver, tmp_path, monkeypatch):
    """The host layer runs three kinds of helper threads (the prefetch producer, the text parsers, the row formatters)
    beside the thread that owns the C ABI: the driver rebuilt with -fsanitize=thread goes through first pass, kinship and
    association in many small blocks and must finish without a ThreadSanitizer report, with the same output bytes."""
    exe = str(tmp_path / "driver_tsan")
    libdir = os.path.dirname(driver)
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "gemma_file_driver.cpp"), "-L" + libdir, "-lgemma_hip",
                        "-Wl,-rpath," + libdir, "-lz", "-pthread", "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime here: " + r.stderr[-200:])
    monkeypatch.setenv("GEMMA_HIP_IO_BLOCK", "97")
    monkeypatch.setenv("GEMMA_HIP_IO_THREADS", "4")
    T = fc.TXT
    base = ["-g", os.path.join(T, "bxd_mean_genotypes.txt.gz"), "-p", os.path.join(T, "bxd_trait.txt.gz"),
            "-c", os.path.join(T, "bxd_cvt.txt.gz"), "-a", os.path.join(T, "bxd_anno.txt.gz"), "-outdir", str(tmp_path)]
    runs = [base + ["-gk", "-o", "B"],
            base + ["-k", str(tmp_path / "B.cXX.txt"), "-lmm", "1", "-maf", "0.1", "-o", "L"],
            ["-bfile", os.path.join(T, "P"), "-outdir", str(tmp_path), "-gk", "-o", "P"],
            ["-bfile", os.path.join(T, "P"), "-outdir", str(tmp_path), "-k", str(tmp_path / "P.cXX.txt"), "-lmm", "4", "-o", "P4"]]
    for args in runs:
        r = subprocess.run([exe] + args, capture_output=True, text=True)
        if "FATAL: ThreadSanitizer" in r.stderr:  # the runtime cannot map its shadow memory on this kernel / ASLR setting
            pytest.skip("ThreadSanitizer runtime unusable here: " + r.stderr.strip().split("\n")[0])
        assert r.returncode == 0, r.stdout + r.stderr
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert open(tmp_path / "L.assoc.txt").read().split("\n")[:120] == open(os.path.join(T, "L1.assoc.head.txt")).read().split("\n")[:120]
    fc.compare_assoc(str(tmp_path / "P4.assoc.txt"), os.path.join(T, "P4.assoc.txt.gz"))


@pytest.mark.parametrize("world", [2, 3])
def test_snp_sharded_ranks_write_the_single_process_file(driver, tmp_path, world):
    """SURVEY 8e at the C++ host level: `-gpus N` forks one process per GPU; every rank analyses the contiguous share
    [r ceil(p/N), (r+1) ceil(p/N)) of the analysed SNPs (shard_range, the rule of gemma_amd/dist.py) and writes its part, the
    parent concatenates the parts in rank order -- byte for byte the file of the single-process run, for PLINK and for
    BIMBAM text input.  (Here every rank runs on the CPU test double; on a GPU node rank r takes device r.)"""
    out = str(tmp_path)
    T = fc.TXT
    pb = ["-bfile", os.path.join(T, "P"), "-outdir", out]
    fc.drive(driver, *pb, "-gk", "-o", "P")
    cxx = os.path.join(out, "P.cXX.txt")
    fc.drive(driver, *pb, "-k", cxx, "-lmm", 4, "-c", os.path.join(T, "P.cov.txt"), "-o", "one")
    kv = fc.drive(driver, *pb, "-k", cxx, "-lmm", 4, "-c", os.path.join(T, "P.cov.txt"), "-gpus", world, "-o", "many")
    assert int(kv["ranks"]) == world
    assert open(os.path.join(out, "many.assoc.txt"), "rb").read() == open(os.path.join(out, "one.assoc.txt"), "rb").read()
    assert not [f for f in os.listdir(out) if ".rank" in f]
    fc.compare_assoc(os.path.join(out, "many.assoc.txt"), os.path.join(T, "P4c.assoc.txt.gz"))
    bb = ["-g", os.path.join(T, "bxd_mean_genotypes.txt.gz"), "-p", os.path.join(T, "bxd_trait.txt.gz"),
          "-c", os.path.join(T, "bxd_cvt.txt.gz"), "-a", os.path.join(T, "bxd_anno.txt.gz"), "-outdir", out]
    fc.drive(driver, *bb, "-gk", "-o", "B")
    fc.drive(driver, *bb, "-k", os.path.join(out, "B.cXX.txt"), "-lmm", 1, "-maf", "0.1", "-o", "b1")
    fc.drive(driver, *bb, "-k", os.path.join(out, "B.cXX.txt"), "-lmm", 1, "-maf", "0.1", "-gpus", world, "-o", "bN")
    assert open(os.path.join(out, "bN.assoc.txt"), "rb").read() == open(os.path.join(out, "b1.assoc.txt"), "rb").read()


def test_snp_sharded_ranks_lm_and_multivariate(driver, tmp_path, monkeypatch):
    """the same for `-lm` (class LM) and the multivariate LMM (class MVLMM): 2 ranks, parts concatenated = one process"""
    import glob
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if libs:
        monkeypatch.setenv("GEMMA_DOUBLE_LAPACK", libs[0])
    out = str(tmp_path)
    T = fc.TXT
    pb = ["-bfile", os.path.join(T, "P"), "-outdir", out]
    fc.drive(driver, *pb, "-lm", 4, "-o", "lm1")
    fc.drive(driver, *pb, "-lm", 4, "-gpus", 2, "-o", "lm2")
    assert open(os.path.join(out, "lm2.assoc.txt"), "rb").read() == open(os.path.join(out, "lm1.assoc.txt"), "rb").read()
    hb = ["-bfile", os.path.join(T, "H"), "-outdir", out]  # two correlated traits from the single-trait set: trait, trait^2
    fam = [l.split() for l in open(os.path.join(T, "H.fam"))]
    with open(os.path.join(out, "H2.fam"), "w") as f:
        for r in fam:
            y = float(r[5])
            f.write(" ".join(r[:5]) + (" -9 -9\n" if y == -9 else " %r %r\n" % (y, 0.3 * y * y - y)))
    for ext in (".bed", ".bim"):
        open(os.path.join(out, "H2" + ext), "wb").write(open(os.path.join(T, "H" + ext), "rb").read())
    h2 = ["-bfile", os.path.join(out, "H2"), "-outdir", out]
    fc.drive(driver, *h2, "-gk", "-o", "H2")
    cxx = os.path.join(out, "H2.cXX.txt")
    fc.drive(driver, *h2, "-k", cxx, "-lmm", 1, "-n", 1, 2, "-o", "mv1")
    fc.drive(driver, *h2, "-k", cxx, "-lmm", 1, "-n", 1, 2, "-gpus", 2, "-o", "mv2")
    assert open(os.path.join(out, "mv2.assoc.txt"), "rb").read() == open(os.path.join(out, "mv1.assoc.txt"), "rb").read()
    assert len(open(os.path.join(out, "mv1.assoc.txt")).read().strip().split("\n")) == 601
    # the multivariate -gxe (MVLMM::AnalyzePlinkGXE) shards the same way: an environment column over all individuals
    import numpy as np
    rng = np.random.default_rng(7)
    np.savetxt(os.path.join(out, "env.txt"), np.round(rng.standard_normal(len(fam)), 5), fmt="%.5f")
    with open(os.path.join(out, "few.txt"), "w") as f:
        f.writelines(l.split()[1] + "\n" for l in list(open(os.path.join(T, "H.bim")))[::10])
    gx = ["-k", cxx, "-lmm", 4, "-n", 1, 2, "-gxe", os.path.join(out, "env.txt"), "-snps", os.path.join(out, "few.txt")]
    fc.drive(driver, *h2, *gx, "-o", "gx1")
    fc.drive(driver, *h2, *gx, "-gpus", 2, "-o", "gx2")
    assert open(os.path.join(out, "gx2.assoc.txt"), "rb").read() == open(os.path.join(out, "gx1.assoc.txt"), "rb").read()
    assert len(open(os.path.join(out, "gx1.assoc.txt")).read().strip().split("\n")) >= 50


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_kinship_allreduce_and_eigen_broadcast(driver, tmp_path, world):
    """`-inproc 1 -gpus N` and `-k K -gpus N` through the communicator entry points of the C ABI (the double runs them over
    the host shared-memory transport the library itself uses for its 1-GPU tests): SNP-sharded kinship + all-reduce, rank-0
    eigendecomposition + ONE broadcast of (U, eval), SNP-sharded association."""
    fc.sharded_inproc_workflow(driver, tmp_path, world=world)


def test_shard_range_rule_matches_the_python_side():
    from gemma_amd.dist import shard_range
    for p in (0, 1, 7, 574, 1000001):
        for w in (1, 2, 3, 8):
            per = (p + w - 1) // w
            for r in range(w):
                b = min(p, per * r)
                assert shard_range(p, r, w) == (b, min(p, b + per))


def test_failed_rank_does_not_hang_the_run(driver, tmp_path):
    """with `-k K -gpus N` only rank 0 reads the kinship; when that fails (here: a truncated file) the other ranks
    sit in the broadcast, which has no time-out.  The parent must end them and report the failure instead of waiting for ever."""
    import subprocess
    out = str(tmp_path)
    base = ["-bfile", os.path.join(fc.TXT, "P"), "-outdir", out]
    fc.drive(driver, *base, "-gk", "-o", "P")
    good = open(os.path.join(out, "P.cXX.txt")).read().split("\n")
    bad = os.path.join(out, "bad.cXX.txt")
    with open(bad, "w") as f:
        f.write("\n".join(good[: len(good) // 2]) + "\nnot_a_number\t1\n")
    r = subprocess.run([driver] + [str(a) for a in base + ["-k", bad, "-lmm", 1, "-gpus", 2, "-o", "hang"]],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert not os.path.exists(os.path.join(out, "hang.assoc.txt"))
