"""Host layer either side of the hot path (include/gemma_io_host.hpp), on the CPU:
the readers that select individuals and SNPs, the multi-threaded BIMBAM text parser and the writers of the
artefacts GEMMA leaves between runs -- against the reference binary's own files (tests/golden/text/, written by
tests/golden/make_text_fixtures.py from oracle/_ref/gemma) byte for byte, and against libc's atof for every token.
The harness (tests/cpp/io_host_check.cpp) makes no device call."""
import ctypes
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TXT = os.path.join(ROOT, "tests", "golden", "text")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    from gemma_amd import build
    build.build()  # the header's inline wrappers reference the C ABI; the library loads without a GPU
    out = str(tmp_path_factory.mktemp("iohost") / "io_host_check")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "io_host_check.cpp"),
                           "-L" + os.path.join(ROOT, "gemma_amd"), "-lgemma_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "gemma_amd"), "-lz", "-pthread", "-o", out])
    return out


def run(exe, *args, stdin=None, ok=True):
    r = subprocess.run([exe] + [str(a) for a in args], input=stdin, capture_output=True, text=True)
    if ok:
        assert r.returncode == 0, r.stdout + r.stderr
    return r


def test_parse_double_is_atof(exe):
    libc = ctypes.CDLL(None)
    libc.atof.restype = ctypes.c_double
    libc.atof.argtypes = [ctypes.c_char_p]
    rng = np.random.default_rng(11)
    toks = ["0", "1", "2", "0.5", "-0", "+1.5", "1.", ".5", ".", "-", "1e", "1e+", "1e5", "1E-3", "2.5e22", "2.5e23", "1e-22",
            "1e-23", "9007199254740992", "9007199254740993", "9007199254740993e3", "0.1", "0.30000000000000004",
            "123456789012345678901234567890", "0.000000000000000000000000000001", "1e400", "1e-400", "0x10", "inf", "-inf",
            "nan", "NAN", "abc", "1.5abc", "1,5", "4.9e-324", "2.2250738585072014e-308", "1.7976931348623157e308",
            "0.1e1", "00012.500", "12345678901234567890e-5", "1e0010", "-1.25E+2", "8.98846567431158e307",
            "0.500000000000000166533453693773481063544750213623046875", "17.000000000000000000001"]
    for _ in range(4000):
        kind = rng.integers(0, 5)
        if kind == 0:
            toks.append("%.3f" % rng.uniform(0, 2))
        elif kind == 1:
            toks.append(repr(float(rng.standard_normal() * 10.0 ** rng.integers(-30, 30))))
        elif kind == 2:
            toks.append("%.*e" % (int(rng.integers(0, 20)), rng.standard_normal() * 10.0 ** rng.integers(-5, 5)))
        elif kind == 3:
            toks.append("%d.%0*d" % (rng.integers(0, 10 ** 6), int(rng.integers(1, 18)), rng.integers(0, 10 ** 9)))
        else:
            toks.append("%de%d" % (rng.integers(-10 ** 17, 10 ** 17), rng.integers(-30, 30)))
    alphabet = list("0123456789.eE+-xXpPnNaAiIfF")
    for _ in range(3000):  # fuzz: whatever strtod makes of it, parse_double must make the same
        toks.append("".join(rng.choice(alphabet, size=int(rng.integers(1, 13)))))
    out = run(exe, "parse", stdin="\n".join(toks) + "\n").stdout.split()
    assert len(out) == len(toks)
    for t, h in zip(toks, out):
        want = np.float64(libc.atof(t.encode())).view(np.uint64)
        assert int(h, 16) == int(want), (t, h, hex(int(want)))


def _tok(line):
    return line.replace(",", " ").split()


def test_bxd_individual_selection_matches_reference_log(exe, oracle, tmp_path):
    """ReadFile_pheno + ReadFile_cvt + ProcessCvtPhen on the BXD example: the counts the reference's log reports,
    W / y equal to the oracle's restatement of the same functions."""
    log = json.load(open(os.path.join(TXT, "L1.log.json")))
    ph_gz, cv_gz = os.path.join(TXT, "bxd_trait.txt.gz"), os.path.join(TXT, "bxd_cvt.txt.gz")
    out = run(exe, "cvtphen", ph_gz, cv_gz, 1).stdout.split("\n")  # the C++ readers take gzip directly
    ph, cv = str(tmp_path / "trait.txt"), str(tmp_path / "cvt.txt")  # the oracle's readers want plain text
    for src, dst in ((ph_gz, ph), (cv_gz, cv)):
        open(dst, "wb").write(gzip.open(src, "rb").read())
    ni_test, n_cvt = map(int, out[0].split())
    ind = np.array(out[1].split(), dtype=int)
    assert ni_test == int(log["number of analyzed individuals"]) and n_cvt == int(log["number of covariates"])
    assert ind.size == int(log["number of total individuals"]) and ind.sum() == ni_test
    y_all, ind_ph = oracle.read_pheno(ph, 1)
    cvt, ind_cvt = oracle.read_cvt(cv)
    ind_ref, W_ref = oracle.process_cvt_phen(ind_ph, cvt, ind_cvt)
    assert (ind == ind_ref).all()
    W = np.array(out[2].split(), dtype=float).reshape(ni_test, n_cvt)
    y = np.array(out[3].split(), dtype=float)
    assert (W == W_ref).all() and (y == y_all[ind_ref == 1]).all()


@pytest.mark.parametrize("case", ["no_intercept", "all_constant", "with_intercept", "na_rows", "none"])
def test_cvt_intercept_rules(exe, oracle, tmp_path, case):
    """PARAM::CheckCvt (src/param.cpp:1937-1990): a column of 1s is appended when no column is constant, covariates
    made only of constant columns are dropped, rows with NA leave the analysis."""
    rng = np.random.default_rng(5)
    n = 40
    y = rng.standard_normal(n)
    miss = rng.random(n) < 0.2
    ph = tmp_path / "p.txt"
    ph.write_text("".join("NA\n" if m else "%r\n" % float(v) for v, m in zip(y, miss)))
    cv = tmp_path / "c.txt"
    C = rng.standard_normal((n, 2))
    if case == "all_constant":
        C[:] = [1.0, 3.0]
    if case == "with_intercept":
        C[:, 0] = 1.0
    rows = [" ".join(repr(float(v)) for v in r) for r in C]
    if case == "na_rows":
        rows[3] = "NA 0.5"
        rows[7] = "0.25\tNA"
    cv.write_text("\n".join(rows) + "\n")
    out = run(exe, "cvtphen", ph, "-" if case == "none" else cv, 1).stdout.split("\n")
    ni_test, n_cvt = map(int, out[0].split())
    ind = np.array(out[1].split(), dtype=int)
    y_all, ind_ph = oracle.read_pheno(str(ph), 1)
    if case == "none":
        ind_ref, W_ref = oracle.process_cvt_phen(ind_ph)
    else:
        cvt, ind_cvt = oracle.read_cvt(str(cv))
        ind_ref, W_ref = oracle.process_cvt_phen(ind_ph, cvt, ind_cvt)
    assert (ind == ind_ref).all() and ni_test == ind_ref.sum()
    W = np.array(out[2].split(), dtype=float).reshape(ni_test, n_cvt)
    expect = {"no_intercept": 3, "all_constant": 1, "with_intercept": 2, "na_rows": 3, "none": 1}[case]
    assert n_cvt == expect and W.shape == W_ref.shape and (W == W_ref).all()
    if case == "na_rows":
        assert ind[3] == 0 and ind[7] == 0


def test_fam_reader_missing_codes_and_columns(exe, tmp_path):
    """ReadFile_fam (src/gemma_io.cpp:559-635): phenotype n is column 5 + n; NA and -9 both mean missing."""
    lines = ["f1 i1 0 0 1 1.5 2.5 NA", "f2 i2 0 0 2 -9 0.25 7", "f3\ti3\t0\t0\t1\tNA\t-9.0\t8", "f4 i4 0 0 1 3 4 5"]
    fam = tmp_path / "x.fam"
    fam.write_text("\n".join(lines) + "\n")
    out = run(exe, "fam", fam, 1, 3).stdout.strip().split("\n")
    assert out[-1] == "ids 4"
    got = [l.split() for l in out[:-1]]
    assert got == [["1", "1.5", "0", "-9"], ["0", "-9", "1", "7"], ["0", "-9", "1", "8"], ["1", "3", "1", "5"]]
    out = run(exe, "fam", fam, 2).stdout.strip().split("\n")
    assert [l.split() for l in out[:-1]] == [["1", "2.5"], ["1", "0.25"], ["0", "-9"], ["1", "4"]]
    short = tmp_path / "s.fam"
    short.write_text("f1 i1 0 0 1\n")
    assert run(exe, "fam", short, 1, ok=False).returncode == 1


def test_bim_and_anno_readers(exe, tmp_path):
    bim = tmp_path / "x.bim"
    bim.write_text("1\trs1\t0\t1000\tA\tG\n2 rs2 0.5 2000 C T\nX\trs3\t1e-2\t3000\tG\tA\r\n")
    got = run(exe, "bim", bim).stdout.strip().split("\n")
    assert got == ["1 rs1 0 1000 A G", "2 rs2 0.5 2000 C T", "X rs3 0.01 3000 G A"]
    got = run(exe, "anno", os.path.join(TXT, "bxd_anno_head.txt")).stdout.strip().split("\n")
    ref = sorted(l.split() for l in open(os.path.join(TXT, "bxd_anno_head.txt")))
    assert [g.split()[:3] for g in got] == [[r[0], r[1], r[2]] for r in ref] and all(g.split()[3] == "-9" for g in got)
    an = tmp_path / "a.txt"
    an.write_text("rsA, 100, 3, 0.5\nrsB, NA, NA\nrsC\t7\n")
    got = run(exe, "anno", an).stdout.strip().split("\n")
    assert got == ["rsA 100 3 0.5", "rsB -9 -9 -9", "rsC 7 -9 -9"]


def _expected_rows(lines, ni_total):
    X = np.full((len(lines), ni_total), np.nan)
    names = []
    for r, line in enumerate(lines):
        t = _tok(line)
        names.append(" ".join(t[:3]))
        for i in range(ni_total):
            if t[3 + i] != "NA":
                X[r, i] = float(t[3 + i])
    return X, names


@pytest.mark.parametrize("threads,block", [(1, 1000), (8, 1000), (8, 37), (3, 16)])
def test_bimbam_reader_bxd_head(exe, tmp_path, threads, block):
    """The threaded BIMBAM parser on the first lines of the reference's own BXD genotype file: every value the double
    atof gives, NA -> NaN, rs / alleles kept, whatever the thread count and block size."""
    src = os.path.join(TXT, "bxd_mean_genotypes_head.txt")
    lines = [l for l in open(src).read().split("\n") if l]
    ni_total = len(_tok(lines[0])) - 3
    want, names = _expected_rows(lines, ni_total)
    out = tmp_path / "x.bin"
    r = run(exe, "geno", src, ni_total, threads, block, out)
    got = np.fromfile(out).reshape(-1, ni_total)
    assert got.shape == want.shape and (np.isnan(got) == np.isnan(want)).all()
    assert (got[~np.isnan(got)] == want[~np.isnan(want)]).all()
    assert r.stdout.strip().split("\n") == names


def test_bimbam_reader_formats_gzip_selection(exe, tmp_path):
    """Tabs / commas / blanks as separators, CR-LF line ends, exponent and long-mantissa dosages, a gzip-compressed
    file, SNP rows dropped through `keep` and individuals through `cols` (what BimbamKin / AnalyzeBimbam ask for)."""
    rng = np.random.default_rng(3)
    ni_total, ns = 53, 700
    lines = []
    for s in range(ns):
        vals = []
        for i in range(ni_total):
            u = rng.random()
            if u < 0.05:
                vals.append("NA")
            elif u < 0.5:
                vals.append(str(int(rng.integers(0, 3))))
            elif u < 0.8:
                vals.append("%.3f" % rng.uniform(0, 2))
            elif u < 0.9:
                vals.append(repr(float(rng.uniform(0, 2))))
            else:
                vals.append("%.12e" % rng.uniform(0, 2))
        sep = [", ", "\t", " ", ","][s % 4]
        lines.append(sep.join(["rs%d" % s, "A", "G"] + vals))
    path = tmp_path / "g.txt.gz"
    with gzip.open(path, "wt", newline="") as f:
        f.write("\r\n".join(lines) + "\r\n")
    want, names = _expected_rows(lines, ni_total)
    out = tmp_path / "all.bin"
    r = run(exe, "geno", path, ni_total, 8, 64, out)
    got = np.fromfile(out).reshape(-1, ni_total)
    assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True)
    assert r.stdout.strip().split("\n") == names
    # a text cap of a few lines: read_block hands back partial blocks, the caller loops -- same rows
    capped = tmp_path / "capped.bin"
    r2 = subprocess.run([exe, "geno", str(path), str(ni_total), "8", "64", str(capped)], capture_output=True, text=True,
                        env=dict(os.environ, GEMMA_HIP_IO_TEXT_CAP="2048"))
    assert r2.returncode == 0 and open(capped, "rb").read() == open(out, "rb").read() and r2.stdout == r.stdout
    keep = (rng.random(ns) < 0.6).astype(int)
    cols = (rng.random(ni_total) < 0.7).astype(int)
    (tmp_path / "keep.txt").write_text(" ".join(map(str, keep)))
    (tmp_path / "cols.txt").write_text(" ".join(map(str, cols)))
    out2 = tmp_path / "sel.bin"
    r = run(exe, "geno", path, ni_total, 5, 50, out2, tmp_path / "keep.txt", tmp_path / "cols.txt")
    got = np.fromfile(out2).reshape(-1, int(cols.sum()))
    assert np.array_equal(got, want[keep == 1][:, cols == 1], equal_nan=True)
    assert r.stdout.strip().split("\n") == [n for n, k in zip(names, keep) if k]
    bad = tmp_path / "bad.txt"
    bad.write_text("\n".join(lines[:5]) + "\nrsX A G 1 2\n")
    assert run(exe, "geno", bad, ni_total, 4, 64, tmp_path / "bad.bin", ok=False).returncode == 1


def test_bimbam_reader_buffer_stays_bounded_over_a_dropped_prefix(exe, tmp_path):
    """A shard of rank r > 0 (or -loco / -snps with the kept SNPs late in the file) drops a long run of leading lines:
    the reader must discard their text as it goes instead of buffering the whole prefix (the text buffer then tracks
    one read chunk; the cap below makes a chunk 4 KiB so that a 1.5 MB file shows it)."""
    ni_total, ns = 40, 9000
    rng = np.random.default_rng(8)
    lines = ["rs%d A G " % s + " ".join(str(int(v)) for v in rng.integers(0, 3, ni_total)) for s in range(ns)]
    path = tmp_path / "long.txt"
    path.write_text("\n".join(lines) + "\n")
    keep = np.zeros(ns, dtype=int)
    keep[-10:] = 1
    (tmp_path / "keep.txt").write_text(" ".join(map(str, keep)))
    out = tmp_path / "tail.bin"
    r = subprocess.run([exe, "geno", str(path), str(ni_total), "4", "64", str(out), str(tmp_path / "keep.txt")],
                       capture_output=True, text=True, env=dict(os.environ, GEMMA_HIP_IO_TEXT_CAP="4096"))
    assert r.returncode == 0, r.stderr
    want, names = _expected_rows(lines[-10:], ni_total)
    assert np.array_equal(np.fromfile(out).reshape(-1, ni_total), want)
    assert r.stdout.strip().split("\n") == names
    cap = int(r.stderr.split("text_buffer_bytes")[1].split()[0])
    assert cap < 64 * 1024 < os.path.getsize(path) // 10, cap


def test_eigen_artefacts_byte_identical_to_reference(exe, tmp_path):
    """`-eigen` writes <o>.eigenU.txt / <o>.eigenD.txt (src/gemma.cpp:1779-1800): reading the reference's files with
    ReadFile_eigenU / ReadFile_eigenD and writing them back through WriteEigen gives the same bytes."""
    n = len(open(os.path.join(TXT, "E.eigenD.txt")).read().split())
    run(exe, "eigen", os.path.join(TXT, "E.eigenU.txt"), os.path.join(TXT, "E.eigenD.txt"), n, tmp_path, "R")
    for suf in ("eigenU", "eigenD"):
        assert open(tmp_path / ("R.%s.txt" % suf), "rb").read() == open(os.path.join(TXT, "E.%s.txt" % suf), "rb").read()
    assert run(exe, "eigen", os.path.join(TXT, "E.eigenU.txt"), os.path.join(TXT, "E.eigenD.txt"), n - 1, tmp_path, "X",
               ok=False).returncode == 1


def test_kinship_text_byte_identical_to_reference(exe, tmp_path):
    """ReadFile_kin -> PARAM::WriteMatrix (precision(10), tabs, src/param.cpp:1886-1911) on the reference's cXX corner"""
    src = os.path.join(TXT, "BXD.cXX.corner.txt")
    run(exe, "kin", src, 24, tmp_path / "k.txt")
    assert open(tmp_path / "k.txt", "rb").read() == open(src, "rb").read()


def test_threaded_kinship_reader_drops_individuals_and_rejects_bad_files(exe, tmp_path):
    """ReadFile_kin_threaded == ReadFile_kin bit for bit (the harness compares them) with rows / columns of non-analysed
    individuals dropped; a row with a surplus column, a missing row and a surplus row are errors as in the reference."""
    rng = np.random.default_rng(9)
    n = 57
    K = rng.standard_normal((n, n))
    src = tmp_path / "k.txt"
    src.write_text("\n".join("\t".join("%.10g" % v for v in r) for r in K) + "\n")
    ind = (rng.random(n) < 0.7).astype(int)
    (tmp_path / "ind.txt").write_text(" ".join(map(str, ind)))
    run(exe, "kin", src, n, tmp_path / "sub.txt", tmp_path / "ind.txt")
    got = np.loadtxt(tmp_path / "sub.txt")
    want = np.loadtxt(src)[ind == 1][:, ind == 1]
    assert np.array_equal(got, want)
    # a text cap far below the file size: the reader returns partial blocks and the kinship reader loops over them
    r = subprocess.run([exe, "kin", str(src), str(n), str(tmp_path / "sub2.txt"), str(tmp_path / "ind.txt")],
                       capture_output=True, text=True, env=dict(os.environ, GEMMA_HIP_IO_TEXT_CAP="3000"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(tmp_path / "sub2.txt", "rb").read() == open(tmp_path / "sub.txt", "rb").read()
    lines = src.read_text().strip().split("\n")
    bad = tmp_path / "bad1.txt"
    r = int(np.flatnonzero(ind == 1)[5])  # rows of dropped individuals are skipped unparsed, as in the reference
    bad.write_text("\n".join(lines[:r] + [lines[r] + "\t0.5"] + lines[r + 1:]) + "\n")
    assert run(exe, "kin", bad, n, tmp_path / "o.txt", tmp_path / "ind.txt", ok=False).returncode == 1
    bad.write_text("\n".join(lines[:-1]) + "\n")
    assert run(exe, "kin", bad, n, tmp_path / "o.txt", tmp_path / "ind.txt", ok=False).returncode == 1
    bad.write_text("\n".join(lines + [lines[0]]) + "\n")
    assert run(exe, "kin", bad, n, tmp_path / "o.txt", tmp_path / "ind.txt", ok=False).returncode == 1
    bad.write_text("\n".join(lines[:r] + ["\t".join(lines[r].split("\t")[:-1])] + lines[r + 1:]) + "\n")
    assert run(exe, "kin", bad, n, tmp_path / "o.txt", tmp_path / "ind.txt", ok=False).returncode == 1


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 9])
def test_assoc_writer_byte_identical_to_reference(exe, tmp_path, mode):
    """LMM::WriteFiles (src/lmm.cpp:101-225): header, column order and number formats of every -lmm mode, on the
    first lines the reference wrote for BXD."""
    src = os.path.join(TXT, "L%d.assoc.head.txt" % mode)
    run(exe, "assoc", src, mode, tmp_path, "W")
    assert open(tmp_path / "W.assoc.txt", "rb").read() == open(src, "rb").read()


def test_oracle_hwe_filter_keeps_the_reference_snp_set(oracle):
    """Pins the oracle's restatement of CalcHWE (src/mathfunc.cpp:546-640) and of the filter order of ReadFile_bed on the
    reference itself: `gemma -hwe 0.05` on tests/golden/text/H.* (a synthetic PLINK set with heterozygotes) keeps 567 of 600
    SNPs -- the same ones the oracle's first pass keeps; the GPU first pass is checked against the oracle in
    tests/test_gpu_parity.py and against this file in tests/test_gpu_workflow_files.py."""
    import filecases as fc
    all_rs, hwe_rs = fc.hwe_reference_sets()
    raw = np.fromfile(os.path.join(TXT, "H.bed"), dtype=np.uint8)[3:].reshape(600, -1)
    fam = [l.split() for l in open(os.path.join(TXT, "H.fam"))]
    ind = np.array([0 if f[5] in ("-9", "NA") else 1 for f in fam], dtype=np.int32)
    G = oracle.bed_decode(raw, len(fam))[:, ind == 1]
    W = np.ones((int(ind.sum()), 1))
    rs = np.array([l.split()[1] for l in open(os.path.join(TXT, "H.bim"))])
    keep_all = oracle.qc_snps_bed(G, W)
    keep_hwe = oracle.qc_snps_bed(G, W, hwe_level=0.05)
    keep_all = keep_all[0] if isinstance(keep_all, tuple) else keep_all
    keep_hwe = keep_hwe[0] if isinstance(keep_hwe, tuple) else keep_hwe
    assert list(rs[np.asarray(keep_all) == 1]) == all_rs
    assert list(rs[np.asarray(keep_hwe) == 1]) == hwe_rs and len(hwe_rs) == 567


def test_text_reader_crlf_across_buffer_boundary(exe, tmp_path):
    """TextFile refills a 1 MiB buffer: a "\r\n" whose two bytes fall either side of the refill must still end ONE line
    (safeGetline's behaviour, src/gemma_io.cpp:118-151), also for "\r" alone at the boundary and for a last line without end."""
    for first_len, ending in ((1048575, "\r\n"), (1048576, "\r\n"), (1048575, "\r"), (1048574, "\n")):
        f = tmp_path / ("p%d_%d.txt" % (first_len, len(ending)))
        with open(f, "w", newline="") as g:
            g.write("7" + "0" * (first_len - 3) + ".5" + ending + "1.5" + ending + "NA" + ending + "2.25")
        out = run(exe, "pheno", f, 1).stdout.strip().split("\n")
        assert len(out) == 4, (first_len, ending, len(out))
        assert [o.split()[0] for o in out] == ["1", "1", "0", "1"]
        assert float(out[1].split()[1]) == 1.5 and float(out[3].split()[1]) == 2.25 and float(out[0].split()[1]) == float("inf")


def test_block_reader_cr_at_the_end_of_a_read(exe, tmp_path):
    """BimbamReader reads the text in chunks: when a chunk ends on the "\r" of a "\r\n" the line is only closed once the
    next chunk shows whether a "\n" follows.  Lines of exactly chunk - 1 bytes put every "\r" there."""
    ni = 100
    rng = np.random.default_rng(12)
    lines, want = [], []
    for k in range(9):
        vals = ["%.3f" % v for v in rng.uniform(0, 2, ni)]
        body = ", A, G, " + ", ".join(vals)
        name = "rs" + "x" * (2047 - 2 - len(body))
        assert len(name + body) == 2047
        lines.append(name + body)
        want.append([float(v) for v in vals])
    f = tmp_path / "cr.txt"
    with open(f, "w", newline="") as g:
        g.write("\r\n".join(lines) + "\r\n")
    out = tmp_path / "cr.bin"
    r = subprocess.run([exe, "geno", str(f), str(ni), "3", "4", str(out)], capture_output=True, text=True,
                       env=dict(os.environ, GEMMA_HIP_IO_TEXT_CAP="2048"))
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out).reshape(-1, ni)
    assert got.shape == (9, ni) and np.array_equal(got, np.array(want))
    assert [l.split()[0] for l in r.stdout.strip().split("\n")] == [l.split(",")[0] for l in lines]


def test_block_prefetch_scenarios(exe):
    """BlockPrefetch (gemma_host.hpp): blocks in order and intact while held, a producer exception rethrown by next(), a
    consumer that stops early (destructor joins a producer that still has work), empty and malformed sources."""
    r = subprocess.run([exe, "prefetch"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "prefetch ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
