"""File-driven runs of tests/cpp/gemma_file_driver.cpp (gemma's own command lines: -g/-p/-c/-a or -bfile, -gk, -k ... -lmm,
-eigen, -d/-u) checked against what the reference binary wrote for the same files (tests/golden/text/, made by
tests/golden/make_text_fixtures.py from oracle/_ref/gemma).  Shared by tests/test_file_driver_cpu.py (the C++ host layer
over the oracle-backed ABI test double, CPU) and tests/test_gpu_workflow_files.py (the same driver over the HIP library)."""
import gzip
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TXT = os.path.join(ROOT, "tests", "golden", "text")
STAT_TOL = 2e-6  # 7 printed digits (5e-7) + the 1e-6 parity bar
LAM_COLS = ("l_remle", "l_mle")


def build_driver(tmp, libdir, extra_rpath=()):
    exe = os.path.join(str(tmp), "gemma_file_driver")
    rp = ["-Wl,-rpath," + d for d in (libdir,) + tuple(extra_rpath)]
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "gemma_file_driver.cpp"), "-L" + libdir, "-lgemma_hip"] + rp +
                          ["-lz", "-pthread", "-o", exe])
    return exe


def drive(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True)
    assert r.returncode == 0, "driver failed: %s\n%s\n%s" % (" ".join(map(str, args)), r.stdout, r.stderr)
    return dict(kv.split("=", 1) for kv in r.stdout.split() if "=" in kv)


def read_assoc(path):
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rt") as f:
        lines = f.read().strip().split("\n")
    hdr = lines[0].split("\t")
    rows = [l.split("\t") for l in lines[1:]]
    return hdr, rows


def compare_assoc(got_path, ref_path, n_ref_rows=None, n_anno=7):
    """Same header, same SNPs in the same order, identical annotation columns, statistics to the printed digits;
    lambda: >= 98 % to the printed digits, all within 1e-3 (the reference reports the penultimate Newton iterate)."""
    gh, gr = read_assoc(got_path)
    rh, rr = read_assoc(ref_path)
    assert gh == rh
    if n_ref_rows is None:
        assert len(gr) == len(rr)
    else:
        assert len(gr) == n_ref_rows
        gr = gr[:len(rr)]
    for g, r in zip(gr, rr):
        assert g[:n_anno] == r[:n_anno], (g[:n_anno], r[:n_anno])
    for j in range(n_anno, len(gh)):
        got = np.array([float(g[j]) for g in gr])
        ref = np.array([float(r[j]) for r in rr])
        both_nan = np.isnan(got) & np.isnan(ref)
        rel = np.where(both_nan, 0.0, np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300))
        if gh[j] in LAM_COLS:
            assert (rel <= 1e-3).all() and (rel <= STAT_TOL).mean() >= 0.98, (gh[j], np.nanmax(rel))
        else:
            flips = np.isnan(got) != np.isnan(ref)  # a failed lambda search is a Newton loop that cycled: <= 0.1 % may flip
            assert flips.mean() <= 1e-3 and (rel[~flips] <= STAT_TOL).all(), (gh[j], np.nanmax(rel[~flips]))


def check_log(kv, log_json):
    log = json.load(open(os.path.join(TXT, log_json)))
    assert int(kv["ni_total"]) == int(log["number of total individuals"])
    assert int(kv["ni_test"]) == int(log["number of analyzed individuals"])
    assert int(kv["n_cvt"]) == int(log["number of covariates"])
    assert int(kv["ns_total"]) == int(log["number of total SNPs/var"])
    assert int(kv["ns_test"]) == int(log["number of analyzed SNPs/var"])
    if "pve" in kv:  # the log prints 6 significant digits
        for key, name in (("pve", "pve estimate in the null model"), ("se_pve", "se(pve) in the null model"),
                          ("vg", "vg estimate in the null model"), ("ve", "ve estimate in the null model"),
                          ("logl_remle_H0", "REMLE log-likelihood in the null model"),
                          ("logl_mle_H0", "MLE log-likelihood in the null model")):
            assert abs(float(kv[key]) - float(log[name])) <= 2e-5 * abs(float(log[name])) + 1e-12, (key, kv[key], log[name])


def check_log_file(path, log_json):
    """<o>.log.txt written by the driver (RunLog, the lines of GEMMA::WriteLog this path owns) against the same lines of the
    reference's log: counts equal, null-model estimates equal at the 6 digits both print; the timer lines the reference's
    tests grep for are present."""
    ref = json.load(open(os.path.join(TXT, log_json)))
    got = {}
    for line in open(path):
        if line.startswith("## ") and "=" in line:
            k, v = line[3:].split("=", 1)
            got[k.strip()] = v.strip()
    for k, v in ref.items():
        assert k in got, k
        if k.startswith("number of"):
            assert got[k] == v, (k, got[k], v)
        else:
            assert got[k] == v or abs(float(got[k]) - float(v)) <= 2e-5 * abs(float(v)), (k, got[k], v)
    assert "total computation time" in got and got["total computation time"].endswith("min")
    assert "device" in got and "Command Line Input" in got


def bxd_bimbam_workflow(exe, out, modes=(1, 4, 9)):
    """BIMBAM text input with covariates and annotation: -gk, -lmm 1/4/9 through the 10-digit cXX hand-off, -eigen,
    then -lmm from the -d/-u artefacts."""
    out = str(out)
    base = ["-g", os.path.join(TXT, "bxd_mean_genotypes.txt.gz"), "-p", os.path.join(TXT, "bxd_trait.txt.gz"),
            "-c", os.path.join(TXT, "bxd_cvt.txt.gz"), "-a", os.path.join(TXT, "bxd_anno.txt.gz"), "-outdir", out]
    kv = drive(exe, *base, "-gk", "-o", "BXD")
    assert (int(kv["ni_total"]), int(kv["ni_test"]), int(kv["ns_total"]), int(kv["ns_test"])) == (198, 67, 7320, 7317)
    cxx = os.path.join(out, "BXD.cXX.txt")
    K = np.loadtxt(cxx)
    ref = np.load(os.path.join(ROOT, "tests", "golden", "ref_bxd.npz"))["cXX"]
    assert K.shape == ref.shape and np.abs(K - ref).max() <= 2e-10  # one unit of the 10th printed digit
    corner = np.loadtxt(os.path.join(TXT, "BXD.cXX.corner.txt"))
    assert np.abs(K[:24, :24] - corner).max() <= 2e-10
    for m in modes:
        kv = drive(exe, *base, "-k", cxx, "-lmm", m, "-maf", "0.1", "-o", "L%d" % m)
        check_log(kv, "L1.log.json")
        check_log_file(os.path.join(out, "L%d.log.txt" % m), "L1.log.json")
        compare_assoc(os.path.join(out, "L%d.assoc.txt" % m), os.path.join(TXT, "L%d.assoc.head.txt" % m), n_ref_rows=7317)
        full = np.load(os.path.join(ROOT, "tests", "golden", "ref_bxd.npz"))
        hdr, rows = read_assoc(os.path.join(out, "L%d.assoc.txt" % m))
        assert [r[1] for r in rows] == list(full["rs"])
        for j, name in enumerate(hdr[7:], start=7):
            got = np.array([float(r[j]) for r in rows])
            want = full["lmm%d_%s" % (m, name)]
            rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-300)
            rel[np.isnan(got) & np.isnan(want)] = 0.0
            flips = np.isnan(got) != np.isnan(want)
            assert flips.mean() <= 1e-3
            if name in LAM_COLS:
                assert (rel[~flips] <= 1e-3).all() and (rel[~flips] <= STAT_TOL).mean() >= 0.98, (m, name)
            else:
                assert (rel[~flips] <= STAT_TOL).all(), (m, name, np.nanmax(rel[~flips]))
    drive(exe, *base, "-k", cxx, "-eigen", "-o", "E")
    D = np.loadtxt(os.path.join(out, "E.eigenD.txt"))
    Dref = np.loadtxt(os.path.join(TXT, "E.eigenD.txt"))
    assert D.shape == Dref.shape and np.allclose(D, Dref, rtol=1e-8, atol=1e-9)
    U = np.loadtxt(os.path.join(out, "E.eigenU.txt"))
    assert np.abs(U.T @ U - np.eye(len(D))).max() < 1e-8
    kv = drive(exe, *base, "-d", os.path.join(out, "E.eigenD.txt"), "-u", os.path.join(out, "E.eigenU.txt"), "-lmm", 1,
               "-maf", "0.1", "-o", "L1du")
    compare_assoc(os.path.join(out, "L1du.assoc.txt"), os.path.join(TXT, "L1.assoc.head.txt"), n_ref_rows=7317)


def plink_workflow(exe, out):
    """PLINK input with missing calls and unphenotyped individuals (.fam phenotypes, -9 = missing): -gk, -lmm 4 with and
    without a covariate file (no intercept column, one NA row), tightened -miss / -maf filters."""
    out = str(out)
    base = ["-bfile", os.path.join(TXT, "P"), "-outdir", out]
    kv = drive(exe, *base, "-gk", "-o", "P")
    check_log(kv, "P4.log.json")
    cxx = os.path.join(out, "P.cXX.txt")
    head = np.loadtxt(os.path.join(TXT, "P.cXX.head.txt"))
    K = np.loadtxt(cxx)
    assert K.shape == (240, 240) and np.abs(K[:8] - head).max() <= 2e-10 and np.abs(K - K.T).max() == 0
    kv = drive(exe, *base, "-k", cxx, "-lmm", 4, "-o", "P4")
    check_log(kv, "P4.log.json")
    compare_assoc(os.path.join(out, "P4.assoc.txt"), os.path.join(TXT, "P4.assoc.txt.gz"))
    kv = drive(exe, *base, "-k", cxx, "-lmm", 4, "-c", os.path.join(TXT, "P.cov.txt"), "-o", "P4c")
    check_log(kv, "P4c.log.json")
    check_log_file(os.path.join(out, "P4c.log.txt"), "P4c.log.json")
    compare_assoc(os.path.join(out, "P4c.assoc.txt"), os.path.join(TXT, "P4c.assoc.txt.gz"))
    kv = drive(exe, *base, "-k", cxx, "-lmm", 1, "-miss", "0.02", "-maf", "0.05", "-o", "P1q")
    check_log(kv, "P1q.log.json")
    compare_assoc(os.path.join(out, "P1q.assoc.txt"), os.path.join(TXT, "P1q.assoc.txt.gz"))


def loco_workflow(exe, out, chrs=(2,), modes=(1,)):
    """`-loco C` on BIMBAM text (src/param.cpp:52-66,497-500): the issue188 genotypes written as a mean-genotype file with
    four chromosomes in the annotation -- the input tests/golden/make_ref_fixtures.py::loco gave the reference binary --
    kinship from the SNPs off chromosome C, association on the SNPs on it, against the reference's cXX rows and .assoc.txt
    values (tests/golden/ref_loco.npz)."""
    out = str(out)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_issue188.npz"))
    ref = np.load(os.path.join(ROOT, "tests", "golden", "ref_loco.npz"))
    n_total = int(fx["n_total"])
    nb = (n_total + 3) // 4
    raw = fx["bed"][3:].reshape(-1, nb)
    codes = np.stack([(raw >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(raw.shape[0], nb * 4)[:, :n_total]
    text = np.array(["2", "NA", "1", "0"])[codes]  # 00 -> 2, 01 -> missing, 10 -> 1, 11 -> 0 (src/lmm.cpp:1797-1812)
    p = text.shape[0]
    with open(os.path.join(out, "g.txt"), "w") as f:
        for t in range(p):
            f.write("rs%d, A, T, %s\n" % (t, ", ".join(text[t])))
    with open(os.path.join(out, "ph.txt"), "w") as f:
        for v in fx["pheno_col6"]:
            f.write(("NA" if v in ("-9", "NA") else str(v)) + "\n")
    with open(os.path.join(out, "anno.txt"), "w") as f:
        for t in range(p):
            f.write("rs%d\t%d\t%d\n" % (t, 1000 + t, ref["chr"][t]))
    base = ["-g", os.path.join(out, "g.txt"), "-p", os.path.join(out, "ph.txt"), "-a", os.path.join(out, "anno.txt"),
            "-outdir", out]
    for c in chrs:
        kv = drive(exe, *base, "-gk", 1, "-loco", c, "-o", "k%d" % c)
        assert int(kv["gwasnps"]) == int((ref["chr"] == c).sum()) and int(kv["ksnps"]) == int((ref["chr"] != c).sum())
        cxx = os.path.join(out, "k%d.cXX.txt" % c)
        K = np.loadtxt(cxx)
        assert np.abs(K[:16] - ref["c%d_cXX_rows" % c]).max() <= 2e-10 and np.abs(np.diag(K) - ref["c%d_cXX_diag" % c]).max() <= 2e-10
        for m in modes:
            tag = "c%d_lmm%d" % (c, m)
            drive(exe, *base, "-k", cxx, "-lmm", m, "-loco", c, "-o", tag)
            hdr, rows = read_assoc(os.path.join(out, tag + ".assoc.txt"))
            idx = ref[tag + "_snp"]
            assert [r[1] for r in rows] == ["rs%d" % t for t in idx] and all(r[0] == str(c) for r in rows)
            for j, name in enumerate(hdr[7:], start=7):
                if tag + "_" + name not in ref.files:
                    continue
                got = np.array([float(r[j]) for r in rows])
                want = ref[tag + "_" + name]
                both = np.isnan(got) & np.isnan(want)
                rel = np.where(both, 0.0, np.abs(got - want) / np.maximum(np.abs(want), 1e-300))
                if name in LAM_COLS:
                    assert (rel <= 1e-3).all() and (rel <= STAT_TOL).mean() >= 0.98, (tag, name)
                else:
                    assert (rel <= STAT_TOL).all(), (tag, name, np.nanmax(rel))


def mvlmm_workflow(exe, out, modes=(1, 4), bimbam=False, crt=False, gxe=False):
    """`-lmm m -n 1 2` (multivariate LMM, class MVLMM) from PLINK files: the reference's test/data/issue243 set (1000
    individuals, 2 traits, first 800 SNPs) rebuilt from tests/golden/ref_mv.npz, -gk then -k ... -lmm, against the
    reference's .assoc.txt columns.  An EM that stops one iteration earlier or later moves the estimates by ~1e-4:
    >= 97 % of the SNPs to the printed digits, all within 5e-3 (the criterion of tests/test_gpu_mvlmm.py).
    crt: the same runs with -crt against tests/golden/ref_mv_crt.npz (the reference with -crt: the SNP that reaches MphNR gets
    PCRT's corrected p_wald / p_lrt; its corrected value must come out, not the uncorrected one).
    gxe: `-gxe env.txt -snps list` on the same files against fixture `g` of tests/golden/ref_mv_wide.npz (the reference's
    MVLMM::AnalyzePlinkGXE, src/mvlmm.cpp:4416-4870, on every 5th SNP)."""
    import refcases as R
    out = str(out)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_mv.npz"))
    fx_plain = fx
    tag, extra = "a", ()
    if gxe:
        fw = np.load(os.path.join(ROOT, "tests", "golden", "ref_mv_wide.npz"))
        fx = dict(fx)
        fx.update({k: fw[k] for k in fw.files if k.startswith("g_")})
        tag = "g"
        np.savetxt(os.path.join(out, "envg.txt"), fw["g_env"], fmt="%.6f")
        with open(os.path.join(out, "snpsg.txt"), "w") as f:
            f.writelines("rs%d\n" % t for t in fw["g_snps_listed"])
        extra = ("-gxe", os.path.join(out, "envg.txt"), "-snps", os.path.join(out, "snpsg.txt"))
    if crt:
        fc = np.load(os.path.join(ROOT, "tests", "golden", "ref_mv_crt.npz"))
        fx = dict(fx)
        fx.update({k.replace("a_crt_", "a_", 1): fc[k] for k in fc.files if k.startswith("a_crt_m")})
    Y = fx["a_pheno"]
    n_total = Y.shape[0]
    nb = (n_total + 3) // 4
    ns = (fx["a_bed"].size - 3) // nb
    pre = os.path.join(out, "mv2")
    open(pre + ".bed", "wb").write(fx["a_bed"].tobytes())
    with open(pre + ".bim", "w") as f:
        for t in range(ns):
            f.write("1\trs%d\t0\t%d\tA\tG\n" % (t, t + 1))
    with open(pre + ".fam", "w") as f:
        for i in range(n_total):
            f.write("f%d i%d 0 0 1 %r %r\n" % (i, i, float(Y[i, 0]), float(Y[i, 1])))
    base = ["-bfile", pre, "-outdir", out]
    if bimbam:  # the same data as a mean-genotype text file: the reference prints the same table for it (checked when
        # this case was written: oracle/_ref/gemma -g ... -n 1 2 gives the PLINK run's numbers digit for digit)
        codes = np.stack([(fx["a_bed"][3:].reshape(-1, nb) >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(ns, nb * 4)[:, :n_total]
        text = np.array(["2", "NA", "1", "0"])[codes]
        with open(os.path.join(out, "g.txt"), "w") as f:
            for t in range(ns):
                f.write("rs%d, A, G, %s\n" % (t, ", ".join(text[t])))
        with open(os.path.join(out, "ph.txt"), "w") as f:
            for i in range(n_total):
                f.write("%r %r\n" % (float(Y[i, 0]), float(Y[i, 1])))
        base = ["-g", os.path.join(out, "g.txt"), "-p", os.path.join(out, "ph.txt"), "-outdir", out]
    drive(exe, *base, "-gk", "-o", "mv2")
    cxx = os.path.join(out, "mv2.cXX.txt")
    for m in modes:
        kv = drive(exe, *base, "-k", cxx, "-lmm", m, "-n", 1, 2, *(("-crt",) if crt else ()), *extra, "-o", "mv2_m%d" % m)
        assert abs(float(kv["logl_remle_H0"]) - fx[tag + "_logl_null"][0]) <= 2e-6 * abs(fx[tag + "_logl_null"][0])
        assert abs(float(kv["logl_mle_H0"]) - fx[tag + "_logl_null"][1]) <= 2e-6 * abs(fx[tag + "_logl_null"][1])
        hdr, rows = read_assoc(os.path.join(out, "mv2_m%d.assoc.txt" % m))
        want_hdr = ["chr", "rs", "ps", "n_miss", "allele1", "allele0", "af", "beta_1", "beta_2", "Vbeta_1_1", "Vbeta_1_2",
                    "Vbeta_2_2"] + {1: ["p_wald"], 2: ["p_lrt"], 3: ["p_score"], 4: ["p_wald", "p_lrt", "p_score"]}[m]
        assert hdr == want_hdr
        assert [r[1] for r in rows] == ["rs%d" % t for t in fx[tag + "_snp"]]
        col = {name: np.array([float(r[j]) for r in rows]) for j, name in enumerate(hdr) if j >= 7}
        got = {"beta": np.column_stack([col["beta_1"], col["beta_2"]]),
               "Vbeta": np.column_stack([col["Vbeta_1_1"], col["Vbeta_1_2"], col["Vbeta_2_2"]])}
        for c in ("p_wald", "p_lrt", "p_score"):
            if c in col:
                got[c] = col[c]
        err = R.mv_row_err(got, R.mv_ref_table(fx, tag, m, 2))
        assert np.mean(err <= STAT_TOL) >= 0.97 and err.max() <= 5e-3, (m, float(np.mean(err <= STAT_TOL)), float(err.max()))
        if crt:  # the rows -crt changes in the reference: the corrected value, to the digits the EM's stopping point allows
            for c in ("p_wald", "p_lrt"):
                key = "a_m%d_%s" % (m, c)
                if c in got:
                    rows_c = np.flatnonzero(fx[key] != fx_plain[key])
                    assert rows_c.size >= 1
                    for r in rows_c:
                        corrected, plain = fx[key][r], fx_plain[key][r]
                        assert abs(got[c][r] - corrected) <= 5e-3 * corrected < abs(got[c][r] - plain), (c, r, got[c][r], corrected, plain)


def perl_checksum(path):
    """the one-liner of test/dev_test_suite.sh:52,68: sum over all fields of sprintf('%.2f', substr(field, 0, 6))"""
    import re
    num = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?")
    tot = 0.0
    for line in open(path):
        for x in line.split():
            m = num.match(x[:6])
            if m:
                tot += float("%.2f" % float(m.group(0)))
    return tot


def lm_workflow(exe, out):
    """`-lm m` (class LM: no kinship) from BIMBAM text and from PLINK files.  BXD `-lm 4`: the reference's OWN golden for this
    run -- 95134 words, field checksum 3089042886 (test/dev_test_suite.sh:60-68) -- on the file this driver writes, and every
    column against the reference binary's output; PLINK subset with covariates against its `-lm 4` file."""
    out = str(out)
    base = ["-g", os.path.join(TXT, "bxd_mean_genotypes.txt.gz"), "-p", os.path.join(TXT, "bxd_trait.txt.gz"),
            "-c", os.path.join(TXT, "bxd_cvt.txt.gz"), "-a", os.path.join(TXT, "bxd_anno.txt.gz"), "-outdir", out]
    full = np.load(os.path.join(ROOT, "tests", "golden", "ref_bxd.npz"))
    for m in (1, 2, 3, 4):
        drive(exe, *base, "-lm", m, "-maf", "0.1", "-o", "LM%d" % m)
        path = os.path.join(out, "LM%d.assoc.txt" % m)
        hdr, rows = read_assoc(path)
        assert hdr[:8] == ["chr", "rs", "ps", "n_mis", "n_obs", "allele1", "allele0", "af"]
        assert [r[1] for r in rows] == list(full["rs"])
        for j, name in enumerate(hdr[8:], start=8):
            got = np.array([float(r[j]) for r in rows])
            want = full["lm%d_%s" % (m, name)]
            assert (np.abs(got - want) <= STAT_TOL * np.abs(want)).all(), (m, name)
        if m == 4:
            assert len(open(path).read().split()) == 95134
            assert "%.0f" % perl_checksum(path) == "3089042886"
    kv = drive(exe, "-bfile", os.path.join(TXT, "P"), "-outdir", out, "-lm", 4, "-c", os.path.join(TXT, "P.cov.txt"), "-o", "Plm4c")
    check_log(kv, "Plm4c.log.json")
    compare_assoc(os.path.join(out, "Plm4c.assoc.txt"), os.path.join(TXT, "Plm4c.assoc.txt.gz"), n_anno=8)


def gxe_workflow(exe, out, modes=(1,)):
    """`-gxe env.txt` on PLINK files (LMM::AnalyzePlinkGXE): the issue188 set rebuilt from tests/golden/ref_issue188.npz
    (1008 individuals, 132 without phenotype, missing calls), -gk, then -k ... -lmm m -gxe, against the reference's output."""
    out = str(out)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_issue188.npz"))
    n_total = int(fx["n_total"])
    nb = (n_total + 3) // 4
    ns = (fx["bed"].size - 3) // nb
    pre = os.path.join(out, "p188")
    open(pre + ".bed", "wb").write(fx["bed"].tobytes())
    with open(pre + ".bim", "w") as f:
        for t in range(ns):
            f.write("1\trs%d\t0\t%d\tA\tG\n" % (t, t + 1))
    with open(pre + ".fam", "w") as f:
        for i, v in enumerate(fx["pheno_col6"]):
            f.write("f%d i%d 0 0 1 %s\n" % (i, i, v))
    np.savetxt(os.path.join(out, "env.txt"), fx["env"], fmt="%.10g")
    base = ["-bfile", pre, "-outdir", out]
    drive(exe, *base, "-gk", "-o", "k1")
    cxx = os.path.join(out, "k1.cXX.txt")
    K = np.loadtxt(cxx)
    assert np.abs(K[:24] - fx["cXX_rows"]).max() <= 2e-10 and np.abs(np.diag(K) - fx["cXX_diag"]).max() <= 2e-10
    for m in modes:
        tag = "gxe%d" % m
        drive(exe, *base, "-k", cxx, "-lmm", m, "-gxe", os.path.join(out, "env.txt"), "-o", tag)
        hdr, rows = read_assoc(os.path.join(out, tag + ".assoc.txt"))
        assert [r[1] for r in rows] == ["rs%d" % t for t in fx[tag + "_snp"]]
        assert np.array_equal(np.array([float(r[3]) for r in rows]), fx[tag + "_n_miss"])
        assert np.array_equal(np.array([float(r[6]) for r in rows]), fx[tag + "_af"])
        for j, name in enumerate(hdr[7:], start=7):
            got = np.array([float(r[j]) for r in rows])
            want = fx[tag + "_" + name]
            both = np.isnan(got) & np.isnan(want)
            rel = np.where(both, 0.0, np.abs(got - want) / np.maximum(np.abs(want), 1e-300))
            if name in LAM_COLS:
                assert (rel <= 1e-3).all() and (rel <= STAT_TOL).mean() >= 0.98, (tag, name)
            else:
                assert (rel <= STAT_TOL).all(), (tag, name, np.nanmax(rel))


def gene_workflow(exe, out, modes=(1,)):
    """`-gene expr.txt -p pheno -k K -lmm m` (LMM::AnalyzeGene from the file): 40 expression rows over the issue188
    individuals (tests/golden/ref_gene.npz), kinship from the PLINK set, against the reference's geneID table."""
    out = str(out)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_issue188.npz"))
    ge = np.load(os.path.join(ROOT, "tests", "golden", "ref_gene.npz"))
    n_total = int(fx["n_total"])
    nb = (n_total + 3) // 4
    ns = (fx["bed"].size - 3) // nb
    pre = os.path.join(out, "pg")
    open(pre + ".bed", "wb").write(fx["bed"].tobytes())
    with open(pre + ".bim", "w") as f:
        for t in range(ns):
            f.write("1\trs%d\t0\t%d\tA\tG\n" % (t, t + 1))
    with open(pre + ".fam", "w") as f:
        for i, v in enumerate(fx["pheno_col6"]):
            f.write("f%d i%d 0 0 1 %s\n" % (i, i, v))
    with open(os.path.join(out, "gene.txt"), "w") as f:
        f.write("id\t" + "\t".join("i%d" % i for i in range(n_total)) + "\n")
        for r in range(ge["expr"].shape[0]):
            f.write("g%d\t" % r + "\t".join("%.10g" % v for v in ge["expr"][r]) + "\n")
    with open(os.path.join(out, "gph.txt"), "w") as f:
        for v in fx["pheno_col6"]:
            f.write(("NA" if v in ("-9", "NA") else str(v)) + "\n")
    drive(exe, "-bfile", pre, "-outdir", out, "-gk", "-o", "kg")
    cxx = os.path.join(out, "kg.cXX.txt")
    for m in modes:
        drive(exe, "-gene", os.path.join(out, "gene.txt"), "-p", os.path.join(out, "gph.txt"), "-k", cxx, "-lmm", m,
              "-outdir", out, "-o", "ge%d" % m)
        hdr, rows = read_assoc(os.path.join(out, "ge%d.assoc.txt" % m))
        assert hdr[0] == "geneID" and [r[0] for r in rows] == ["g%d" % r for r in range(ge["expr"].shape[0])]
        for j, name in enumerate(hdr[1:], start=1):
            got = np.array([float(r[j]) for r in rows])
            want = ge["lmm%d_%s" % (m, name)]
            rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-300)
            if name in LAM_COLS:
                assert (rel <= 1e-3).all() and (rel <= STAT_TOL).mean() >= 0.9, (m, name)
            else:
                assert (rel <= STAT_TOL).all(), (m, name, float(rel.max()))


def selection_options_workflow(exe, out):
    """The switches that change WHICH SNPs / kinship entries are used: `-snps list` (BIMBAM: unlisted SNPs stay in the file
    order with indicator 0), `-notsnp` (no maf filter) and `-km 2` (kinship as id-pair triples over the .fam ids), each
    against the reference binary's output for the same command."""
    out = str(out)
    base = ["-g", os.path.join(TXT, "bxd_mean_genotypes.txt.gz"), "-p", os.path.join(TXT, "bxd_trait.txt.gz"),
            "-c", os.path.join(TXT, "bxd_cvt.txt.gz"), "-a", os.path.join(TXT, "bxd_anno.txt.gz"), "-outdir", out]
    drive(exe, *base, "-gk", "-o", "BXD")
    kv = drive(exe, *base, "-k", os.path.join(out, "BXD.cXX.txt"), "-lmm", 1, "-maf", "0.1", "-snps",
               os.path.join(TXT, "bxd_snps7.txt"), "-o", "Ls")
    assert int(kv["ns_total"]) == 7320 and int(kv["ns_test"]) == 1045
    compare_assoc(os.path.join(out, "Ls.assoc.txt"), os.path.join(TXT, "Ls.assoc.txt.gz"))
    pb = ["-bfile", os.path.join(TXT, "P"), "-outdir", out]
    drive(exe, *pb, "-gk", "-o", "P")
    cxx = os.path.join(out, "P.cXX.txt")
    kv = drive(exe, *pb, "-k", cxx, "-lmm", 1, "-notsnp", "-o", "P1n")
    check_log(kv, "P1n.log.json")
    compare_assoc(os.path.join(out, "P1n.assoc.txt"), os.path.join(TXT, "P1n.assoc.txt.gz"))
    ids = [l.split()[1] for l in open(os.path.join(TXT, "P.fam"))]
    toks = [l.rstrip("\n").split("\t") for l in open(cxx)]
    with open(os.path.join(out, "P.km2.txt"), "w") as f:
        for i in range(len(ids)):
            for j in range(i, len(ids)):
                f.write("%s\t%s\t%s\n" % (ids[i], ids[j], toks[i][j]))
    kv = drive(exe, *pb, "-k", os.path.join(out, "P.km2.txt"), "-km", 2, "-lmm", 1, "-o", "P1km2")
    check_log(kv, "P1km2.log.json")
    compare_assoc(os.path.join(out, "P1km2.assoc.txt"), os.path.join(TXT, "P1km2.assoc.txt.gz"))


def hwe_reference_sets():
    """(rs kept without, rs kept with `-hwe 0.05`) as the reference binary printed them for tests/golden/text/H.*"""
    _, all_rows = read_assoc(os.path.join(TXT, "Hall.assoc.txt.gz"))
    _, hwe_rows = read_assoc(os.path.join(TXT, "Hhwe.assoc.txt.gz"))
    return [r[1] for r in all_rows], [r[1] for r in hwe_rows]


def hwe_workflow(exe, out):
    """`-hwe 0.05` on a PLINK set with heterozygotes: the exact test of CalcHWE (src/mathfunc.cpp:546-640) runs in the device's
    first pass; same surviving SNPs and statistics as the reference."""
    out = str(out)
    pb = ["-bfile", os.path.join(TXT, "H"), "-outdir", out]
    drive(exe, *pb, "-gk", "-o", "H")
    cxx = os.path.join(out, "H.cXX.txt")
    kv = drive(exe, *pb, "-k", cxx, "-lmm", 1, "-hwe", "0.05", "-o", "Hhwe")
    all_rs, hwe_rs = hwe_reference_sets()
    assert int(kv["ns_test"]) == len(hwe_rs) < len(all_rs)
    compare_assoc(os.path.join(out, "Hhwe.assoc.txt"), os.path.join(TXT, "Hhwe.assoc.txt.gz"))


def mvlmm3_workflow(exe, out, modes=(1, 3), crt=False):
    """Three traits with missing phenotypes (fixture `b` of ref_mv.npz: issue188 genotypes, simulated correlated traits, 25 NA
    entries): the `-gk` run selects individuals by trait 1 alone, the `-lmm m -n 1 2 3` run by all three -- as the reference
    does; REML and score modes (the reference's ML EM for d >= 3 is basis-unstable).
    crt: with -crt against tests/golden/ref_mv_crt.npz -- the reference corrects the p_wald of 52 SNPs here."""
    import refcases as R
    out = str(out)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_mv.npz"))
    fx_plain = fx
    if crt:
        fc = np.load(os.path.join(ROOT, "tests", "golden", "ref_mv_crt.npz"))
        fx = dict(fx)
        fx.update({k.replace("b_crt_", "b_", 1): fc[k] for k in fc.files if k.startswith("b_crt_m")})
    f188 = np.load(os.path.join(ROOT, "tests", "golden", "ref_issue188.npz"))
    txt = fx["b_pheno_txt"]
    n_total = txt.shape[0]
    nb = (n_total + 3) // 4
    ns = (f188["bed"].size - 3) // nb
    pre = os.path.join(out, "mv3")
    open(pre + ".bed", "wb").write(f188["bed"].tobytes())
    with open(pre + ".bim", "w") as f:
        for t in range(ns):
            f.write("1\trs%d\t0\t%d\tA\tG\n" % (t, t + 1))
    with open(pre + ".fam", "w") as f:
        for i in range(n_total):
            f.write("f%d i%d 0 0 1 %s\n" % (i, i, " ".join(txt[i])))
    base = ["-bfile", pre, "-outdir", out]
    drive(exe, *base, "-gk", "-o", "mv3")
    cxx = os.path.join(out, "mv3.cXX.txt")
    for m in modes:
        kv = drive(exe, *base, "-k", cxx, "-lmm", m, "-n", 1, 2, 3, *(("-crt",) if crt else ()), "-o", "mv3_m%d" % m)
        assert int(kv["ni_test"]) == int((np.array([["NA" in r for r in txt]]) == 0).sum())
        assert abs(float(kv["logl_remle_H0"]) - fx["b_logl_null"][0]) <= 2e-6 * abs(fx["b_logl_null"][0])
        hdr, rows = read_assoc(os.path.join(out, "mv3_m%d.assoc.txt" % m))
        assert [r[1] for r in rows] == ["rs%d" % t for t in fx["b_snp"]]
        col = {name: np.array([float(r[j]) for r in rows]) for j, name in enumerate(hdr) if j >= 7}
        got = {"beta": np.column_stack([col["beta_%d" % (i + 1)] for i in range(3)]),
               "Vbeta": np.column_stack([col["Vbeta_%d_%d" % (i + 1, j + 1)] for i in range(3) for j in range(i, 3)])}
        for c in ("p_wald", "p_lrt", "p_score"):
            if c in col:
                got[c] = col[c]
        err = R.mv_row_err(got, R.mv_ref_table(fx, "b", m, 3))
        assert np.mean(err <= STAT_TOL) >= 0.97 and err.max() <= 5e-3, (m, float(np.mean(err <= STAT_TOL)), float(err.max()))
        if crt and "p_wald" in got:  # the rows the reference's -crt changes: the corrected value, and not the uncorrected one
            key = "b_m%d_p_wald" % m
            rows_c = np.flatnonzero(fx[key] != fx_plain[key])
            assert rows_c.size >= 40
            dc = np.abs(got["p_wald"][rows_c] - fx[key][rows_c]) / fx[key][rows_c]
            dp = np.abs(got["p_wald"][rows_c] - fx_plain[key][rows_c]) / fx_plain[key][rows_c]
            assert dc.max() <= 5e-3 and np.mean(dc <= STAT_TOL) >= 0.9 and np.all(dc < dp), (float(dc.max()), float(np.mean(dc <= STAT_TOL)))


def standardised_kinship_workflow(exe, out):
    """`-gk 2` from BIMBAM text (rows scaled by 1/sqrt(var), src/gemma_io.cpp:1535-1538) against the reference's sXX.txt"""
    out = str(out)
    base = ["-g", os.path.join(TXT, "bxd_mean_genotypes.txt.gz"), "-p", os.path.join(TXT, "bxd_trait.txt.gz"),
            "-c", os.path.join(TXT, "bxd_cvt.txt.gz"), "-a", os.path.join(TXT, "bxd_anno.txt.gz"), "-outdir", out]
    drive(exe, *base, "-gk", 2, "-o", "BXD2")
    S = np.loadtxt(os.path.join(out, "BXD2.sXX.txt"))
    assert S.shape == (198, 198) and np.abs(S[:24, :24] - np.loadtxt(os.path.join(TXT, "BXD.sXX.corner.txt"))).max() <= 2e-10


def sharded_inproc_workflow(exe, out, world=2, samegpu=False):
    """SURVEY 8e through the C++ host layer with the collectives in: `-inproc 1 -lmm m -gpus N` = every rank accumulates
    the kinship of ITS share of the SNPs, one all-reduce (gemma_hip_kin_end_keep), rank 0 alone runs the eigensolver on the
    kept K, one broadcast of (U, eval) (gemma_hip_kept_bcast), every rank analyses its SNP share; `-k K -lmm m -gpus N` =
    rank 0 alone reads the kinship file and decomposes, one broadcast.  Against the single-process `-inproc` run (the
    kinship sums are added in another order: statistics to the printed digits, not byte for byte); the kinship-file route
    also against the reference's own file."""
    out = str(out)
    same = ["-samegpu"] if samegpu else []
    base = ["-bfile", os.path.join(TXT, "P"), "-outdir", out]
    kv1 = drive(exe, *base, "-inproc", 1, "-lmm", 4, "-o", "in1")
    kvN = drive(exe, *base, "-inproc", 1, "-lmm", 4, "-gpus", world, *same, "-o", "inN")
    assert int(kvN["ranks"]) == world and int(kv1["snps"]) > 0
    compare_assoc(os.path.join(out, "inN.assoc.txt"), os.path.join(out, "in1.assoc.txt"))
    # mode 1 on PLINK input: the shard of rank > 0 seeds AnalyzePlink's beta / se carry (LMM::seed_plink_carry)
    drive(exe, *base, "-inproc", 1, "-lmm", 1, "-o", "w1")
    drive(exe, *base, "-inproc", 1, "-lmm", 1, "-gpus", world, *same, "-o", "wN")
    compare_assoc(os.path.join(out, "wN.assoc.txt"), os.path.join(out, "w1.assoc.txt"))
    # the kinship-file route: rank 0 reads and decomposes, (U, eval) are broadcast -- same file as one process, byte for byte
    drive(exe, *base, "-gk", "-o", "P")
    cxx = os.path.join(out, "P.cXX.txt")
    drive(exe, *base, "-k", cxx, "-lmm", 4, "-o", "k1")
    drive(exe, *base, "-k", cxx, "-lmm", 4, "-gpus", world, *same, "-o", "kN")
    assert open(os.path.join(out, "kN.assoc.txt"), "rb").read() == open(os.path.join(out, "k1.assoc.txt"), "rb").read()
    compare_assoc(os.path.join(out, "kN.assoc.txt"), os.path.join(TXT, "P4.assoc.txt.gz"))  # and the reference's own file
    assert not [f for f in os.listdir(out) if ".rank" in f]
