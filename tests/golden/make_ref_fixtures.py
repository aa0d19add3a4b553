"""Generates tests/golden/ref_*.npz: outputs of the REFERENCE ITSELF (oracle/_ref/gemma = /root/reference/src/*.cpp
compiled unchanged against oracle/gslshim, see oracle/Makefile `ref`) on the datasets that ship in the reference tree,
together with the inputs a test needs to redo the run without /root/reference (the GPU box never sees it).

Run in the build container:  python tests/golden/make_ref_fixtures.py

Before anything is stored the binary has to reproduce the reference's own golden values (test/dev_tests.rb:26-55,
test/dev_test_suite.sh:40-118): BXD kinship checksum -116 / 198 lines, BXD -lm 4 word count 95134 and checksum
3089042886, BXD -lmm 2 p_lrt 1.234747e-01 / max 9.997119e-01 / 73180 words, BXD -lmm 9 max l_mle 0.7531109 / 80498
words, issue188 kinship checksum 194.  That is what validates the GSL shim under it.

Fixtures:
  ref_bxd.npz       BIMBAM + covariates (c = 3): cXX, -lmm 1/2/3/4/9 -maf 0.1, -lm 1..4
  ref_issue188.npz  PLINK with missing calls and 132 unphenotyped individuals: -gk 1/2, -lmm 1..4, -lmm 4 with
                    covariates, -lmm 1 on sXX, -lm 4, -gxe
  ref_loco.npz      -loco (BIMBAM only in the reference: PlinkKin ignores it, src/param.cpp:1307): issue188 re-written as a
                    mean-genotype file with four synthetic chromosomes; -gk 1 -loco c and -lmm 1/4 -loco c for c = 2, 4
  ref_gene.npz      -gene (LMM::AnalyzeGene): 40 simulated expression rows over issue188's individuals, -lmm 1 and 4
  ref_mv.npz        multivariate LMM: issue243 (first 800 SNPs, 2 traits) and issue188 genotypes with 3 simulated
                    traits, -lmm 1..4 (-n 1 2 [3])
  ref_mv_crt.npz    fixture (a) of ref_mv.npz with -crt (run `make_ref_fixtures.py mv_crt`; not part of the default list)
  ref_mv_wide.npz   multivariate -gxe (2 traits) and six traits with three covariates (`make_ref_fixtures.py mv_wide`)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GEMMA = os.path.join(ROOT, "oracle", "_ref", "gemma")
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
STAT_COLS = ("beta", "se", "logl_H1", "l_remle", "l_mle", "p_wald", "p_lrt", "p_score")


def gemma(tmp, *args):
    r = subprocess.run([GEMMA] + [str(a) for a in args], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("reference failed: %s\n%s" % (" ".join(map(str, args)), r.stdout.decode()[-2000:]))


def read_assoc(path):
    with open(path) as f:
        hdr = f.readline().split()
        rows = [l.split() for l in f if l.strip()]
    out = {"rs": np.array([r[hdr.index("rs")] for r in rows])}
    for j, h in enumerate(hdr):
        if h in ("chr", "rs", "allele1", "allele0"):
            continue
        out[h] = np.array([float(r[j]) for r in rows])
    return out, len(hdr) * (len(rows) + 1)


def read_log(path):
    """`## key = value` scalars and the small matrices mvLMM prints under `## ...:` headings."""
    sc, mats, cur = {}, {}, None
    for line in open(path):
        s = line.rstrip("\n")
        if s.startswith("## ") and "=" in s and not s.rstrip().endswith(":"):
            k, v = s[3:].split("=", 1)
            try:
                sc[k.strip()] = float(v.split()[0])
            except (ValueError, IndexError):
                pass
            cur = None
        elif s.startswith("## ") and s.rstrip().endswith(":"):
            cur = s[3:].rstrip().rstrip(":").strip()
            mats[cur] = []
        elif cur is not None and s and not s.startswith("#"):
            mats[cur].extend(float(x) for x in s.split())
    return sc, {k: np.array(v) for k, v in mats.items() if v}


def checksum(path):
    """perl one-liner of test/dev_test_suite.sh:52: sum of sprintf('%.2f', substr(field, 0, 6))."""
    tot = 0.0
    num = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?")  # perl's numeric prefix of a string
    for line in open(path):
        for x in line.split():
            m = num.match(x[:6])
            if m:
                tot += float("%.2f" % float(m.group(0)))
    return tot


def read_matrix(path):
    return np.array([[float(x) for x in l.split()] for l in open(path) if l.strip()])


def stats_of(a, prefix, dst):
    for c in STAT_COLS:
        if c in a:
            dst["%s_%s" % (prefix, c)] = a[c]


def bxd(tmp):
    E = REF + "/example/"
    base = ["-g", E + "BXD_geno.txt.gz", "-p", E + "BXD_pheno.txt", "-c", E + "BXD_covariates2.txt", "-a", E + "BXD_snps.txt"]
    gemma(tmp, *base, "-gk", "-o", "BXD")
    cxx = os.path.join(tmp, "output", "BXD.cXX.txt")
    assert len(open(cxx).readlines()) == 198 and "%.0f" % checksum(cxx) == "-116"
    d = {"cXX": read_matrix(cxx)}
    for m in (1, 2, 3, 4, 9):
        gemma(tmp, *base, "-k", cxx, "-lmm", m, "-no-check", "-maf", "0.1", "-o", "L%d" % m)
        a, words = read_assoc(os.path.join(tmp, "output", "L%d.assoc.txt" % m))
        if m == 2:
            assert words == 73180 and "%.6e" % a["p_lrt"][0] == "1.234747e-01" and "%.6e" % a["p_lrt"].max() == "9.997119e-01"
        if m == 9:
            assert words == 80498 and "%.7g" % a["l_mle"].max() == "0.7531109" and "%.6e" % a["p_lrt"].max() == "9.997119e-01"
        stats_of(a, "lmm%d" % m, d)
        d["rs"] = a["rs"]
        d["af"] = a["af"]
        if m == 1:
            sc, _ = read_log(os.path.join(tmp, "output", "L1.log.txt"))
            d["null"] = np.array([sc["pve estimate in the null model"], sc["se(pve) in the null model"],
                                  sc["vg estimate in the null model"], sc["ve estimate in the null model"],
                                  sc["REMLE log-likelihood in the null model"], sc["MLE log-likelihood in the null model"]])
    for m in (1, 2, 3, 4):
        gemma(tmp, *base, "-lm", m, "-maf", "0.1", "-o", "LM%d" % m)
        f = os.path.join(tmp, "output", "LM%d.assoc.txt" % m)
        a, words = read_assoc(f)
        if m == 4:
            assert words == 95134 and "%.0f" % checksum(f) == "3089042886"  # dev_test_suite.sh:67-68
        stats_of(a, "lm%d" % m, d)
    np.savez_compressed(os.path.join(OUT, "ref_bxd.npz"), **d)
    print("ref_bxd.npz:", len(d["rs"]), "SNPs")


def copy_plink(src_prefix, dst_prefix, fam_lines=None, n_snps=None):
    fam = [l for l in open(src_prefix + ".fam") if l.strip()]
    bim = [l for l in open(src_prefix + ".bim") if l.strip()]
    raw = np.fromfile(src_prefix + ".bed", dtype=np.uint8)
    nb = (len(fam) + 3) // 4
    if n_snps is not None:
        bim = bim[:n_snps]
        raw = raw[: 3 + n_snps * nb]
    open(dst_prefix + ".bed", "wb").write(raw.tobytes())
    open(dst_prefix + ".bim", "w").writelines(bim)
    open(dst_prefix + ".fam", "w").writelines(fam if fam_lines is None else fam_lines)
    return raw, fam, bim


def rs_index(bim, rs):
    pos = {l.split()[1]: i for i, l in enumerate(bim)}
    return np.array([pos[r] for r in rs], dtype=np.int64)


def issue188(tmp):
    src = REF + "/test/data/issue188/2000"
    raw, fam, bim = copy_plink(src, os.path.join(tmp, "p188"))
    n_total = len(fam)
    rng = np.random.default_rng(188)
    cov = np.column_stack([np.ones(n_total), rng.standard_normal(n_total), rng.integers(0, 2, n_total).astype(float)])
    env = rng.standard_normal(n_total)
    np.savetxt(os.path.join(tmp, "cov.txt"), cov, fmt="%.10g")
    np.savetxt(os.path.join(tmp, "env.txt"), env, fmt="%.10g")
    d = {"bed": raw, "n_total": np.array(n_total), "pheno_col6": np.array([l.split()[5] for l in fam]),
         "cov": np.loadtxt(os.path.join(tmp, "cov.txt")), "env": np.loadtxt(os.path.join(tmp, "env.txt"))}
    gemma(tmp, "-bfile", "p188", "-gk", 1, "-o", "k1")
    gemma(tmp, "-bfile", "p188", "-gk", 2, "-o", "k2")
    cxx, sxx = os.path.join(tmp, "output", "k1.cXX.txt"), os.path.join(tmp, "output", "k2.sXX.txt")
    assert "%.0f" % checksum(cxx) == "194"  # dev_test_suite.sh:110
    for tag, f in (("cXX", cxx), ("sXX", sxx)):
        K = read_matrix(f)
        d[tag + "_rows"] = K[:24]  # first rows + diagonal + a checksum are enough to pin the kinship; K itself is 8 MB
        d[tag + "_diag"] = np.diag(K).copy()
        d[tag + "_sum"] = np.array(K.sum())
    runs = [("lmm%d" % m, ["-k", cxx, "-lmm", m]) for m in (1, 2, 3, 4)]
    runs += [("lmm4cov", ["-k", cxx, "-lmm", 4, "-c", "cov.txt"]), ("lmm1sxx", ["-k", sxx, "-lmm", 1]),
             ("lm4", ["-lm", 4]), ("lm1cov", ["-lm", 1, "-c", "cov.txt"]), ("gxe1", ["-k", cxx, "-lmm", 1, "-gxe", "env.txt"]),
             ("gxe4", ["-k", cxx, "-lmm", 4, "-gxe", "env.txt"])]
    for tag, args in runs:
        gemma(tmp, "-bfile", "p188", *args, "-o", tag)
        a, _ = read_assoc(os.path.join(tmp, "output", tag + ".assoc.txt"))
        stats_of(a, tag, d)
        d[tag + "_snp"] = rs_index(bim, a["rs"])
        for extra in ("n_miss", "af"):
            if extra in a:
                d[tag + "_" + extra] = a[extra]
        if tag in ("lmm1", "lmm4cov"):
            sc, _ = read_log(os.path.join(tmp, "output", tag + ".log.txt"))
            d[tag + "_null"] = np.array([sc["pve estimate in the null model"], sc["se(pve) in the null model"],
                                         sc["vg estimate in the null model"], sc["ve estimate in the null model"],
                                         sc["REMLE log-likelihood in the null model"], sc["MLE log-likelihood in the null model"]])
            d[tag + "_counts"] = np.array([sc["number of analyzed individuals"], sc["number of analyzed SNPs/var"]])
    np.savez_compressed(os.path.join(OUT, "ref_issue188.npz"), **d)
    print("ref_issue188.npz:", {k: v.shape for k, v in d.items() if k.endswith("_snp")})
    return raw, fam, bim


def loco(tmp, raw188, fam188, bim188):
    sys.path.insert(0, ROOT)
    from oracle import oracle as O  # decoding the .bed into the text format only (no statistic involved)
    n_total = len(fam188)
    nb = (n_total + 3) // 4
    G = O.bed_decode(np.ascontiguousarray(raw188[3:].reshape(-1, nb)), n_total)
    p = G.shape[0]
    chrs = 1 + (np.arange(p) * 4) // p
    rs = [l.split()[1] for l in bim188]
    with open(os.path.join(tmp, "g.txt"), "w") as f:
        for t in range(p):
            f.write("%s, A, T, %s\n" % (rs[t], ", ".join("NA" if np.isnan(v) else "%g" % v for v in G[t])))
    with open(os.path.join(tmp, "ph.txt"), "w") as f:
        for l in fam188:
            v = l.split()[5]
            f.write(("NA" if v in ("-9", "NA") else v) + "\n")
    with open(os.path.join(tmp, "anno.txt"), "w") as f:
        for t in range(p):
            f.write("%s\t%d\t%d\n" % (rs[t], 1000 + t, chrs[t]))
    d = {"chr": chrs.astype(np.int32)}
    base = ["-g", "g.txt", "-p", "ph.txt", "-a", "anno.txt"]
    for c in (2, 4):
        gemma(tmp, *base, "-gk", 1, "-loco", c, "-o", "k%d" % c)
        K = read_matrix(os.path.join(tmp, "output", "k%d.cXX.txt" % c))
        d["c%d_cXX_rows" % c] = K[:16]
        d["c%d_cXX_diag" % c] = np.diag(K).copy()
        for m in (1, 4):
            tag = "c%d_lmm%d" % (c, m)
            gemma(tmp, *base, "-k", os.path.join(tmp, "output", "k%d.cXX.txt" % c), "-lmm", m, "-loco", c, "-o", tag)
            a, _ = read_assoc(os.path.join(tmp, "output", tag + ".assoc.txt"))
            stats_of(a, tag, d)
            d[tag + "_snp"] = rs_index(bim188, a["rs"])
    np.savez_compressed(os.path.join(OUT, "ref_loco.npz"), **d)
    print("ref_loco.npz:", {k: v.shape for k, v in d.items() if k.endswith("_snp")})


def gene(tmp, raw188, fam188, bim188):
    """LMM::AnalyzeGene (src/lmm.cpp:1365-1471): every row of the expression file is a phenotype, the -p phenotype is the
    tested variable."""
    n_total = len(fam188)
    nb = (n_total + 3) // 4
    codes = np.unpackbits(raw188[3:].reshape(-1, nb)[:300], axis=1, bitorder="little").reshape(300, -1, 2)[:, :n_total]
    g = (codes[:, :, 0] + codes[:, :, 1]).astype(float)
    g = (g - g.mean(1, keepdims=True)) / (g.std(1, keepdims=True) + 1e-9)
    rng = np.random.default_rng(1365)
    h2 = rng.uniform(0.0, 0.8, 40)
    E = np.sqrt(h2)[:, None] * (rng.standard_normal((40, 300)) @ g / np.sqrt(300)) + np.sqrt(1 - h2)[:, None] * rng.standard_normal((40, n_total))
    ph = np.array([(np.nan if l.split()[5] in ("-9", "NA") else float(l.split()[5])) for l in fam188])
    E[:8] += 0.02 * np.nan_to_num(ph)[None, :] * rng.standard_normal((8, 1))  # a few rows that depend on the tested variable
    with open(os.path.join(tmp, "gene.txt"), "w") as f:
        f.write("id\t" + "\t".join("i%d" % i for i in range(n_total)) + "\n")
        for r in range(40):
            f.write("g%d\t" % r + "\t".join("%.10g" % v for v in E[r]) + "\n")
    with open(os.path.join(tmp, "gph.txt"), "w") as f:
        f.writelines(("NA" if np.isnan(v) else "%.10g" % v) + "\n" for v in ph)
    E = np.array([[float("%.10g" % v) for v in row] for row in E])
    d = {"expr": E}
    copy_plink(REF + "/test/data/issue188/2000", os.path.join(tmp, "pg"))
    gemma(tmp, "-bfile", "pg", "-gk", 1, "-o", "kg")
    for m in (1, 4):
        gemma(tmp, "-gene", "gene.txt", "-p", "gph.txt", "-k", os.path.join(tmp, "output", "kg.cXX.txt"), "-lmm", m, "-o", "ge%d" % m)
        with open(os.path.join(tmp, "output", "ge%d.assoc.txt" % m)) as f:
            hdr = f.readline().split()
            rows = [l.split() for l in f if l.strip()]
        assert len(rows) == 40
        for j, h in enumerate(hdr[1:], 1):
            d["lmm%d_%s" % (m, h)] = np.array([float(r[j]) for r in rows])
    np.savez_compressed(os.path.join(OUT, "ref_gene.npz"), **d)
    print("ref_gene.npz: 40 rows; interior lambda on", int((d["lmm1_l_remle"] > 2e-5).sum()))


MV_COLS_EXTRA = ("p_wald", "p_lrt", "p_score")


def mv_run(tmp, prefix, tag, kfile, traits, d, bim, modes=(1, 2, 3, 4), extra=()):
    for m in modes:
        gemma(tmp, "-bfile", prefix, "-k", kfile, "-lmm", m, "-n", *traits, *extra, "-o", "%s_m%d" % (tag, m))
        a, _ = read_assoc(os.path.join(tmp, "output", "%s_m%d.assoc.txt" % (tag, m)))
        for c, v in a.items():
            if c.startswith("beta_") or c.startswith("Vbeta_") or c in MV_COLS_EXTRA:
                d["%s_m%d_%s" % (tag, m, c)] = v
        d["%s_snp" % tag] = rs_index(bim, a["rs"])
        if m == modes[-1]:
            sc, mats = read_log(os.path.join(tmp, "output", "%s_m%d.log.txt" % (tag, m)))
            d[tag + "_logl_null"] = np.array([sc["REMLE log-likelihood in the null model"], sc["MLE log-likelihood in the null model"]])
            for k, v in mats.items():
                key = k.replace(" ", "_").replace("(", "").replace(")", "").replace(",", "")
                d["%s_log_%s" % (tag, key[:60])] = v


def mv(tmp, raw188, fam188, bim188):
    d = {}
    # (a) issue243: 1000 individuals, 2 traits, first 800 SNPs
    src = REF + "/test/data/issue243/multivariate_2traits"
    raw, fam, bim = copy_plink(src, os.path.join(tmp, "mv2"), n_snps=800)
    d["a_bed"] = raw
    d["a_pheno"] = np.array([[float(x) for x in l.split()[5:7]] for l in fam])
    gemma(tmp, "-bfile", "mv2", "-gk", 1, "-o", "mv2")
    mv_run(tmp, "mv2", "a", os.path.join(tmp, "output", "mv2.cXX.txt"), (1, 2), d, bim)
    # (b) issue188 genotypes, 3 simulated traits with a shared polygenic component (interior V_g, V_e), a few NA
    n_total = len(fam188)
    nb = (n_total + 3) // 4
    codes = np.unpackbits(raw188[3:].reshape(-1, nb)[:400], axis=1, bitorder="little").reshape(400, -1, 2)[:, :n_total]
    g = (codes[:, :, 0] + codes[:, :, 1]).astype(float)  # rough allele count (missing 01 -> 1): only used to simulate
    g = (g - g.mean(1, keepdims=True)) / (g.std(1, keepdims=True) + 1e-9)
    rng = np.random.default_rng(243)
    A = np.array([[1.0, 0.5, 0.2], [0.0, 0.8, 0.3], [0.0, 0.0, 0.7]])
    Y = (g.T @ rng.standard_normal((400, 3)) / np.sqrt(400)) @ A + rng.standard_normal((n_total, 3)) @ np.array(
        [[0.9, 0.2, 0.0], [0.0, 0.8, 0.1], [0.0, 0.0, 1.0]])
    Y += 0.35 * g[7][:, None] * np.array([1.0, -0.5, 0.8])  # one SNP with an effect large enough for Newton-Raphson
    txt = [["%.8g" % v for v in row] for row in Y]
    for i in rng.choice(n_total, 25, replace=False):
        txt[i][rng.integers(0, 3)] = "NA"
    fam_lines = [" ".join(l.split()[:5] + t) + "\n" for l, t in zip(fam188, txt)]
    copy_plink(REF + "/test/data/issue188/2000", os.path.join(tmp, "mv3"), fam_lines=fam_lines)
    d["b_pheno_txt"] = np.array(txt)
    gemma(tmp, "-bfile", "mv3", "-gk", 1, "-o", "mv3")
    mv_run(tmp, "mv3", "b", os.path.join(tmp, "output", "mv3.cXX.txt"), (1, 2, 3), d, bim188)
    # (c) five traits (the pairwise two-trait initialisation of MphInitial, src/mvlmm.cpp:2805-2884) and a covariate besides the
    #     intercept, on every 8th SNP (-snps); REML and score modes (the ML EM of d >= 3 is not reproducible, see the tests)
    A5 = np.triu(rng.uniform(0.2, 0.9, (5, 5)))
    E5 = np.triu(rng.uniform(0.1, 0.5, (5, 5))) + 0.8 * np.eye(5)
    Y5 = (g.T @ rng.standard_normal((400, 5)) / np.sqrt(400)) @ A5 + rng.standard_normal((n_total, 5)) @ E5
    cov = np.column_stack([np.ones(n_total), rng.standard_normal(n_total)])
    Y5 += 0.4 * cov[:, 1:2] * rng.standard_normal((1, 5))
    txt5 = [["%.8g" % v for v in row] for row in Y5]
    fam_lines = [" ".join(l.split()[:5] + t) + "\n" for l, t in zip(fam188, txt5)]
    copy_plink(REF + "/test/data/issue188/2000", os.path.join(tmp, "mv5"), fam_lines=fam_lines)
    np.savetxt(os.path.join(tmp, "cov5.txt"), cov, fmt="%.10g")
    with open(os.path.join(tmp, "snps5.txt"), "w") as f:
        f.writelines(l.split()[1] + "\n" for l in bim188[::8])
    d["c_pheno"] = np.array([[float(x) for x in row] for row in txt5])
    d["c_cov"] = np.loadtxt(os.path.join(tmp, "cov5.txt"))
    d["c_snps_listed"] = np.arange(0, len(bim188), 8)
    gemma(tmp, "-bfile", "mv5", "-gk", 1, "-o", "mv5")
    mv_run(tmp, "mv5", "c", os.path.join(tmp, "output", "mv5.cXX.txt"), (1, 2, 3, 4, 5), d, bim188, modes=(1, 3),
           extra=("-c", "cov5.txt", "-snps", "snps5.txt"))
    np.savez_compressed(os.path.join(OUT, "ref_mv.npz"), **d)
    print("ref_mv.npz:", len(d["a_snp"]), "+", len(d["b_snp"]), "+", len(d["c_snp"]), "SNPs")


def mv_crt(tmp):
    """-crt (PARAM::crt, src/gemma.cpp:1398-1399): the Edgeworth-corrected p values of the SNPs that reach MphNR, on fixture (a)
    of ref_mv.npz (its PLINK files are rebuilt from the stored .bed bytes and phenotypes) and on (b); written to its own file so
    that ref_mv.npz stays byte for byte what it was."""
    fx = np.load(os.path.join(OUT, "ref_mv.npz"))
    d = {}
    Y = fx["a_pheno"]
    n_total = Y.shape[0]
    nb = (n_total + 3) // 4
    ns = (fx["a_bed"].size - 3) // nb
    pre = os.path.join(tmp, "mvc")
    open(pre + ".bed", "wb").write(fx["a_bed"].tobytes())
    with open(pre + ".bim", "w") as f:
        for t in range(ns):
            f.write("1\trs%d\t0\t%d\tA\tG\n" % (t, t + 1))
    with open(pre + ".fam", "w") as f:
        for i in range(n_total):
            f.write("f%d i%d 0 0 1 %r %r\n" % (i, i, float(Y[i, 0]), float(Y[i, 1])))
    gemma(tmp, "-bfile", "mvc", "-gk", 1, "-o", "mvc")
    kfile = os.path.join(tmp, "output", "mvc.cXX.txt")
    bim = ["1\trs%d\t0\t%d\tA\tG\n" % (t, t + 1) for t in range(ns)]
    mv_run(tmp, "mvc", "a", kfile, (1, 2), d, bim, extra=("-crt",))
    changed = {}
    for m in (1, 2, 3, 4):
        for c in MV_COLS_EXTRA:
            k = "a_m%d_%s" % (m, c)
            if k in d:
                assert np.array_equal(d["a_snp"], fx["a_snp"])
                changed[k] = int((d[k] != fx[k]).sum())
    assert changed["a_m1_p_wald"] >= 1 and changed["a_m2_p_lrt"] >= 1, changed
    out = {k.replace("a_", "a_crt_", 1): v for k, v in d.items() if k.startswith("a_m") or k == "a_snp"}
    out["a_crt_rows_changed"] = np.array([changed.get("a_m1_p_wald", 0), changed.get("a_m2_p_lrt", 0), changed.get("a_m3_p_score", 0)])
    # (b) three traits with missing phenotypes (issue188 genotypes from ref_issue188.npz, phenotypes from ref_mv.npz), REML and
    #     score modes as in ref_mv.npz
    f188 = np.load(os.path.join(OUT, "ref_issue188.npz"))
    txt = fx["b_pheno_txt"]
    n3 = txt.shape[0]
    nb3 = (n3 + 3) // 4
    ns3 = (f188["bed"].size - 3) // nb3
    pre3 = os.path.join(tmp, "mvc3")
    open(pre3 + ".bed", "wb").write(f188["bed"].tobytes())
    bim3 = ["1\trs%d\t0\t%d\tA\tG\n" % (t, t + 1) for t in range(ns3)]
    open(pre3 + ".bim", "w").writelines(bim3)
    with open(pre3 + ".fam", "w") as f:
        for i in range(n3):
            f.write("f%d i%d 0 0 1 %s\n" % (i, i, " ".join(txt[i])))
    gemma(tmp, "-bfile", "mvc3", "-gk", 1, "-o", "mvc3")
    d3 = {}
    mv_run(tmp, "mvc3", "b", os.path.join(tmp, "output", "mvc3.cXX.txt"), (1, 2, 3), d3, bim3, modes=(1, 3), extra=("-crt",))
    assert np.array_equal(d3["b_snp"], fx["b_snp"])
    changed_b = {}
    for m in (1, 3):
        for c in MV_COLS_EXTRA:
            k = "b_m%d_%s" % (m, c)
            if k in d3:
                changed_b[k] = int((d3[k] != fx[k]).sum())
    assert changed_b["b_m1_p_wald"] >= 1, changed_b
    out.update({k.replace("b_", "b_crt_", 1): v for k, v in d3.items() if k.startswith("b_m") or k == "b_snp"})
    np.savez_compressed(os.path.join(OUT, "ref_mv_crt.npz"), **out)
    print("ref_mv_crt.npz:", len(d["a_snp"]), "+", len(d3["b_snp"]), "SNPs; rows that -crt changes:", changed, changed_b)


def mv_wide(tmp):
    """ref_mv_wide.npz: what the fixed multivariate kernels do not cover.
    (g) -gxe with two traits (MVLMM::AnalyzePlinkGXE, src/mvlmm.cpp:4416-4870): fixture (a) of ref_mv.npz (issue243 genotypes and
        phenotypes, rebuilt from the stored bytes), a simulated environment variable, every 5th SNP, -lmm 1..4;
    (w) six traits and two covariates besides the intercept on the issue188 genotypes, every 16th SNP, REML and score modes (the
        reference's ML EM is not reproducible from d = 3 on, see tests/test_reference_pin.py)."""
    fx = np.load(os.path.join(OUT, "ref_mv.npz"))
    d = {}
    Y = fx["a_pheno"]
    n_total = Y.shape[0]
    nb = (n_total + 3) // 4
    ns = (fx["a_bed"].size - 3) // nb
    pre = os.path.join(tmp, "mvg")
    open(pre + ".bed", "wb").write(fx["a_bed"].tobytes())
    bim = ["1\trs%d\t0\t%d\tA\tG\n" % (t, t + 1) for t in range(ns)]
    open(pre + ".bim", "w").writelines(bim)
    with open(pre + ".fam", "w") as f:
        for i in range(n_total):
            f.write("f%d i%d 0 0 1 %r %r\n" % (i, i, float(Y[i, 0]), float(Y[i, 1])))
    rng = np.random.default_rng(4416)
    env = np.round(rng.standard_normal(n_total), 6)
    np.savetxt(os.path.join(tmp, "envg.txt"), env, fmt="%.6f")
    with open(os.path.join(tmp, "snpsg.txt"), "w") as f:
        f.writelines("rs%d\n" % t for t in range(0, ns, 5))
    d["g_env"] = env
    d["g_snps_listed"] = np.arange(0, ns, 5)
    gemma(tmp, "-bfile", "mvg", "-gk", 1, "-o", "mvg")
    mv_run(tmp, "mvg", "g", os.path.join(tmp, "output", "mvg.cXX.txt"), (1, 2), d, bim,
           extra=("-gxe", "envg.txt", "-snps", "snpsg.txt"))
    # (w)
    raw188, fam188, bim188 = copy_plink(REF + "/test/data/issue188/2000", os.path.join(tmp, "p188w"))
    n188 = len(fam188)
    nb188 = (n188 + 3) // 4
    codes = np.unpackbits(raw188[3:].reshape(-1, nb188)[:400], axis=1, bitorder="little").reshape(400, -1, 2)[:, :n188]
    g = (codes[:, :, 0] + codes[:, :, 1]).astype(float)
    g = (g - g.mean(1, keepdims=True)) / (g.std(1, keepdims=True) + 1e-9)
    A6 = np.triu(rng.uniform(0.2, 0.8, (6, 6)))
    E6 = np.triu(rng.uniform(0.1, 0.4, (6, 6))) + 0.8 * np.eye(6)
    Y6 = (g.T @ rng.standard_normal((400, 6)) / np.sqrt(400)) @ A6 + rng.standard_normal((n188, 6)) @ E6
    cov = np.column_stack([np.ones(n188), rng.standard_normal(n188), rng.binomial(1, 0.4, n188).astype(float)])
    Y6 += 0.3 * cov[:, 1:2] * rng.standard_normal((1, 6)) + 0.4 * g[11][:, None] * rng.standard_normal((1, 6))
    txt6 = [["%.8g" % v for v in row] for row in Y6]
    fam_lines = [" ".join(l.split()[:5] + t) + "\n" for l, t in zip(fam188, txt6)]
    copy_plink(REF + "/test/data/issue188/2000", os.path.join(tmp, "mv6"), fam_lines=fam_lines)
    np.savetxt(os.path.join(tmp, "cov6.txt"), cov, fmt="%.10g")
    with open(os.path.join(tmp, "snps6.txt"), "w") as f:
        f.writelines(l.split()[1] + "\n" for l in bim188[::16])
    d["w_pheno"] = np.array([[float(x) for x in row] for row in txt6])
    d["w_cov"] = np.loadtxt(os.path.join(tmp, "cov6.txt"))
    d["w_snps_listed"] = np.arange(0, len(bim188), 16)
    gemma(tmp, "-bfile", "mv6", "-gk", 1, "-o", "mv6")
    mv_run(tmp, "mv6", "w", os.path.join(tmp, "output", "mv6.cXX.txt"), (1, 2, 3, 4, 5, 6), d, bim188, modes=(1, 3),
           extra=("-c", "cov6.txt", "-snps", "snps6.txt"))
    np.savez_compressed(os.path.join(OUT, "ref_mv_wide.npz"), **d)
    print("ref_mv_wide.npz:", len(d["g_snp"]), "+", len(d["w_snp"]), "SNPs")


def main():
    if not os.path.exists(GEMMA):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    tmp = tempfile.mkdtemp(prefix="gemma_ref_")
    try:
        which = sys.argv[1:] or ["bxd", "issue188", "mv", "loco", "gene"]
        if "bxd" in which:
            bxd(tmp)
        raw, fam, bim = None, None, None
        if "issue188" in which:
            raw, fam, bim = issue188(tmp)
        if raw is None and ("mv" in which or "loco" in which or "gene" in which):
            raw, fam, bim = copy_plink(REF + "/test/data/issue188/2000", os.path.join(tmp, "p188"))
        if "mv" in which:
            mv(tmp, raw, fam, bim)
        if "mv_crt" in which:
            mv_crt(tmp)
        if "mv_wide" in which:
            mv_wide(tmp)
        if "loco" in which:
            loco(tmp, raw, fam, bim)
        if "gene" in which:
            gene(tmp, raw, fam, bim)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
