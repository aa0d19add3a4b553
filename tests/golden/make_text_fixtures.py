"""Text artefacts of the reference binary (oracle/_ref/gemma = /root/reference/src/*.cpp compiled unchanged, see
oracle/Makefile) for the host-layer tests of include/gemma_io_host.hpp / gemma_host.hpp -- run in the build container only:

    python tests/golden/make_text_fixtures.py

Writes tests/golden/text/ (deterministic: a second run reproduces the same bytes):
* inputs: the reference's BXD example (mean genotypes, trait, covariates, annotation), stored gzip-compressed under names of
  their own (bxd_*.gz; the readers take gzip directly) plus the first 120 genotype lines as plain text; P.* = the first
  240 individuals x 800 SNPs of test/data/issue188 re-packed (missing calls, -9 phenotypes) with a covariate file that has
  no intercept column and one NA row; H.* = a 300 x 600 synthetic PLINK set WITH heterozygotes (tests/cpp/io_host_check.cpp
  plinkgen) for the HWE filter;
* what the reference wrote for them: BXD -gk 1 / -gk 2 (24 x 24 corners of cXX / sXX), -eigen (eigenD / eigenU), the first
  120 lines of every -lmm 1/2/3/4/9 .assoc.txt, -lmm 1 -snps (Ls); P: cXX head, -lmm 4 with and without covariates, -lmm 1
  with tightened -miss / -maf (P1q), -notsnp (P1n), -km 2 (P1km2), -lm 4 with covariates (Plm4c); H: -lmm 1 with and
  without -hwe 0.05; and the counts / null-model lines of each run's log (*.log.json).
"""
import gzip
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GEMMA = os.path.join(ROOT, "oracle", "_ref", "gemma")
E = "/root/reference/example/"
OUT = os.path.join(ROOT, "tests", "golden", "text")
HEAD = 120  # lines kept of the long tables


def gemma(tmp, *args):
    r = subprocess.run([GEMMA] + [str(a) for a in args], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(r.stdout.decode()[-2000:])


def head(src, dst, n):
    with open(src) as f, open(dst, "w") as g:
        for i, line in enumerate(f):
            if i >= n:
                break
            g.write(line)


def plink_subset(tmp, ni=240, ns=800):
    """The first `ni` individuals x `ns` SNPs of the reference's test/data/issue188/2000 PLINK set (missing calls,
    unphenotyped individuals), re-packed, and what the reference writes for it: cXX (first rows), -lmm 4 with and
    without covariates, -lmm 1 with tightened filters, -lm 4 with covariates."""
    src = "/root/reference/test/data/issue188/2000"
    fam = [l for l in open(src + ".fam") if l.strip()]
    bim = [l for l in open(src + ".bim") if l.strip()][:ns]
    raw = np.fromfile(src + ".bed", dtype=np.uint8)
    nb = (len(fam) + 3) // 4
    rows = raw[3:3 + ns * nb].reshape(ns, nb)
    codes = np.stack([(rows >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(ns, nb * 4)[:, :ni]
    pad = np.full((ns, (ni + 3) // 4 * 4), 0, dtype=np.uint8)
    pad[:, :ni] = codes
    packed = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
    pre = os.path.join(tmp, "P")
    open(pre + ".bed", "wb").write(bytes([0x6C, 0x1B, 0x01]) + packed.tobytes())
    open(pre + ".bim", "w").writelines(bim)
    open(pre + ".fam", "w").writelines(fam[:ni])
    rng = np.random.default_rng(240)
    cov = np.column_stack([rng.standard_normal(ni), rng.integers(0, 2, ni).astype(float)])  # no intercept column
    lines = [" ".join("%.10g" % v for v in r) for r in cov]
    lines[5] = "NA 1"
    open(os.path.join(tmp, "P.cov.txt"), "w").write("\n".join(lines) + "\n")
    gemma(tmp, "-bfile", "P", "-gk", 1, "-o", "P")
    cxx = os.path.join(tmp, "output", "P.cXX.txt")
    gemma(tmp, "-bfile", "P", "-k", cxx, "-lmm", 4, "-o", "P4")
    gemma(tmp, "-bfile", "P", "-k", cxx, "-lmm", 4, "-c", "P.cov.txt", "-o", "P4c")
    gemma(tmp, "-bfile", "P", "-k", cxx, "-lmm", 1, "-miss", 0.02, "-maf", 0.05, "-o", "P1q")
    gemma(tmp, "-bfile", "P", "-lm", 4, "-c", "P.cov.txt", "-o", "Plm4c")
    gemma(tmp, "-bfile", "P", "-k", cxx, "-lmm", 1, "-notsnp", "-o", "P1n")
    # -km 2: the same kinship as "id id value" triples (upper triangle) over the .fam ids
    ids = [l.split()[1] for l in fam[:ni]]
    toks = [l.rstrip("\n").split("\t") for l in open(cxx)]
    with open(os.path.join(tmp, "P.km2.txt"), "w") as f:
        for i in range(ni):
            for j in range(i, ni):
                f.write("%s\t%s\t%s\n" % (ids[i], ids[j], toks[i][j]))
    gemma(tmp, "-bfile", "P", "-k", "P.km2.txt", "-km", 2, "-lmm", 1, "-o", "P1km2")
    for ext in (".bed", ".bim", ".fam", ".cov.txt"):
        shutil.copy(pre + ext, os.path.join(OUT, "P" + ext))
    head(cxx, os.path.join(OUT, "P.cXX.head.txt"), 8)
    for tag in ("P4", "P4c", "P1q", "Plm4c", "P1n", "P1km2"):
        with open(os.path.join(tmp, "output", tag + ".assoc.txt"), "rb") as f, \
                gzip.GzipFile(os.path.join(OUT, tag + ".assoc.txt.gz"), "wb", mtime=0) as g:
            g.write(f.read())
        meta = {}
        for line in open(os.path.join(tmp, "output", tag + ".log.txt")):
            if "=" in line and line.startswith("##"):
                k, v = line[2:].split("=", 1)
                if k.strip().startswith(("number of", "pve", "se(pve)", "vg", "ve", "REMLE", "MLE")):
                    meta[k.strip()] = v.strip()
        json.dump(meta, open(os.path.join(OUT, tag + ".log.json"), "w"), indent=1, sort_keys=True)


def hwe_set(tmp):
    """A small synthetic PLINK set WITH heterozygotes (the issue188 lines are inbred: no hets, every SNP fails any HWE
    test) from tests/cpp/io_host_check.cpp's generator, and the reference's `-hwe 0.05` run on it: which SNPs survive
    CalcHWE (src/mathfunc.cpp:546-640) and their statistics."""
    exe = os.path.join(tmp, "io_check")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "io_host_check.cpp"), "-L" + os.path.join(ROOT, "gemma_amd"),
                           "-lgemma_hip", "-Wl,-rpath," + os.path.join(ROOT, "gemma_amd"), "-lz", "-pthread", "-o", exe])
    subprocess.check_call([exe, "plinkgen", os.path.join(tmp, "H"), "300", "600", "2"])
    gemma(tmp, "-bfile", "H", "-gk", 1, "-o", "H")
    cxx = os.path.join(tmp, "output", "H.cXX.txt")
    gemma(tmp, "-bfile", "H", "-k", cxx, "-lmm", 1, "-hwe", 0.05, "-o", "Hhwe")
    gemma(tmp, "-bfile", "H", "-k", cxx, "-lmm", 1, "-o", "Hall")
    for ext in (".bed", ".bim", ".fam"):
        shutil.copy(os.path.join(tmp, "H" + ext), os.path.join(OUT, "H" + ext))
    for tag in ("Hhwe", "Hall"):
        with open(os.path.join(tmp, "output", tag + ".assoc.txt"), "rb") as f, \
                gzip.GzipFile(os.path.join(OUT, tag + ".assoc.txt.gz"), "wb", mtime=0) as g:
            g.write(f.read())


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp()
    base = ["-g", E + "BXD_geno.txt.gz", "-p", E + "BXD_pheno.txt", "-c", E + "BXD_covariates2.txt", "-a", E + "BXD_snps.txt"]
    # the reference's example inputs are stored compressed under names of their own (the readers take gzip directly)
    for f, dst in (("BXD_pheno.txt", "bxd_trait.txt.gz"), ("BXD_covariates2.txt", "bxd_cvt.txt.gz")):
        with open(E + f, "rb") as src, gzip.GzipFile(os.path.join(OUT, dst), "wb", mtime=0) as g:
            g.write(src.read())
    with gzip.open(E + "BXD_geno.txt.gz", "rt") as f, open(os.path.join(OUT, "bxd_mean_genotypes_head.txt"), "w") as g:
        rs = []
        for i, line in enumerate(f):
            if i >= HEAD:
                break
            g.write(line)
            rs.append(line.split(",")[0].strip())
    keep = set(rs)
    with open(E + "BXD_snps.txt") as f, open(os.path.join(OUT, "bxd_anno_head.txt"), "w") as g:
        for line in f:
            if line.split()[0].strip(",") in keep:
                g.write(line)
    gemma(tmp, *base, "-gk", "-o", "BXD")
    cxx = os.path.join(tmp, "output", "BXD.cXX.txt")
    with open(cxx) as f, open(os.path.join(OUT, "BXD.cXX.corner.txt"), "w") as g:  # 24 x 24 corner, tokens verbatim
        for i, line in enumerate(f):
            if i < 24:
                g.write("\t".join(line.rstrip("\n").split("\t")[:24]) + "\n")
    gemma(tmp, *base, "-gk", 2, "-o", "BXD2")  # standardised kinship from the text input
    with open(os.path.join(tmp, "output", "BXD2.sXX.txt")) as f, open(os.path.join(OUT, "BXD.sXX.corner.txt"), "w") as g:
        for i, line in enumerate(f):
            if i < 24:
                g.write("\t".join(line.rstrip("\n").split("\t")[:24]) + "\n")
    gemma(tmp, *base, "-k", cxx, "-eigen", "-o", "E")
    shutil.copy(os.path.join(tmp, "output", "E.eigenD.txt"), os.path.join(OUT, "E.eigenD.txt"))
    shutil.copy(os.path.join(tmp, "output", "E.eigenU.txt"), os.path.join(OUT, "E.eigenU.txt"))
    meta = {}
    for m in (1, 2, 3, 4, 9):
        gemma(tmp, *base, "-k", cxx, "-lmm", m, "-no-check", "-maf", "0.1", "-o", "L%d" % m)
        head(os.path.join(tmp, "output", "L%d.assoc.txt" % m), os.path.join(OUT, "L%d.assoc.head.txt" % m), HEAD)
    # -snps: a listed subset is analysed (-ksnps / -gwasnps are in the help text but are not parsed, src/gemma.cpp:468)
    rs_all = [l.split()[0] for l in open(E + "BXD_snps.txt")]
    with open(os.path.join(OUT, "bxd_snps7.txt"), "w") as f:
        f.writelines(r + "\n" for r in rs_all[::7])
    gemma(tmp, *base, "-k", cxx, "-lmm", 1, "-no-check", "-maf", "0.1", "-snps", os.path.join(OUT, "bxd_snps7.txt"), "-o", "Ls")
    with open(os.path.join(tmp, "output", "Ls.assoc.txt"), "rb") as f, gzip.GzipFile(os.path.join(OUT, "Ls.assoc.txt.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    for line in open(os.path.join(tmp, "output", "L1.log.txt")):
        if "=" in line and line.startswith("##"):
            k, v = line[2:].split("=", 1)
            if k.strip().startswith(("number of", "pve", "se(pve)", "vg", "ve", "REMLE", "MLE")):
                meta[k.strip()] = v.strip()
    json.dump(meta, open(os.path.join(OUT, "L1.log.json"), "w"), indent=1, sort_keys=True)
    plink_subset(tmp)
    hwe_set(tmp)
    with gzip.open(E + "BXD_geno.txt.gz", "rb") as f, gzip.GzipFile(os.path.join(OUT, "bxd_mean_genotypes.txt.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    with open(E + "BXD_snps.txt", "rb") as f, gzip.GzipFile(os.path.join(OUT, "bxd_anno.txt.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    shutil.rmtree(tmp)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
