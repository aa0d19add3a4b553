"""Text artefacts of the reference binary (oracle/_ref/gemma = /root/reference/src/*.cpp compiled unchanged, see
oracle/Makefile) for the host-layer tests of include/gemma_io_host.hpp -- run in the build container only:

    python tests/golden/make_text_fixtures.py

Writes tests/golden/text/: the BXD inputs the readers are tested on (phenotypes, covariates, annotation and the first
genotype lines -- small public example files of the GEMMA tree, kept verbatim because the tests compare parsers
byte for byte), and what the reference wrote for them: `-gk` cXX (24 x 24 corner), `-eigen` eigenD / eigenU, the first
lines of every `-lmm 1/2/3/4/9` .assoc.txt, and the individual / SNP selection the log reports.
"""
import gzip
import json
import os
import shutil
import subprocess
import tempfile

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


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp()
    base = ["-g", E + "BXD_geno.txt.gz", "-p", E + "BXD_pheno.txt", "-c", E + "BXD_covariates2.txt", "-a", E + "BXD_snps.txt"]
    for f in ("BXD_pheno.txt", "BXD_covariates2.txt"):
        shutil.copy(E + f, os.path.join(OUT, f))
    with gzip.open(E + "BXD_geno.txt.gz", "rt") as f, open(os.path.join(OUT, "BXD_geno_head.txt"), "w") as g:
        rs = []
        for i, line in enumerate(f):
            if i >= HEAD:
                break
            g.write(line)
            rs.append(line.split(",")[0].strip())
    keep = set(rs)
    with open(E + "BXD_snps.txt") as f, open(os.path.join(OUT, "BXD_snps_head.txt"), "w") as g:
        for line in f:
            if line.split()[0].strip(",") in keep:
                g.write(line)
    gemma(tmp, *base, "-gk", "-o", "BXD")
    cxx = os.path.join(tmp, "output", "BXD.cXX.txt")
    with open(cxx) as f, open(os.path.join(OUT, "BXD.cXX.corner.txt"), "w") as g:  # 24 x 24 corner, tokens verbatim
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
    for line in open(os.path.join(tmp, "output", "L1.log.txt")):
        if "=" in line and line.startswith("##"):
            k, v = line[2:].split("=", 1)
            if k.strip().startswith(("number of", "pve", "se(pve)", "vg", "ve", "REMLE", "MLE")):
                meta[k.strip()] = v.strip()
    json.dump(meta, open(os.path.join(OUT, "L1.log.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
