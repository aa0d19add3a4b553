"""Generates tests/golden/*.npz from the reference's example data (run in the build container,
where /root/reference exists; the GPU box never sees /root/reference).

bxd.npz: the BXD run of test/dev_tests.rb:26-55 (gemma -gk, then -lmm 2 / -lmm 9 -maf 0.1) pushed
through the oracle: the rotated inputs (U, eval, UtW, Uty), the analysed genotype rows, the null
model, the oracle's SUMSTAT for every a_mode, and the reference's own golden numbers.
The oracle outputs stored here reproduce those golden numbers to every printed digit
(asserted below and again in tests/test_oracle_golden.py).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF = "/root/reference/example/"


def main():
    rs, G = O.read_bimbam_geno(REF + "BXD_geno.txt.gz")
    y, indp = O.read_pheno(REF + "BXD_pheno.txt")
    cvt, indc = O.read_cvt(REF + "BXD_covariates2.txt")
    ind, W = O.process_cvt_phen(indp, cvt, indc)
    isnp_k, _, _ = O.qc_snps(G, ind, W)  # -gk run: default -maf 0.01
    K = O.calc_kin(G[isnp_k == 1], 1)
    checksum = sum(float("%.2f" % float(("%.10g" % v)[:6])) for v in K.ravel())
    assert "%.0f" % checksum == "-116"  # test/dev_test_suite.sh:52
    K10 = O.round10(K)  # cXX.txt hand-off at 10 significant digits
    isnp, maf, nmiss = O.qc_snps(G, ind, W, maf_level=0.1)
    out = {}
    aux = None
    for mode in (1, 2, 3, 4, 9):
        st, null, aux = O.run_lmm(mode, G, ind, isnp, y, W, K10)
        out["stat_mode%d" % mode] = st
    st2, st9 = out["stat_mode2"], out["stat_mode9"]
    assert "%.6e" % st2["p_lrt"][0] == "1.234747e-01"      # dev_tests.rb:42
    assert "%.6e" % np.nanmax(st2["p_lrt"]) == "9.997119e-01"  # dev_tests.rb:43
    assert "%.7g" % np.nanmax(st9["lambda_mle"]) == "0.7531109"  # dev_tests.rb:53
    assert st2.shape[0] == 7317  # 73180 words = (7317+1)*10, dev_test_suite.sh:83
    sel = ind == 1
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "bxd.npz"),
        U=aux["U"], eval=aux["eval"], UtW=aux["UtW"], Uty=aux["Uty"], X=aux["X"].astype(np.float32),
        K_sub=K10[np.ix_(sel, sel)], K_full_corner=K[:8, :8], kin_checksum=np.array(checksum),
        G_kin_head=G[isnp_k == 1][:64].astype(np.float32), K_head=O.calc_kin(G[isnp_k == 1][:64], 1),
        null=np.array([null[k] for k in ("l_mle_null", "logl_mle_H0", "l_remle_null", "logl_remle_H0",
                                          "pve", "pve_se", "trace_G")]),
        golden=np.array([1.234747e-01, 9.997119e-01, 0.7531109]),
        **out)
    print("wrote bxd.npz", aux["X"].shape)


if __name__ == "__main__":
    main()
