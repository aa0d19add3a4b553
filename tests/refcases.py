"""Shared by test_reference_pin.py (CPU: oracle vs reference) and test_gpu_reference.py (GPU: HIP path vs reference):
loads tests/golden/ref_*.npz -- outputs of the reference itself (oracle/_ref/gemma, see tests/golden/make_ref_fixtures.py)
-- and the input decoding both need.  Nothing here computes a statistic."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# .assoc.txt column -> SUMSTAT field (src/lmm.cpp:101-225)
COLS = {"beta": "beta", "se": "se", "logl_H1": "logl_H1", "l_remle": "lambda_remle", "l_mle": "lambda_mle",
        "p_wald": "p_wald", "p_lrt": "p_lrt", "p_score": "p_score"}
# the reference prints 7 significant digits ("%.6e"): half a unit of the last digit is 5e-7 relative
PRINT_TOL = 1.5e-6


def load(name):
    d = np.load(os.path.join(GOLD, name))
    return {k: d[k] for k in d.files}


def rel_err(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    with np.errstate(all="ignore"):
        e = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
    e[np.isnan(got) & np.isnan(ref)] = 0.0
    return e


def assert_stats(got, fx, tag, tol=PRINT_TOL, lam_tol=None, lam_frac=1.0):
    """Every column the reference printed for run `tag` against SUMSTAT records `got`."""
    seen = 0
    for col, field in COLS.items():
        key = "%s_%s" % (tag, col)
        if key not in fx:
            continue
        seen += 1
        e = rel_err(got[field], fx[key])
        if col in ("l_remle", "l_mle") and lam_tol is not None:
            assert np.mean(e <= tol) >= lam_frac, (tag, col, float(np.mean(e <= tol)))
            assert np.nanmax(e) <= lam_tol, (tag, col, float(np.nanmax(e)))
        else:
            assert np.nanmax(e) <= tol, (tag, col, float(np.nanmax(e)), int(np.nanargmax(e)))
    assert seen >= 1, tag


def issue188_inputs(fx):
    """bed rows (p x ceil(n/4) uint8), n_total, phenotype vector (-9 where missing) and its indicator."""
    n_total = int(fx["n_total"])
    nb = (n_total + 3) // 4
    raw = np.ascontiguousarray(fx["bed"][3:].reshape(-1, nb))
    ph = fx["pheno_col6"]
    indp = np.array([0 if s in ("-9", "NA") else 1 for s in ph], dtype=np.int32)
    y_all = np.array([(-9.0 if s in ("-9", "NA") else float(s)) for s in ph])
    return raw, n_total, y_all, indp


def mv_case_inputs(fx, f188, tag):
    """(bed rows, n_total, Y_all (n_total x d), indicator over all traits, indicator of trait 1 -- what the `-gk` run,
    which reads phenotype column 1 only, filtered its SNPs with)."""
    if tag == "a":
        Y = fx["a_pheno"]
        n_total = Y.shape[0]
        bed = fx["a_bed"]
        ind = np.ones(n_total, dtype=np.int32)
        ind1 = ind
    elif tag == "c":
        Y = fx["c_pheno"]
        n_total = Y.shape[0]
        bed = f188["bed"]
        ind = np.ones(n_total, dtype=np.int32)
        ind1 = ind
    else:
        txt = fx["b_pheno_txt"]
        n_total = txt.shape[0]
        bed = f188["bed"]
        ind = np.array([0 if "NA" in row else 1 for row in txt], dtype=np.int32)
        ind1 = np.array([0 if row[0] == "NA" else 1 for row in txt], dtype=np.int32)
        Y = np.array([[(-9.0 if x == "NA" else float(x)) for x in row] for row in txt])
    nb = (n_total + 3) // 4
    return np.ascontiguousarray(bed[3:].reshape(-1, nb)), n_total, Y, ind, ind1


def mv_ref_table(fx, tag, mode, d):
    """The reference's mvLMM columns for one run as a dict shaped like MVLMM.sumStat."""
    pre = "%s_m%d_" % (tag, mode)
    out = {"beta": np.column_stack([fx[pre + "beta_%d" % (i + 1)] for i in range(d)]),
           "Vbeta": np.column_stack([fx[pre + "Vbeta_%d_%d" % (i + 1, j + 1)] for i in range(d) for j in range(i, d)])}
    for c in ("p_wald", "p_lrt", "p_score"):
        if pre + c in fx:
            out[c] = fx[pre + c]
    return out


def mv_row_err(got, ref):
    """Per-SNP worst relative error over the printed fields, each field scaled by its largest entry in that row."""
    l = ref["beta"].shape[0]
    worst = np.zeros(l)
    for k, r in ref.items():
        g = np.asarray(got[k]).reshape(l, -1)
        r = np.asarray(r).reshape(l, -1)
        scale = np.maximum(np.abs(r).max(axis=1, keepdims=True), 1e-300)
        worst = np.maximum(worst, (np.abs(g - r) / scale).max(axis=1))
    return worst
