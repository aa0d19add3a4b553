#!/usr/bin/env python3
"""Bind the REFERENCE to libgemma_hip.so, for real (INTEGRATION.md sections 1, 3, 4).

    apply_patch.py <reference src dir> <scratch dir>

copies the reference's sources into a scratch directory OUTSIDE the repository history (oracle/Makefile `ref_hip` uses a
temporary directory) and inserts, under `#ifdef GEMMA_WITH_HIP`, the calls a GEMMA maintainer would add at four waists:

  src/fastblas.cpp  fast_cblas_dgemm   cblas_dgemm            -> gemma_hip_dgemm   (kinship accumulation, CalcUtX, U^T X)
  src/mathfunc.cpp  CenterMatrix       dgemv / dsyr2 / dsyr   -> gemma_hip_center
  src/lapack.cpp    EigenDecomp_Zeroed EigenDecomp (dsyevr)   -> gemma_hip_eigh
  src/lmm.cpp       LMM::Analyze       the batch_compute body -> gemma_hip_lmm_setup / lmm_batch / lmm_finish

Every insertion is anchored on one line of the reference that must occur exactly once; nothing else of the reference is
changed, its CLI, PARAM, readers, QC, null model and writers run as they are.  This file holds only the inserted lines."""
import os
import shutil
import sys

PROLOGUE = '''
#ifdef GEMMA_WITH_HIP
#include <vector>
#include "gemma_hip.h"
#define enforce_hip(call)                                                          \\
  do {                                                                             \\
    int rc_hip_ = (call);                                                          \\
    if (rc_hip_) fail_msg(std::string("gemma_hip: ") + gemma_hip_last_error());    \\
  } while (0)
#endif
'''


def insert_once(text, anchor, new, where, fname):
    n = text.count(anchor)
    if n != 1:
        raise SystemExit("%s: anchor occurs %d times (expected 1): %r" % (fname, n, anchor))
    i = text.index(anchor)
    if where == "before":
        return text[:i] + new + text[i:]
    if where == "before_second_line":  # between the first and the second line of a two-line anchor
        j = text.index("\n", i) + 1
        return text[:j] + new + text[j:]
    j = i + len(anchor)
    return text[:j] + new + text[j:]


def after_last_include(text, new, fname):
    """after the last #include of the file's leading include block"""
    lines = text.split("\n")
    last = max(k for k, l in enumerate(lines[:150]) if l.startswith("#include"))
    return "\n".join(lines[:last + 1]) + "\n" + new + "\n".join(lines[last + 1:])


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        if f.endswith((".cpp", ".h")):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))

    def edit(fname, fn):
        p = os.path.join(dst, fname)
        t = open(p).read()
        t = fn(after_last_include(t, PROLOGUE, fname))
        open(p, "w").write(t)

    # ---- 1. fast_cblas_dgemm: the one cblas_dgemm call of the path
    def fastblas(t):
        t = insert_once(t, "    cblas_dgemm (CblasRowMajor, transA, transB, M, N, NA,",
                        "#ifdef GEMMA_WITH_HIP\n"
                        "    enforce_hip(gemma_hip_dgemm(*TransA, *TransB, M, N, NA, alpha, A->data, A->tda, B->data, B->tda, beta,\n"
                        "                                C->data, C->tda));\n"
                        "#else\n", "before", "fastblas.cpp")
        return insert_once(t, "                 C->data, C->tda);\n", "#endif\n", "after", "fastblas.cpp")
    edit("fastblas.cpp", fastblas)

    # ---- 2. CenterMatrix (section 3)
    edit("mathfunc.cpp", lambda t: insert_once(
        t, "void CenterMatrix(gsl_matrix *G) {\n",
        "#ifdef GEMMA_WITH_HIP\n"
        "  if (G->tda == G->size2) { enforce_hip(gemma_hip_center(G->data, G->size1)); return; }\n"
        "#endif\n", "after", "mathfunc.cpp"))

    # ---- 3. EigenDecomp_Zeroed (section 3)
    edit("lapack.cpp", lambda t: insert_once(
        t, "                          const size_t flag_largematrix) {\n  EigenDecomp(G,U,eval,flag_largematrix);\n",
        "#ifdef GEMMA_WITH_HIP\n"
        "  if (G->tda == G->size2 && U->tda == U->size2 && eval->stride == 1) {\n"
        "    double trace_hip = 0.0; // eigenvalues < 1e-10 zeroed, mean(eval) returned: this function's body, on the device\n"
        "    enforce_hip(gemma_hip_eigh(G->data, G->size1, U->data, eval->data, &trace_hip));\n"
        "    return trace_hip;\n"
        "  }\n"
        "#endif\n", "before_second_line", "lapack.cpp"))

    # ---- 4. LMM::Analyze: batch_compute becomes one call (section 4)
    def lmm(t):
        t = insert_once(t, "  auto batch_compute = [&](size_t l) { // using a C++ closure\n",
                        "#ifdef GEMMA_WITH_HIP\n"
                        "    { // Xlarge in the reference's own layout (individuals x SNPs); results in SNP order\n"
                        "      if (l == 0) return;\n"
                        "      std::vector<gemma_sumstat> hip_out(l);\n"
                        "      enforce_hip(gemma_hip_lmm_batch(GEMMA_GENO_F64_IDV_MAJOR, Xlarge->data, l, Xlarge->tda, hip_out.data()));\n"
                        "      for (size_t i = 0; i < l; i++) {\n"
                        "        SUMSTAT SNPs = {hip_out[i].beta, hip_out[i].se, hip_out[i].lambda_remle, hip_out[i].lambda_mle,\n"
                        "                        hip_out[i].p_wald, hip_out[i].p_lrt, hip_out[i].p_score, hip_out[i].logl_H1};\n"
                        "        sumStat.push_back(SNPs);\n"
                        "      }\n"
                        "      gsl_matrix_set_zero(Xlarge);\n"
                        "      return;\n"
                        "    }\n"
                        "#endif\n", "after", "lmm.cpp")
        t = insert_once(t, "  auto batch_compute = [&](size_t l) { // using a C++ closure\n",
                        "#ifdef GEMMA_WITH_HIP\n"
                        "  {\n"
                        "    gemma_lmm_cfg hip_cfg = {a_mode, ni_test, n_cvt, l_min, l_max, n_region, l_mle_null, logl_mle_H0, 0};\n"
                        "    enforce(U->tda == U->size2 && UtW->tda == UtW->size2 && eval->stride == 1 && Uty->stride == 1);\n"
                        "    enforce_hip(gemma_hip_lmm_setup(&hip_cfg, U->data, eval->data, UtW->data, Uty->data));\n"
                        "  }\n"
                        "#endif\n", "before", "lmm.cpp")
        return insert_once(t, "  batch_compute(c % msize);\n",
                           "#ifdef GEMMA_WITH_HIP\n"
                           "  enforce_hip(gemma_hip_lmm_finish(&time_UtX, &time_opt)); // GPU minutes of the two stages, as the log expects\n"
                           "#endif\n", "after", "lmm.cpp")
    edit("lmm.cpp", lmm)
    print("patched copy of %s in %s" % (src, dst))


if __name__ == "__main__":
    main()
