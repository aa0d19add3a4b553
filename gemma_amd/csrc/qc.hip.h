// First-pass SNP QC statistics on device: what ReadFile_geno (GEMMA src/gemma_io.cpp:753-853) and
// ReadFile_bed (:942-1049) compute per SNP over the ANALYSED individuals before any kinship / LMM work:
// n_miss, maf = sum g / (2 (n - n_miss)), genotype class counts, polymorphism, and -- for the r2 filter --
// W^T x and x^T x with missing calls replaced by 2*maf.  One wavefront per SNP, two streaming passes.
// The threshold logic itself (order of filters, HWE exact test, r2 = x^T W (W^T W)^-1 W^T x / x^T x) is
// O(c^2) scalar work per SNP and runs on the host in snp_qc_finish().
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include <vector>

#include "ingest.hip.h"

namespace gemma_hip {

constexpr int QC_NSTAT = 8; // n_miss, sum, n0, n1, n2, min, max, v_x ; then c values of W^T x

struct QcArgs {
  const void *src;
  long ld, l;
  const int *idx_map; // analysed individual j -> column in the row (nullptr = identity)
  int n;              // analysed individuals
  int c;              // covariates
  const double *Wt;   // c x n (covariate-major)
  double *out;        // l x (QC_NSTAT + c)
};

template <bool PLINK>
__global__ __launch_bounds__(256) void snp_qc_kernel(QcArgs g) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= g.l) return;
  const int n = g.n;
  const double *xs = reinterpret_cast<const double *>(g.src) + s * g.ld;
  const unsigned char *bs = reinterpret_cast<const unsigned char *>(g.src) + s * g.ld;
  double n_miss = 0.0, sum = 0.0, n0 = 0.0, n1 = 0.0, n2 = 0.0, mn = INFINITY, mx = -INFINITY;
  for (int i = lane; i < n; i += 64) {
    const int p = g.idx_map ? g.idx_map[i] : i;
    double v;
    bool miss;
    if (PLINK) {
      v = plink_value((bs[p >> 2] >> (2 * (p & 3))) & 3u, miss);
    } else {
      v = xs[p];
      miss = isnan(v);
    }
    if (miss) {
      n_miss += 1.0;
    } else {
      sum += v;
      if (v >= 0 && v <= 0.5) n0 += 1.0;           // src/gemma_io.cpp:767-775
      if (v > 0.5 && v < 1.5) n1 += 1.0;
      if (v >= 1.5 && v <= 2.0) n2 += 1.0;
      mn = fmin(mn, v);
      mx = fmax(mx, v);
    }
  }
  n_miss = wsum(n_miss); sum = wsum(sum); n0 = wsum(n0); n1 = wsum(n1); n2 = wsum(n2);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    mn = fmin(mn, __shfl_xor(mn, off, 64));
    mx = fmax(mx, __shfl_xor(mx, off, 64));
  }
  const double maf = sum / (2.0 * ((double)n - n_miss));
  const double fill = maf * 2.0; // :835, :1029
  double vx = 0.0;
  double *o = g.out + s * (QC_NSTAT + g.c);
  for (int a = 0; a < g.c; ++a) {
    double acc = 0.0;
    for (int i = lane; i < n; i += 64) {
      const int p = g.idx_map ? g.idx_map[i] : i;
      double v;
      bool miss;
      if (PLINK) {
        v = plink_value((bs[p >> 2] >> (2 * (p & 3))) & 3u, miss);
      } else {
        v = xs[p];
        miss = isnan(v);
      }
      v = miss ? fill : v;
      acc += g.Wt[(long)a * n + i] * v;
      if (a == 0) vx += v * v;
    }
    acc = wsum(acc);
    if (lane == 0) o[QC_NSTAT + a] = acc;
  }
  vx = wsum(vx);
  if (lane == 0) {
    o[0] = n_miss; o[1] = sum; o[2] = n0; o[3] = n1; o[4] = n2; o[5] = mn; o[6] = mx; o[7] = vx;
  }
}

// Exact HWE test of Wigginton, Cutler & Abecasis (2005), as GEMMA's CalcHWE (src/mathfunc.cpp:546-627)
static inline double calc_hwe_host(size_t n_hom1, size_t n_hom2, size_t n_ab) {
  if (n_hom1 + n_hom2 + n_ab == 0) return 1;
  const long n_aa = (long)(n_hom1 < n_hom2 ? n_hom1 : n_hom2), n_bb = (long)(n_hom1 < n_hom2 ? n_hom2 : n_hom1);
  const long rare = 2 * n_aa + (long)n_ab, geno = (long)n_ab + n_bb + n_aa;
  std::vector<double> het(rare + 1, 0.0);
  long mid = (rare * (2 * geno - rare)) / (2 * geno);
  if ((rare & 1) ^ (mid & 1)) mid++;
  long homr = (rare - mid) / 2, homc = geno - mid - homr;
  het[mid] = 1.0;
  double sum = 1.0;
  for (long h = mid; h > 1; h -= 2) {
    het[h - 2] = het[h] * h * (h - 1.0) / (4.0 * (homr + 1.0) * (homc + 1.0));
    sum += het[h - 2];
    homr++;
    homc++;
  }
  homr = (rare - mid) / 2;
  homc = geno - mid - homr;
  for (long h = mid; h <= rare - 2; h += 2) {
    het[h + 2] = het[h] * 4.0 * homr * homc / ((h + 2.0) * (h + 1.0));
    sum += het[h + 2];
    homr--;
    homc--;
  }
  double p = 0.0;
  const double ref = het[n_ab] / sum;
  for (long i = 0; i <= rare; ++i) {
    const double v = het[i] / sum;
    if (v > ref) continue;
    p += v;
  }
  return p > 1.0 ? 1.0 : p;
}

// (W^T W)^-1 by Gauss-Jordan with partial pivoting (c x c, host)
static inline bool invert_small(std::vector<double> &A, int c) {
  std::vector<double> I((size_t)c * c, 0.0);
  for (int i = 0; i < c; ++i) I[(size_t)i * c + i] = 1.0;
  for (int j = 0; j < c; ++j) {
    int piv = j;
    for (int i = j + 1; i < c; ++i)
      if (fabs(A[(size_t)i * c + j]) > fabs(A[(size_t)piv * c + j])) piv = i;
    if (A[(size_t)piv * c + j] == 0.0) return false;
    if (piv != j)
      for (int k = 0; k < c; ++k) {
        std::swap(A[(size_t)j * c + k], A[(size_t)piv * c + k]);
        std::swap(I[(size_t)j * c + k], I[(size_t)piv * c + k]);
      }
    const double d = A[(size_t)j * c + j];
    for (int k = 0; k < c; ++k) {
      A[(size_t)j * c + k] /= d;
      I[(size_t)j * c + k] /= d;
    }
    for (int i = 0; i < c; ++i) {
      if (i == j) continue;
      const double f = A[(size_t)i * c + j];
      if (f == 0.0) continue;
      for (int k = 0; k < c; ++k) {
        A[(size_t)i * c + k] -= f * A[(size_t)j * c + k];
        I[(size_t)i * c + k] -= f * I[(size_t)j * c + k];
      }
    }
  }
  A = I;
  return true;
}

struct QcCfgHost {
  double maf_level, miss_level, hwe_level, r2_level;
};

// the filter cascade of src/gemma_io.cpp:805-853 (BIMBAM) / :1006-1049 (PLINK), in the reference's order
static inline void snp_qc_finish(const double *stats, size_t l, int n, int c, const double *WtWi, bool plink,
                                 const QcCfgHost &q, int *indicator_snp, double *maf_out, size_t *n_miss_out) {
  for (size_t s = 0; s < l; ++s) {
    const double *o = stats + s * (QC_NSTAT + c);
    const double n_miss = o[0];
    const double maf = o[1] / (2.0 * ((double)n - n_miss));
    if (maf_out) maf_out[s] = maf;
    if (n_miss_out) n_miss_out[s] = (size_t)n_miss;
    int keep = 1;
    if (n_miss / (double)n > q.miss_level) keep = 0;
    if (keep && (maf < q.maf_level || maf > (1.0 - q.maf_level)) && q.maf_level != -1) keep = 0;
    if (keep) {
      if (plink) {
        if ((o[2] + o[3]) == 0 || (o[3] + o[4]) == 0 || (o[4] + o[2]) == 0) keep = 0; // :1017-1020
      } else {
        if (!(o[5] < o[6])) keep = 0; // flag_poly != 1: all observed values identical (or none), :818
      }
    }
    if (keep && q.hwe_level != 0 && q.maf_level != -1) {
      if (calc_hwe_host((size_t)o[2], (size_t)o[4], (size_t)o[3]) < q.hwe_level) keep = 0;
    }
    if (keep && c != 1) {
      double v_w = 0.0;
      for (int a = 0; a < c; ++a) {
        double t = 0.0;
        for (int b = 0; b < c; ++b) t += WtWi[(size_t)a * c + b] * o[QC_NSTAT + b];
        v_w += o[QC_NSTAT + a] * t;
      }
      if (v_w / o[7] > q.r2_level) keep = 0;
    }
    indicator_snp[s] = keep;
  }
}

} // namespace gemma_hip
