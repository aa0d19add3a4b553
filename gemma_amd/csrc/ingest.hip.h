// Genotype ingest + kinship preparation + matrix centring kernels (gfx950).
// HBM-bound byte/f64 streaming work: one wavefront per SNP row, coalesced accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace gemma_hip {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// PLINK 2-bit code -> genotype (GEMMA src/lmm.cpp:1797-1812; src/gemma_io.cpp:1665-1682):
// v = b0 + 2*b1 : 0 -> 2, 2 -> 1, 3 -> 0, 1 -> missing
__device__ __forceinline__ double plink_value(unsigned v, bool &missing) {
  missing = (v == 1u);
  return (v == 0u) ? 2.0 : (v == 2u) ? 1.0 : 0.0;
}

struct IngestArgs {
  const void *src;    // f64 (ld doubles per SNP) or bytes (ld bytes per SNP)
  long ld;
  long l;
  const int *idx_map; // PLINK: position in ni_total of analysed individual j (nullptr = identity)
  int n;              // individuals written per SNP
  double *dst;        // l x ldo, SNP-major
  long ldo;
  int k_mode;         // kinship only: 1 centred, 2 standardised
};

// LMM ingest: mean-impute only, no centring (GEMMA src/lmm.cpp:1590-1618, :1819-1827).
template <bool PLINK>
__global__ __launch_bounds__(256) void ingest_lmm_kernel(IngestArgs g) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= g.l) return;
  const int n = g.n;
  double tot = 0.0, cnt = 0.0;
  const double *xs = reinterpret_cast<const double *>(g.src) + s * g.ld;
  const unsigned char *bs = reinterpret_cast<const unsigned char *>(g.src) + s * g.ld;
  for (int i = lane; i < n; i += 64) {
    double v;
    bool miss;
    if (PLINK) {
      const int p = g.idx_map ? g.idx_map[i] : i;
      v = plink_value((bs[p >> 2] >> (2 * (p & 3))) & 3u, miss);
    } else {
      v = xs[i];
      miss = isnan(v);
    }
    if (!miss) { tot += v; cnt += 1.0; }
  }
  tot = wsum(tot);
  cnt = wsum(cnt);
  const double mean = tot / cnt; // x_total / (ni_test - n_miss)
  double *d = g.dst + s * g.ldo;
  for (int i = lane; i < n; i += 64) {
    double v;
    bool miss;
    if (PLINK) {
      const int p = g.idx_map ? g.idx_map[i] : i;
      v = plink_value((bs[p >> 2] >> (2 * (p & 3))) & 3u, miss);
    } else {
      v = xs[i];
      miss = isnan(v);
    }
    d[i] = miss ? mean : v;
  }
}

// GXE ingest (GEMMA src/lmm.cpp:2316-2361 BIMBAM, :2487-2536 PLINK): mean-impute, recode 2 - x when x_mean > 1,
// and also write z = x . env; flip[s] records the recoding (beta changes sign, :2403 / :2584).
struct IngestGxeArgs {
  const void *src;
  long ld, l;
  const int *idx_map;
  int n;
  const double *env; // n
  double *X, *Z;     // l x ldo each
  long ldo;
  int *flip;
};
template <bool PLINK>
__global__ __launch_bounds__(256) void ingest_gxe_kernel(IngestGxeArgs g) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= g.l) return;
  const int n = g.n;
  double tot = 0.0, cnt = 0.0;
  const double *xs = reinterpret_cast<const double *>(g.src) + s * g.ld;
  const unsigned char *bs = reinterpret_cast<const unsigned char *>(g.src) + s * g.ld;
  for (int i = lane; i < n; i += 64) {
    double v;
    bool miss;
    if (PLINK) {
      const int p = g.idx_map ? g.idx_map[i] : i;
      v = plink_value((bs[p >> 2] >> (2 * (p & 3))) & 3u, miss);
    } else {
      v = xs[i];
      miss = isnan(v);
    }
    if (!miss) { tot += v; cnt += 1.0; }
  }
  tot = wsum(tot);
  cnt = wsum(cnt);
  const double mean = tot / cnt;
  const bool flip = mean > 1;
  double *dx = g.X + s * g.ldo, *dz = g.Z + s * g.ldo;
  for (int i = lane; i < n; i += 64) {
    double v;
    bool miss;
    if (PLINK) {
      const int p = g.idx_map ? g.idx_map[i] : i;
      v = plink_value((bs[p >> 2] >> (2 * (p & 3))) & 3u, miss);
    } else {
      v = xs[i];
      miss = isnan(v);
    }
    if (miss) v = mean;
    if (flip) v = 2 - v;
    dx[i] = v;
    dz[i] = v * g.env[i];
  }
  if (lane == 0) g.flip[s] = flip ? 1 : 0;
}

// Kinship ingest over ALL individuals: mean over non-missing, impute, centre, optional
// 1/sqrt(var) with var = (sum g^2 + mean^2*n_miss)/n - mean^2
// (GEMMA src/gemma_io.cpp:1487-1538 BimbamKin, :1651-1704 PlinkKin).
template <bool PLINK>
__global__ __launch_bounds__(256) void ingest_kin_kernel(IngestArgs g) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= g.l) return;
  const int n = g.n;
  double tot = 0.0, sq = 0.0, cnt = 0.0;
  const double *xs = reinterpret_cast<const double *>(g.src) + s * g.ld;
  const unsigned char *bs = reinterpret_cast<const unsigned char *>(g.src) + s * g.ld;
  for (int i = lane; i < n; i += 64) {
    double v;
    bool miss;
    if (PLINK) {
      v = plink_value((bs[i >> 2] >> (2 * (i & 3))) & 3u, miss);
    } else {
      v = xs[i];
      miss = isnan(v);
    }
    if (!miss) { tot += v; sq += v * v; cnt += 1.0; }
  }
  tot = wsum(tot);
  sq = wsum(sq);
  cnt = wsum(cnt);
  const double n_miss = (double)n - cnt;
  const double mean = tot / cnt;
  double var = sq + mean * mean * n_miss;
  var /= (double)n;
  var -= mean * mean;
  const bool scale = (g.k_mode == 2 && var != 0);
  const double sc = scale ? 1.0 / sqrt(var) : 1.0;
  double *d = g.dst + s * g.ldo;
  for (int i = lane; i < n; i += 64) {
    double v;
    bool miss;
    if (PLINK) {
      v = plink_value((bs[i >> 2] >> (2 * (i & 3))) & 3u, miss);
    } else {
      v = xs[i];
      miss = isnan(v);
    }
    v = miss ? mean : v;
    v = v + (-1.0 * mean);
    if (scale) v *= sc;
    d[i] = v;
  }
}

// out (cols x ldo) = in^T, in is rows x ld  (individual-major Xlarge -> SNP-major)
__global__ void transpose_kernel(const double *in, long rows, long cols, long ld, double *out,
                                 long ldo) {
  __shared__ double tile[32][33];
  const long c0 = (long)blockIdx.x * 32, r0 = (long)blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y; // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const long i = r0 + r, j = c0 + tx;
    tile[r][tx] = (i < rows && j < cols) ? in[i * ld + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long j = c0 + r, i = r0 + tx;
    if (i < rows && j < cols) out[j * ldo + i] = tile[tx][r];
  }
}

// ---- CenterMatrix (GEMMA src/mathfunc.cpp:147-177) ------------------------------------
// Gw = G * 1 (row sums), one wavefront per row
__global__ __launch_bounds__(256) void rowsum_kernel(const double *G, long n, long ld, double *Gw) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  double s = 0.0;
  const double *row = G + r * ld;
  for (long j = lane; j < n; j += 64) s += row[j];
  s = wsum(s);
  if (lane == 0) Gw[r] = s;
}
// d = sum(Gw) (single block)
__global__ __launch_bounds__(1024) void total_kernel(const double *Gw, long n, double *d) {
  __shared__ double part[16];
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) s += Gw[i];
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += part[w];
    *d = t;
  }
}
// G[i][j] += -(Gw[i] + Gw[j])/n + d/n^2 on the UPPER triangle (dsyr2 + dsyr with CblasUpper), then
// the lower triangle is overwritten by the mirror -- exactly src/mathfunc.cpp:160-171, so a K that
// is not bit-symmetric (cXX.txt at 10 digits) gives the reference's result.
__global__ __launch_bounds__(256) void center_update_kernel(double *G, long n, long ld,
                                                            const double *Gw, const double *d) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  const long i = blockIdx.y;
  if (j >= n || j < i) return;
  const double alpha = -1.0 / (double)n;
  const double beta = (*d) / ((double)n * (double)n);
  double v = G[i * ld + j];
  v += alpha * Gw[i] + alpha * Gw[j];
  v += beta;
  G[i * ld + j] = v;
  if (j != i) G[j * ld + i] = v;
}

// leave-one-chromosome-out kinship from the all-SNP and the per-chromosome matrices (in place on Kc)
__global__ void loco_kernel(const double *__restrict__ Ka, double nsa, double *__restrict__ Kc, double nsc, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) Kc[i] = (nsa * Ka[i] - nsc * Kc[i]) / (nsa - nsc);
}

// eigenvalue post-processing of EigenDecomp_Zeroed (GEMMA src/lapack.cpp:266-277)
__global__ void zero_small_eval_kernel(double *eval, long n, double *trace) {
  __shared__ double part[16];
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    double v = eval[i];
    if (v < 1e-10) { v = 0.0; eval[i] = 0.0; }
    s += v;
  }
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (unsigned w = 0; w < blockDim.x / 64; ++w) t += part[w];
    *trace = t / (double)n;
  }
}

// AnalyzePlink's stale beta/se when CalcRLWald was skipped (GEMMA src/lmm.cpp:1725,1870):
// a failed SNP (NaN logl_H1, a_mode 1) reports the beta/se of the nearest preceding SNP that
// did run CalcRLWald; carry_in covers the previous batch, carry_out receives the batch's last.
struct SumStatRaw { double v[8]; };
__global__ void plink_carry_kernel(SumStatRaw *out, long l, const double *carry_in,
                                   double *carry_out) {
  const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= l) return;
  const bool failed = isnan(out[s].v[7]);
  double b = out[s].v[0], e = out[s].v[1];
  if (failed) {
    long j = s - 1;
    while (j >= 0 && isnan(out[j].v[7])) --j;
    if (j >= 0) { b = out[j].v[0]; e = out[j].v[1]; } else { b = carry_in[0]; e = carry_in[1]; }
    out[s].v[0] = b;
    out[s].v[1] = e;
  }
  if (s == l - 1) { carry_out[0] = b; carry_out[1] = e; }
}

} // namespace gemma_hip
