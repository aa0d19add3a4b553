// gemma_host.hpp -- C++ host-side mirror of the GEMMA interfaces on the kinship + univariate-LMM
// path, implemented over the C ABI of include/gemma_hip.h (header-only, C++11, no GSL needed).
//
// Same names, argument order and error behaviour as the reference so that a GEMMA source file can
// switch with a namespace alias; each function cites what it mirrors (file:line in the GEMMA tree).
// `Matrix` / `Vector` are layout-compatible views of gsl_matrix / gsl_vector data: row-major with
// leading dimension tda, vector stride in elements.
//
// Contents: fast_dgemm / CenterMatrix / EigenDecomp_Zeroed / CalcUtX; PlinkKin / BimbamKin; WriteMatrix / WriteVector /
// ReadFile_kin (the 10-digit hand-off); CalcLambdaNull; class LMM (AnalyzePlink, AnalyzePlinkGXE, AnalyzeFeed / AnalyzeRows,
// AnalyzeGene*, WriteFiles), class LM (-lm), class MVLMM (-lmm -n a b c).  Host-side machinery they share: BlockPrefetch
// (genotype blocks produced one ahead of the device by a helper thread), AssocLine / write_rows (text rows formatted by a
// thread pool, byte-identical to the reference's ofstream output), shard_range / shard_keep (SNP sharding over ranks).
// Link with -pthread.  The file readers and feeders that sit on top of this are in gemma_io_host.hpp.
//
// Errors: GEMMA prints a message and raises SIGINT through fail_msg / enforce_msg (src/debug.h:113-164) or
// sets cPar.error and returns false.  Here void functions throw gemma_amd::HipError (code + text) and the
// bool readers return false after printing the message, exactly where the reference returns false.
#ifndef GEMMA_HOST_HPP
#define GEMMA_HOST_HPP

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <mutex>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gemma_hip.h"

namespace gemma_amd {

constexpr size_t LMM_BATCH_SIZE = 20000; // src/lmm.h:33
constexpr size_t K_BATCH_SIZE = 20000;   // src/param.h:32

struct HipError : std::runtime_error {
  int code;
  HipError(int c, const std::string &where)
      : std::runtime_error(where + ": " + gemma_hip_strerror(c) + " -- " + gemma_hip_last_error()), code(c) {}
};
inline void enforce_hip(int rc, const char *where) {
  if (rc != GEMMA_HIP_OK) throw HipError(rc, where);
}

struct Matrix { // gsl_matrix view
  size_t size1, size2, tda;
  double *data;
};
struct Vector { // gsl_vector view
  size_t size, stride;
  double *data;
};
inline Matrix matrix_view(double *p, size_t r, size_t c) { return Matrix{r, c, c, p}; }
inline Vector vector_view(double *p, size_t n) { return Vector{n, 1, p}; }

// class SUMSTAT, src/param.h:54-66
struct SUMSTAT {
  double beta, se, lambda_remle, lambda_mle, p_wald, p_lrt, p_score, logl_H1;
};
// class SNPINFO, src/param.h:38-52 (fields WriteFiles prints)
struct SNPINFO {
  std::string chr, rs_number;
  double cM;
  long int base_position;
  std::string a_minor, a_major;
  size_t n_miss;
  double missingness, maf;
  size_t n_idv, n_nb, file_position;
};

// fast_dgemm, src/fastblas.cpp:216-230: C = alpha*op(A)*op(B) + beta*C; "Range error in dgemm" on mismatch
inline void fast_dgemm(const char *TransA, const char *TransB, const double alpha, const Matrix *A,
                       const Matrix *B, const double beta, Matrix *C) {
  const bool tA = (*TransA == 'T' || *TransA == 't'), tB = (*TransB == 'T' || *TransB == 't');
  const size_t M = C->size1, N = C->size2;
  const size_t K = tA ? A->size1 : A->size2;
  const size_t Ma = tA ? A->size2 : A->size1, Kb = tB ? B->size2 : B->size1, Nb = tB ? B->size1 : B->size2;
  if (Ma != M || Nb != N || Kb != K) throw HipError(GEMMA_HIP_EINVAL, "Range error in dgemm");
  enforce_hip(gemma_hip_dgemm(*TransA, *TransB, M, N, K, alpha, A->data, A->tda, B->data, B->tda, beta, C->data,
                              C->tda),
              "fast_dgemm");
}
inline void fast_eigen_dgemm(const char *TransA, const char *TransB, const double alpha, const Matrix *A,
                             const Matrix *B, const double beta, Matrix *C) {
  fast_dgemm(TransA, TransB, alpha, A, B, beta, C); // src/fastblas.cpp:232-236
}

// CenterMatrix, src/mathfunc.cpp:147-177
inline void CenterMatrix(Matrix *G) {
  if (G->tda != G->size2 || G->size1 != G->size2) throw HipError(GEMMA_HIP_EINVAL, "CenterMatrix: contiguous square G");
  enforce_hip(gemma_hip_center(G->data, G->size1), "CenterMatrix");
}

// EigenDecomp_Zeroed, src/lapack.cpp:260-291: G destroyed, returns trace_G = mean(eval)
inline double EigenDecomp_Zeroed(Matrix *G, Matrix *U, Vector *eval, const size_t /*flag_largematrix*/) {
  if (G->tda != G->size2 || U->tda != U->size2 || eval->stride != 1 || G->size1 != G->size2 ||
      U->size1 != G->size1 || eval->size != G->size1)
    throw HipError(GEMMA_HIP_EINVAL, "EigenDecomp_Zeroed: shapes");
  double trace = 0.0;
  enforce_hip(gemma_hip_eigh(G->data, G->size1, U->data, eval->data, &trace), "EigenDecomp_Zeroed");
  return trace;
}

// CalcUtX(U, X, UtX), src/mathfunc.cpp:504-506
inline void CalcUtX(const Matrix *U, const Matrix *X, Matrix *UtX) { fast_dgemm("T", "N", 1.0, U, X, 0.0, UtX); }

// ---------------------------------------------------------------------------------------------------------------
// two-slot producer / consumer: a host thread fills block k+1 (file read, text parsing) while the calling thread has
// block k on the device.  The C ABI is entered from the calling thread only (gemma_hip.h: one calling host thread).
// ---------------------------------------------------------------------------------------------------------------
class BlockPrefetch {
public:
  // fill(slot memory, slot index 0/1) -> rows written; 0 = end of input; (size_t)-1 = malformed input
  typedef std::function<size_t(void *, int)> Fill;
  BlockPrefetch(size_t slot_bytes, Fill fill) : fill_(fill), held_(-1), turn_(0), done_(false), stop_(false) {
    for (int k = 0; k < 2; ++k) {
      buf_[k].resize(slot_bytes);
      full_[k] = false;
      rows_[k] = 0;
    }
    th_ = std::thread([this] { produce(); });
  }
  ~BlockPrefetch() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
      full_[0] = full_[1] = false;
    }
    cv_.notify_all();
    th_.join();
  }
  BlockPrefetch(const BlockPrefetch &) = delete;
  BlockPrefetch &operator=(const BlockPrefetch &) = delete;
  // hands back the previous block, waits for the next one; 0 at the end, (size_t)-1 on malformed input.
  // *index (optional) = which of the two slots holds the block (for per-slot side data of the producer)
  size_t next(void *&slot, int *index = nullptr) {
    std::unique_lock<std::mutex> g(m_);
    if (held_ >= 0) {
      full_[held_] = false;
      held_ = -1;
      cv_.notify_all();
    }
    if (done_) return 0;
    const int k = turn_;
    cv_.wait(g, [&] { return full_[k]; });
    if (err_) std::rethrow_exception(err_);
    const size_t n = rows_[k];
    if (n == 0 || n == (size_t)-1) {
      done_ = true;
      return n;
    }
    held_ = k;
    turn_ ^= 1;
    slot = buf_[k].data();
    if (index) *index = k;
    return n;
  }

private:
  void produce() {
    for (int k = 0;; k ^= 1) {
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return !full_[k] || stop_; });
        if (stop_) return;
      }
      size_t n = 0;
      std::exception_ptr err;
      try {
        n = fill_(buf_[k].data(), k);
      } catch (...) {
        err = std::current_exception();
      }
      {
        std::lock_guard<std::mutex> g(m_);
        rows_[k] = n;
        full_[k] = true;
        if (err) err_ = err;
      }
      cv_.notify_all();
      if (err || n == 0 || n == (size_t)-1) return;
    }
  }
  Fill fill_;
  std::vector<unsigned char> buf_[2];
  bool full_[2];
  size_t rows_[2];
  int held_, turn_;
  bool done_, stop_;
  std::exception_ptr err_;
  std::mutex m_;
  std::condition_variable cv_;
  std::thread th_;
};

// rows per block of the text / .bed feeders; GEMMA_HIP_IO_BLOCK overrides (the tests use it to force many blocks)
inline size_t io_block_rows(size_t dflt) {
  const char *env = getenv("GEMMA_HIP_IO_BLOCK");
  return env && atol(env) > 0 ? (size_t)atol(env) : dflt;
}

// up to max_rows rows of the SNPs t >= t_next with indicator_snp[t] != 0 from a .bed stream into dst (n_bit bytes per
// row); consecutive kept SNPs are fetched with one read, the stream is only repositioned across dropped ones
inline size_t read_bed_rows(std::ifstream &infile, const std::vector<int> &indicator_snp, size_t &t_next, size_t n_bit,
                            unsigned char *dst, size_t max_rows) {
  size_t l = 0;
  const size_t ns = indicator_snp.size();
  while (l < max_rows && t_next < ns) {
    if (indicator_snp[t_next] == 0) {
      ++t_next;
      continue;
    }
    size_t run = 1;
    while (l + run < max_rows && t_next + run < ns && indicator_snp[t_next + run] != 0) ++run;
    infile.seekg((std::streamoff)(t_next * n_bit + 3)); // 3 magic bytes, skipped unchecked like the reference
    infile.read(reinterpret_cast<char *>(dst + l * n_bit), (std::streamsize)(run * n_bit));
    if ((size_t)infile.gcount() != run * n_bit) return (size_t)-1;
    l += run;
    t_next += run;
  }
  return l;
}

// SNP sharding over ranks (one process per GPU): rank r of w analyses the analysed SNPs
// [r * ceil(p / w), (r + 1) * ceil(p / w)) -- contiguous, so that every rank reads one contiguous range of the genotype
// file and the per-rank .assoc.txt parts concatenate in rank order (gemma_amd/dist.py: shard_range is the same rule)
inline void shard_range(size_t p, int rank, int world, size_t &begin, size_t &end) {
  const size_t per = world > 0 ? (p + (size_t)world - 1) / (size_t)world : p;
  begin = std::min(p, per * (size_t)rank);
  end = std::min(p, begin + per);
}

// restricts a 0/1 keep vector to the share of `rank` (the k-th kept entry stays iff k lies in shard_range)
inline void shard_keep(std::vector<int> &keep, int rank, int world) {
  if (world <= 1) return;
  size_t p = 0, b, e, k = 0;
  for (int v : keep) p += v != 0;
  shard_range(p, rank, world, b, e);
  for (size_t t = 0; t < keep.size(); ++t)
    if (keep[t]) {
      if (k < b || k >= e) keep[t] = 0;
      ++k;
    }
}

// The device-resident chain (gemma_hip.h: kin_end_keep -> eigh_kept_K -> calc_utx_kept -> lmm_setup_kept): how a kinship
// routine ends.  keep: K stays on the device (matrix_kin->data may be null, size1 must be ni_total); with world > 1 this
// rank accumulates only ITS share of the analysed SNPs and the partial sums are all-reduced (ncclAllReduce) in kin_end_keep.
struct KinKeep {
  bool keep = false;
  int rank = 0, world = 1;
};
inline void kin_finish(Matrix *matrix_kin, const KinKeep &kk, const char *who) {
  size_t ns = 0;
  if (kk.keep)
    enforce_hip(gemma_hip_kin_end_keep(&ns, kk.world > 1 ? 1 : 0), who);
  else
    enforce_hip(gemma_hip_kin_end(matrix_kin->data, &ns), who);
}
// EigenDecomp_Zeroed on the kept K: rows / columns of the analysed individuals, CenterMatrix, eigendecomposition, all on
// the device; eval comes back (U stays: CalcUtXKept / LMM::kept_U), returns trace_G
inline double EigenDecompKept(const std::vector<int> &indicator_idv, Vector *eval) {
  double tr = 0.0;
  enforce_hip(gemma_hip_eigh_kept_K(indicator_idv.data(), indicator_idv.size(), eval->data, &tr), "EigenDecomp_Zeroed (kept K)");
  return tr;
}
// The same as a COLLECTIVE of all ranks (each holds the same kept K after the all-reduce): the back-transformations of the
// eigensolver are shared out by eigenvector and every rank ends with the same kept (U, eval) -- no broadcast afterwards
// (csrc/eigh.hip.h "Several ranks")
inline double EigenDecompKeptSharded(const std::vector<int> &indicator_idv, Vector *eval) {
  double tr = 0.0;
  enforce_hip(gemma_hip_eigh_kept_K_sharded(indicator_idv.data(), indicator_idv.size(), eval->data, &tr),
              "EigenDecomp_Zeroed (kept K, sharded)");
  return tr;
}
// CalcUtX on the kept U
inline void CalcUtXKept(const Matrix *X, Matrix *UtX) {
  if (X->tda != X->size2 || UtX->tda != UtX->size2 || UtX->size1 != X->size1 || UtX->size2 != X->size2)
    throw HipError(GEMMA_HIP_EINVAL, "CalcUtX (kept U): contiguous n x m matrices");
  enforce_hip(gemma_hip_calc_utx_kept(X->data, X->size1, X->size2, UtX->data), "CalcUtX (kept U)");
}

// PlinkKin, src/gemma_io.cpp:1599-1738: the device decodes, imputes, centres/scales and accumulates
inline bool PlinkKin(const std::string &file_bed, std::vector<int> &indicator_snp_in, const int k_mode,
                     const int /*display_pace*/, Matrix *matrix_kin, const KinKeep &kk = KinKeep()) {
  std::vector<int> indicator_snp(indicator_snp_in);
  if (kk.keep) shard_keep(indicator_snp, kk.rank, kk.world);
  std::ifstream infile(file_bed.c_str(), std::ios::binary);
  if (!infile) {
    std::cout << "error reading bed file:" << file_bed << std::endl;
    return false;
  }
  const size_t ni_total = matrix_kin->size1;
  const size_t n_bit = (ni_total + 3) / 4;
  if (matrix_kin->tda != matrix_kin->size2) return false;
  enforce_hip(gemma_hip_kin_begin(ni_total, k_mode), "PlinkKin");
  const size_t B = io_block_rows(K_BATCH_SIZE);
  size_t t_next = 0;
  BlockPrefetch pf(B * n_bit, [&](void *slot, int) {
    return read_bed_rows(infile, indicator_snp, t_next, n_bit, static_cast<unsigned char *>(slot), B);
  });
  for (;;) {
    void *slot = nullptr;
    const size_t l = pf.next(slot);
    if (l == (size_t)-1) {
      std::cout << "error reading bed file:" << file_bed << " (truncated)" << std::endl;
      return false;
    }
    if (l == 0) break;
    enforce_hip(gemma_hip_kin_add(GEMMA_GENO_PLINK_2BIT, slot, l, n_bit), "PlinkKin");
  }
  kin_finish(matrix_kin, kk, "PlinkKin");
  return true;
}

// BimbamKin, src/gemma_io.cpp:1418-1597 (plain-text mean genotype file; "NA" = missing)
inline bool BimbamKin(const std::string file_geno, std::vector<int> &indicator_snp, const int k_mode,
                      const int /*display_pace*/, Matrix *matrix_kin) {
  std::ifstream infile(file_geno.c_str());
  if (!infile) {
    std::cout << "error reading genotype file:" << file_geno << std::endl;
    return false;
  }
  const size_t ni_total = matrix_kin->size1;
  enforce_hip(gemma_hip_kin_begin(ni_total, k_mode), "BimbamKin");
  const size_t bsz = 2048;
  std::vector<double> block(bsz * ni_total);
  size_t l = 0;
  std::string line;
  for (size_t t = 0; t < indicator_snp.size(); ++t) {
    if (!std::getline(infile, line)) break;
    if (indicator_snp[t] == 0) continue;
    char *save = nullptr;
    char *tok = strtok_r(&line[0], " ,\t", &save); // rs
    tok = strtok_r(nullptr, " ,\t", &save);          // allele
    tok = strtok_r(nullptr, " ,\t", &save);          // allele
    for (size_t i = 0; i < ni_total; ++i) {
      tok = strtok_r(nullptr, " ,\t", &save);
      if (!tok) return false;
      block[l * ni_total + i] = (strncmp(tok, "NA", 2) == 0) ? std::numeric_limits<double>::quiet_NaN() : atof(tok);
    }
    if (++l == bsz) {
      enforce_hip(gemma_hip_kin_add(GEMMA_GENO_F64_SNP_MAJOR, block.data(), l, ni_total), "BimbamKin");
      l = 0;
    }
  }
  if (l) enforce_hip(gemma_hip_kin_add(GEMMA_GENO_F64_SNP_MAJOR, block.data(), l, ni_total), "BimbamKin");
  size_t ns = 0;
  enforce_hip(gemma_hip_kin_end(matrix_kin->data, &ns), "BimbamKin");
  return true;
}

// Row formatter of the text writers (.assoc.txt, cXX / eigen matrices): the reference streams every field through ofstream with std::endl (a flush
// per SNP); here the same bytes -- `scientific << setprecision(6)` is printf's "%.6e", `fixed << setprecision(3)` is
// "%.3f" (libstdc++ formats through vsnprintf, so nan / inf spell the same) -- are formatted into per-thread buffers by
// write_rows() and written in row order.
class AssocLine {
public:
  AssocLine &str(const std::string &s) { buf_ += s; return *this; }
  AssocLine &tab() { buf_ += '\t'; return *this; }
  AssocLine &sci(double v) { return fmt("%.6e", v); }
  AssocLine &fix3(double v) { return fmt("%.3f", v); }
  AssocLine &g10(double v) { return fmt("%.10g", v); } // ostream default float format at precision(10)
  AssocLine &num(long v) { char t[32]; buf_.append(t, (size_t)snprintf(t, sizeof t, "%ld", v)); return *this; }
  AssocLine &unum(size_t v) { char t[32]; buf_.append(t, (size_t)snprintf(t, sizeof t, "%zu", v)); return *this; }
  void endl() { buf_ += '\n'; }
  const std::string &text() const { return buf_; }
  void reserve(size_t n) { buf_.reserve(n); }

private:
  AssocLine &fmt(const char *f, double v) { char t[64]; buf_.append(t, (size_t)snprintf(t, sizeof t, f, v)); return *this; }
  std::string buf_;
};

// rows 0..n-1 through row(ln, r), formatted by up to 16 host threads (GEMMA_HIP_IO_THREADS overrides), written in order
template <class RowFn> inline void write_rows(std::ofstream &out, size_t n, RowFn row, size_t approx_row_bytes = 96) {
  unsigned nt = std::thread::hardware_concurrency();
  if (const char *env = getenv("GEMMA_HIP_IO_THREADS"))
    if (atoi(env) > 0) nt = (unsigned)atoi(env);
  nt = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(nt ? nt : 4u, 16u),
                                                       std::min<size_t>(n, n * approx_row_bytes / (size_t(1) << 18) + 1)));
  std::vector<AssocLine> part(nt);
  auto work = [&](unsigned w) {
    part[w].reserve((n / nt + 1) * approx_row_bytes);
    for (size_t r = n * w / nt; r < n * (w + 1) / nt; ++r) row(part[w], r);
  };
  std::vector<std::thread> pool;
  for (unsigned w = 1; w < nt; ++w) pool.emplace_back(work, w);
  work(0);
  for (std::thread &th : pool) th.join();
  for (unsigned w = 0; w < nt; ++w) out.write(part[w].text().data(), (std::streamsize)part[w].text().size());
}

// PARAM::WriteMatrix / WriteVector, src/param.cpp:1886-1935: tab-separated text, precision(10)
inline bool WriteMatrix(const Matrix *M, const std::string &file_str) {
  std::ofstream outfile(file_str.c_str(), std::ofstream::out);
  if (!outfile) {
    std::cout << "error writing file: " << file_str << std::endl;
    return false;
  }
  // bands of rows, each formatted by the thread pool (about 64 MiB of text per band)
  const size_t band = std::max<size_t>(1, (size_t(64) << 20) / (14 * std::max<size_t>(M->size2, 1)));
  for (size_t i0 = 0; i0 < M->size1; i0 += band) {
    write_rows(outfile, std::min(band, M->size1 - i0), [&](AssocLine &ln, size_t r) {
      const double *row = M->data + (i0 + r) * M->tda;
      for (size_t j = 0; j < M->size2; ++j) {
        if (j) ln.tab();
        ln.g10(row[j]);
      }
      ln.endl();
    }, 14 * M->size2);
  }
  return true;
}
inline bool WriteVector(const Vector *v, const std::string &file_str) {
  std::ofstream outfile(file_str.c_str(), std::ofstream::out);
  if (!outfile) {
    std::cout << "error writing file: " << file_str << std::endl;
    return false;
  }
  outfile.precision(10);
  for (size_t i = 0; i < v->size; ++i) outfile << v->data[i * v->stride] << std::endl;
  return true;
}

// ReadFile_kin (k_mode == 1), src/gemma_io.cpp:1186-1243: dense ni_total x ni_total text; rows/columns of
// non-analysed individuals are dropped.  error is set exactly where the reference fails.
inline void ReadFile_kin(const std::string &file_kin, std::vector<int> &indicator_idv, bool &error, Matrix *G) {
  std::ifstream infile(file_kin.c_str());
  if (!infile) {
    std::cout << "error! fail to open kinship file: " << file_kin << std::endl;
    error = true;
    return;
  }
  const size_t ni_total = indicator_idv.size();
  for (size_t i = 0; i < G->size1; ++i)
    for (size_t j = 0; j < G->size2; ++j) G->data[i * G->tda + j] = 0.0;
  std::string line;
  size_t i_test = 0, i_total = 0;
  while (std::getline(infile, line)) {
    if (i_total == ni_total) {
      std::cout << "number of rows in the kinship file is larger than the number of phenotypes" << std::endl;
      error = true;
      return;
    }
    if (indicator_idv[i_total] == 0) {
      i_total++;
      continue;
    }
    size_t j_total = 0, j_test = 0;
    char *save = nullptr;
    for (char *tok = strtok_r(&line[0], " ,\t", &save); tok; tok = strtok_r(nullptr, " ,\t", &save)) {
      if (j_total == ni_total) {
        error = true;
        return;
      }
      const double d = atof(tok);
      if (indicator_idv[j_total] == 1) G->data[i_test * G->tda + j_test++] = d;
      j_total++;
    }
    if (j_total != ni_total) {
      std::cout << "number of columns in the kinship file does not match the number of individuals for row = "
                << i_total << std::endl;
      error = true;
      return;
    }
    i_total++;
    i_test++;
  }
  if (i_total != ni_total) {
    std::cout << "number of rows in the kinship file does not match the number of individuals." << std::endl;
    error = true;
  }
}

// null model: CalcLambda(func, eval, UtW, Uty, ...) src/lmm.cpp:2143-2180 + CalcPve :2183-2205
struct NullModel {
  double l_mle_null, logl_mle_H0, l_remle_null, logl_remle_H0, pve_null, pve_se_null, vg_remle_null, ve_remle_null;
};
inline NullModel CalcLambdaNull(const Vector *eval, const Matrix *UtW, const Vector *Uty, double l_min, double l_max,
                                size_t n_region, double trace_G) {
  if (UtW->tda != UtW->size2 || eval->stride != 1 || Uty->stride != 1)
    throw HipError(GEMMA_HIP_EINVAL, "CalcLambdaNull: contiguous inputs");
  double o[8];
  enforce_hip(gemma_hip_lmm_null(UtW->size1, UtW->size2, eval->data, UtW->data, Uty->data, l_min, l_max, n_region,
                                 trace_G, o),
              "CalcLambda (null)");
  return NullModel{o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
}

// class LMM, src/lmm.h:49-125 -- the members CopyFromParam fills (src/lmm.cpp:56-90) and the drivers
class LMM {
public:
  int a_mode = 1;
  size_t d_pace = 100000;
  std::string file_bfile, file_geno, file_gene, file_out, path_out = "./output/";
  double l_min = 1e-5, l_max = 1e5;
  size_t n_region = 10;
  double l_mle_null = 0.0, logl_mle_H0 = 0.0;
  size_t ni_total = 0, ni_test = 0, n_cvt = 1;
  double time_UtX = 0.0, time_opt = 0.0;
  std::vector<int> indicator_idv, indicator_snp;
  std::vector<SNPINFO> snpInfo;
  std::vector<SUMSTAT> sumStat;
  std::set<std::string> setGWASnps; // -loco / -gwasnps: the SNPs tested (src/lmm.cpp:87,1585-1587); empty = all

  int shard_rank = 0, shard_world = 1; // multi-GPU: this process analyses (and writes) its contiguous share of the SNPs
  bool kept_U = false;                 // U / eval are the library's kept ones (EigenDecompKept / kept_bcast): U->data is not read

  // the SNPs Analyze visits: indicator_snp, with -loco only the members of setGWASnps (src/lmm.cpp:1578-1587), and of
  // those the share of this rank
  std::vector<int> analysed_snps(bool this_rank_only = true) const {
    std::vector<int> keep(indicator_snp);
    if (!setGWASnps.empty())
      for (size_t t = 0; t < keep.size() && t < snpInfo.size(); ++t)
        if (keep[t] && setGWASnps.count(snpInfo[t].rs_number) == 0) keep[t] = 0;
    if (this_rank_only) shard_keep(keep, shard_rank, shard_world);
    return keep;
  }

  // AnalyzePlink prints the PREVIOUS SNP's beta / se for a SNP whose lambda search failed (function-scope variables,
  // src/lmm.cpp:1725,1870-1884), so a shard must start with the carry the unsharded run has at its first SNP.  A
  // successful SNP overwrites the library's carry with its own values and a failed one leaves it alone: analysing the
  // analysed SNPs BEFORE the shard, backwards, one at a time and with the results discarded, until one succeeds (almost
  // always the first) leaves exactly that state.  (gemma_amd/dist.py: seed_plink_carry is the same rule.)
  void seed_plink_carry(std::ifstream &infile, const std::vector<int> &keep_mine, size_t n_bit) {
    const std::vector<int> all = analysed_snps(false);
    size_t first = 0;
    while (first < keep_mine.size() && !keep_mine[first]) ++first;
    std::vector<unsigned char> row(n_bit);
    gemma_sumstat o;
    for (size_t t = first; t-- > 0;) {
      if (!all[t]) continue;
      infile.seekg((std::streamoff)(3 + t * n_bit));
      infile.read(reinterpret_cast<char *>(row.data()), (std::streamsize)n_bit);
      if ((size_t)infile.gcount() != n_bit) throw std::runtime_error("error reading genotype (.bed) file (truncated)");
      enforce_hip(gemma_hip_lmm_batch(GEMMA_GENO_PLINK_2BIT, row.data(), 1, n_bit, &o), "AnalyzePlink (carry)");
      if (o.logl_H1 == o.logl_H1) break; // not NaN: this SNP's beta / se are the carry now
    }
    infile.clear();
  }

  // AnalyzePlink, src/lmm.cpp:1710-1903: raw .bed rows go to the device (decode, drop, impute there)
  void AnalyzePlink(const Matrix *U, const Vector *eval, const Matrix *UtW, const Vector *Uty) {
    const std::string file_bed = file_bfile + ".bed";
    std::ifstream infile(file_bed.c_str(), std::ios::binary);
    if (!infile) throw std::runtime_error("error reading genotype (.bed) file");
    setup(U, eval, UtW, Uty, 1);
    enforce_hip(gemma_hip_lmm_set_indicator(indicator_idv.data(), indicator_idv.size()), "AnalyzePlink");
    const size_t n_bit = (ni_total + 3) / 4;
    const size_t B = io_block_rows(LMM_BATCH_SIZE);
    std::vector<gemma_sumstat> out(B);
    size_t t_next = 0;
    const std::vector<int> keep = analysed_snps();
    if (shard_world > 1 && shard_rank > 0 && a_mode == 1) seed_plink_carry(infile, keep, n_bit);
    // Three stages in flight: a host thread reads the .bed rows of block k + 2 (BlockPrefetch), block k + 1 crosses PCIe
    // from a pinned staging slot, block k computes (gemma_hip_lmm_batch_submit / _collect: two blocks in the library)
    BlockPrefetch pf(B * n_bit, [&](void *slot, int) {
      return read_bed_rows(infile, keep, t_next, n_bit, static_cast<unsigned char *>(slot), B);
    });
    size_t in_flight = 0;
    bool more = true;
    while (more || in_flight) {
      if (more && in_flight < 2) {
        void *slot = nullptr;
        const size_t l = pf.next(slot);
        if (l == (size_t)-1) throw std::runtime_error("error reading genotype (.bed) file (truncated)");
        if (l == 0) {
          more = false;
        } else {
          enforce_hip(gemma_hip_lmm_batch_submit(GEMMA_GENO_PLINK_2BIT, slot, l, n_bit), "batch_compute");
          ++in_flight;
        }
        continue;
      }
      size_t l = 0;
      enforce_hip(gemma_hip_lmm_batch_collect(out.data(), &l), "batch_compute");
      --in_flight;
      for (size_t i = 0; i < l; ++i) {
        SUMSTAT st = {out[i].beta, out[i].se, out[i].lambda_remle, out[i].lambda_mle,
                      out[i].p_wald, out[i].p_lrt, out[i].p_score, out[i].logl_H1};
        sumStat.push_back(st);
      }
    }
    finish();
  }

  // AnalyzePlinkGXE, src/lmm.cpp:2427-2608 (`-gxe`): covariates [W, env, x_s], tested variable x_s . env; env over the
  // ni_test analysed individuals.  (The reference's BIMBAM twin opens file_gene instead of file_geno, :2289, and cannot run.)
  void AnalyzePlinkGXE(const Matrix *U, const Vector *eval, const Matrix *UtW, const Vector *Uty, const Vector *env) {
    const std::string file_bed = file_bfile + ".bed";
    std::ifstream infile(file_bed.c_str(), std::ios::binary);
    if (!infile) throw std::runtime_error("error reading bed file");
    setup(U, eval, UtW, Uty, 1);
    enforce_hip(gemma_hip_lmm_set_indicator(indicator_idv.data(), indicator_idv.size()), "AnalyzePlinkGXE");
    std::vector<double> e(env->size);
    for (size_t i = 0; i < env->size; ++i) e[i] = env->data[i * env->stride];
    enforce_hip(gemma_hip_lmm_set_env(e.data()), "AnalyzePlinkGXE");
    const size_t n_bit = (ni_total + 3) / 4, B = io_block_rows(LMM_BATCH_SIZE);
    std::vector<gemma_sumstat> out(B);
    size_t t_next = 0;
    const std::vector<int> keep = analysed_snps();
    BlockPrefetch pf(B * n_bit, [&](void *slot, int) {
      return read_bed_rows(infile, keep, t_next, n_bit, static_cast<unsigned char *>(slot), B);
    });
    for (;;) {
      void *slot = nullptr;
      const size_t l = pf.next(slot);
      if (l == (size_t)-1) throw std::runtime_error("error reading bed file (truncated)");
      if (l == 0) break;
      enforce_hip(gemma_hip_lmm_gxe_batch(GEMMA_GENO_PLINK_2BIT, slot, l, n_bit, out.data()), "AnalyzePlinkGXE");
      for (size_t i = 0; i < l; ++i) {
        SUMSTAT SNPs = {out[i].beta, out[i].se, out[i].lambda_remle, out[i].lambda_mle,
                        out[i].p_wald, out[i].p_lrt, out[i].p_score, out[i].logl_H1};
        sumStat.push_back(SNPs);
      }
    }
    finish();
  }

  // LMM::Analyze with a caller-supplied SNP-major block source (the fetch_snp closure of src/lmm.cpp:1675-1700
  // factored out): X rows = analysed SNPs over the ni_test analysed individuals, NaN = missing
  void AnalyzeRows(const Matrix *U, const Vector *eval, const Matrix *UtW, const Vector *Uty, const double *X,
                   size_t n_snps, size_t ld) {
    setup(U, eval, UtW, Uty, 0);
    std::vector<gemma_sumstat> out(LMM_BATCH_SIZE);
    for (size_t s0 = 0; s0 < n_snps; s0 += LMM_BATCH_SIZE) {
      const size_t l = std::min(LMM_BATCH_SIZE, n_snps - s0);
      batch_compute(GEMMA_GENO_F64_SNP_MAJOR, X + s0 * ld, l, ld, out);
    }
    finish();
  }

  // LMM::Analyze with a pull-style block source (what fetch_snp, src/lmm.cpp:1675-1700, is to the reference): feed(X)
  // points X at the next block (ld doubles per row, rows = analysed SNPs over the ni_test analysed individuals,
  // NaN = missing, at most max_rows rows, valid until the next call) and returns its row count, 0 at the end.
  // include/gemma_io_host.hpp feeds it from a BIMBAM text file parsed on a pool of host threads, one block ahead.
  typedef std::function<size_t(const double *&X)> RowFeeder;
  void AnalyzeFeed(const Matrix *U, const Vector *eval, const Matrix *UtW, const Vector *Uty, RowFeeder &feed,
                   size_t max_rows, size_t ld) {
    setup(U, eval, UtW, Uty, 0);
    std::vector<gemma_sumstat> out(max_rows);
    for (;;) {
      const double *X = nullptr;
      const size_t l = feed(X);
      if (l == 0) break;
      if (l > max_rows) throw HipError(GEMMA_HIP_EINVAL, "AnalyzeFeed: block larger than max_rows");
      batch_compute(GEMMA_GENO_F64_SNP_MAJOR, X, l, ld, out);
    }
    finish();
  }

  // AnalyzeGene with a pull-style block source (rows = genes over the ni_test analysed individuals)
  void AnalyzeGeneFeed(const Matrix *U, const Vector *eval, const Matrix *UtW, const Vector *Utx, RowFeeder &feed,
                       size_t max_rows, size_t ld) {
    setup(U, eval, UtW, Utx, 0);
    std::vector<gemma_sumstat> out(max_rows);
    for (;;) {
      const double *Y = nullptr;
      const size_t l = feed(Y);
      if (l == 0) break;
      if (l > max_rows) throw HipError(GEMMA_HIP_EINVAL, "AnalyzeGeneFeed: block larger than max_rows");
      enforce_hip(gemma_hip_lmm_gene_batch(Y, l, ld, out.data()), "AnalyzeGene");
      for (size_t i = 0; i < l; ++i) {
        SUMSTAT SNPs = {out[i].beta, out[i].se, out[i].lambda_remle, out[i].lambda_mle,
                        out[i].p_wald, out[i].p_lrt, out[i].p_score, out[i].logl_H1};
        sumStat.push_back(SNPs);
      }
    }
    finish();
  }

  // AnalyzeGene, src/lmm.cpp:1365-1471 (the file reader factored out): Y rows = one phenotype (gene) each over the
  // ni_test analysed individuals, Utx = the rotated fixed tested variable; every row gets its own null fit
  void AnalyzeGeneRows(const Matrix *U, const Vector *eval, const Matrix *UtW, const Vector *Utx, const double *Y,
                       size_t n_genes, size_t ld) {
    setup(U, eval, UtW, Utx, 0);
    std::vector<gemma_sumstat> out(LMM_BATCH_SIZE);
    for (size_t s0 = 0; s0 < n_genes; s0 += LMM_BATCH_SIZE) {
      const size_t l = std::min(LMM_BATCH_SIZE, n_genes - s0);
      enforce_hip(gemma_hip_lmm_gene_batch(Y + s0 * ld, l, ld, out.data()), "AnalyzeGene");
      for (size_t i = 0; i < l; ++i) {
        SUMSTAT SNPs = {out[i].beta, out[i].se, out[i].lambda_remle, out[i].lambda_mle,
                        out[i].p_wald, out[i].p_lrt, out[i].p_score, out[i].logl_H1};
        sumStat.push_back(SNPs);
      }
    }
    finish();
  }

  // AnalyzeBimbamGXE, src/lmm.cpp:2283-2425 (the file reader factored out): X rows = analysed SNPs over the ni_test
  // analysed individuals (NaN = missing), env over the same individuals
  void AnalyzeGXERows(const Matrix *U, const Vector *eval, const Matrix *UtW, const Vector *Uty, const Vector *env,
                      const double *X, size_t n_snps, size_t ld) {
    setup(U, eval, UtW, Uty, 0);
    std::vector<double> e(env->size);
    for (size_t i = 0; i < env->size; ++i) e[i] = env->data[i * env->stride];
    enforce_hip(gemma_hip_lmm_set_env(e.data()), "AnalyzeGXE");
    std::vector<gemma_sumstat> out(LMM_BATCH_SIZE);
    for (size_t s0 = 0; s0 < n_snps; s0 += LMM_BATCH_SIZE) {
      const size_t l = std::min(LMM_BATCH_SIZE, n_snps - s0);
      enforce_hip(gemma_hip_lmm_gxe_batch(GEMMA_GENO_F64_SNP_MAJOR, X + s0 * ld, l, ld, out.data()), "AnalyzeGXE");
      for (size_t i = 0; i < l; ++i) {
        SUMSTAT SNPs = {out[i].beta, out[i].se, out[i].lambda_remle, out[i].lambda_mle,
                        out[i].p_wald, out[i].p_lrt, out[i].p_score, out[i].logl_H1};
        sumStat.push_back(SNPs);
      }
    }
    finish();
  }

  // WriteFiles, src/lmm.cpp:101-225
  void WriteFiles() {
    const std::string file_str = path_out + "/" + file_out + ".assoc.txt";
    std::ofstream outfile(file_str.c_str(), std::ofstream::out);
    if (!outfile) {
      std::cout << "error writing file: " << file_str << std::endl;
      return;
    }
    const bool gene = !file_gene.empty(); // src/lmm.cpp:172-179: one id column instead of the seven SNP columns
    if (shard_rank == 0) outfile << (gene ? "geneID\t" : "chr\trs\tps\tn_miss\tallele1\tallele0\taf\t");
    switch (shard_rank == 0 ? a_mode : -1) { // the parts of ranks > 0 are appended to rank 0's file: no header
    case 1: outfile << "beta\tse\tlogl_H1\tl_remle\tp_wald" << std::endl; break;
    case 2: outfile << "logl_H1\tl_mle\tp_lrt" << std::endl; break;
    case 3: outfile << "beta\tse\tp_score" << std::endl; break;
    case 4: outfile << "beta\tse\tlogl_H1\tl_remle\tl_mle\tp_wald\tp_lrt\tp_score" << std::endl; break;
    case 9: outfile << "beta\tse\tl_mle\tp_lrt" << std::endl; break;
    }
    std::vector<size_t> rows; // snpInfo index of the t-th record of sumStat
    const std::vector<int> keep = analysed_snps(); // indicator_snp, setGWASnps (src/lmm.cpp:208-210), this rank's share
    for (size_t i = 0; i < snpInfo.size(); ++i)
      if (gene || (i < keep.size() && keep[i])) rows.push_back(i);
    write_rows(outfile, std::min(rows.size(), sumStat.size()), [&](AssocLine &ln, size_t t) {
      const SNPINFO &s = snpInfo[rows[t]];
      const SUMSTAT &st = sumStat[t];
      if (gene)
        ln.str(s.rs_number).tab();
      else
        ln.str(s.chr).tab().str(s.rs_number).tab().num(s.base_position).tab().unum(s.n_miss).tab().str(s.a_minor).tab()
            .str(s.a_major).tab().fix3(s.maf).tab();
      switch (a_mode) {
      case 1: ln.sci(st.beta).tab().sci(st.se).tab().sci(st.logl_H1).tab().sci(st.lambda_remle).tab().sci(st.p_wald); break;
      case 2: ln.sci(st.logl_H1).tab().sci(st.lambda_mle).tab().sci(st.p_lrt); break;
      case 3: ln.sci(st.beta).tab().sci(st.se).tab().sci(st.p_score); break;
      case 4:
        ln.sci(st.beta).tab().sci(st.se).tab().sci(st.logl_H1).tab().sci(st.lambda_remle).tab().sci(st.lambda_mle).tab()
            .sci(st.p_wald).tab().sci(st.p_lrt).tab().sci(st.p_score);
        break;
      case 9: ln.sci(st.beta).tab().sci(st.se).tab().sci(st.lambda_mle).tab().sci(st.p_lrt); break;
      }
      ln.endl();
    });
  }

private:
  void setup(const Matrix *U, const Vector *eval, const Matrix *UtW, const Vector *Uty, int plink) {
    if ((!kept_U && U->tda != U->size2) || UtW->tda != UtW->size2 || eval->stride != 1 || Uty->stride != 1)
      throw HipError(GEMMA_HIP_EINVAL, "LMM: contiguous U/UtW/eval/Uty required");
    ni_test = U->size1;
    n_cvt = UtW->size2;
    gemma_lmm_cfg cfg;
    cfg.a_mode = a_mode; cfg.n = ni_test; cfg.n_cvt = n_cvt; cfg.l_min = l_min; cfg.l_max = l_max;
    cfg.n_region = n_region; cfg.l_mle_null = l_mle_null; cfg.logl_mle_H0 = logl_mle_H0; cfg.plink_nan_rule = plink;
    if (kept_U)
      enforce_hip(gemma_hip_lmm_setup_kept(&cfg, UtW->data, Uty->data), "LMM::Analyze");
    else
      enforce_hip(gemma_hip_lmm_setup(&cfg, U->data, eval->data, UtW->data, Uty->data), "LMM::Analyze");
    sumStat.clear();
  }
  // batch_compute, src/lmm.cpp:1513-1564
  void batch_compute(int kind, const void *geno, size_t l, size_t ld, std::vector<gemma_sumstat> &out) {
    if (l == 0) return;
    enforce_hip(gemma_hip_lmm_batch(kind, geno, l, ld, out.data()), "batch_compute");
    for (size_t i = 0; i < l; ++i) {
      SUMSTAT s = {out[i].beta, out[i].se, out[i].lambda_remle, out[i].lambda_mle,
                   out[i].p_wald, out[i].p_lrt, out[i].p_score, out[i].logl_H1};
      sumStat.push_back(s);
    }
  }
  void finish() { enforce_hip(gemma_hip_lmm_finish(&time_UtX, &time_opt), "LMM::Analyze"); }
};

// class LM, src/lm.h -- `-lm 1..4` (a_mode 51..54): per-SNP ordinary regression of y on (W, x), no kinship.
// AnalyzePlink / AnalyzeBimbam (src/lm.cpp:382-640) and WriteFiles (:83-223, the SNP branch; note its "n_mis" / "n_obs"
// header, which the LMM writer does not have).
class LM {
public:
  int a_mode = 51;
  std::string file_bfile, file_geno, file_out, path_out = "./output/";
  size_t ni_total = 0, ni_test = 0;
  std::vector<int> indicator_idv, indicator_snp;
  std::vector<SNPINFO> snpInfo;
  std::vector<SUMSTAT> sumStat;
  int shard_rank = 0, shard_world = 1; // as in class LMM
  std::vector<int> analysed_snps() const {
    std::vector<int> keep(indicator_snp);
    shard_keep(keep, shard_rank, shard_world);
    return keep;
  }

  // W: ni_test x n_cvt, y: ni_test (the analysed individuals)
  void AnalyzePlink(const Matrix *W, const Vector *y) {
    const std::string file_bed = file_bfile + ".bed";
    std::ifstream infile(file_bed.c_str(), std::ios::binary);
    if (!infile) throw std::runtime_error("error reading bed file");
    setup(W, y);
    enforce_hip(gemma_hip_lmm_set_indicator(indicator_idv.data(), indicator_idv.size()), "LM::AnalyzePlink");
    const size_t n_bit = (ni_total + 3) / 4, B = io_block_rows(LMM_BATCH_SIZE);
    std::vector<gemma_sumstat> out(B);
    size_t t_next = 0;
    const std::vector<int> keep = analysed_snps();
    BlockPrefetch pf(B * n_bit, [&](void *slot, int) {
      return read_bed_rows(infile, keep, t_next, n_bit, static_cast<unsigned char *>(slot), B);
    });
    for (;;) {
      void *slot = nullptr;
      const size_t l = pf.next(slot);
      if (l == (size_t)-1) throw std::runtime_error("error reading bed file (truncated)");
      if (l == 0) break;
      batch(GEMMA_GENO_PLINK_2BIT, slot, l, n_bit, out);
    }
    enforce_hip(gemma_hip_lm_finish(), "LM::AnalyzePlink");
  }

  // pull-style block source as in LMM::AnalyzeFeed (rows = analysed SNPs over the analysed individuals, NaN = missing)
  void AnalyzeFeed(const Matrix *W, const Vector *y, LMM::RowFeeder &feed, size_t max_rows, size_t ld) {
    setup(W, y);
    std::vector<gemma_sumstat> out(max_rows);
    for (;;) {
      const double *X = nullptr;
      const size_t l = feed(X);
      if (l == 0) break;
      if (l > max_rows) throw HipError(GEMMA_HIP_EINVAL, "LM::AnalyzeFeed: block larger than max_rows");
      batch(GEMMA_GENO_F64_SNP_MAJOR, X, l, ld, out);
    }
    enforce_hip(gemma_hip_lm_finish(), "LM::AnalyzeBimbam");
  }

  // LM::WriteFiles, src/lm.cpp:83-223 (SNP branch)
  void WriteFiles() {
    const std::string file_str = path_out + "/" + file_out + ".assoc.txt";
    std::ofstream outfile(file_str.c_str(), std::ofstream::out);
    if (!outfile) {
      std::cout << "error writing file: " << file_str << std::endl;
      return;
    }
    if (shard_rank == 0) outfile << "chr\trs\tps\tn_mis\tn_obs\tallele1\tallele0\taf\t";
    switch (shard_rank == 0 ? a_mode : -1) {
    case 51: outfile << "beta\tse\tp_wald" << std::endl; break;
    case 52: outfile << "p_lrt" << std::endl; break;
    case 53: outfile << "beta\tse\tp_score" << std::endl; break;
    case 54: outfile << "beta\tse\tp_wald\tp_lrt\tp_score" << std::endl; break;
    }
    std::vector<size_t> rows;
    const std::vector<int> keep = analysed_snps();
    for (size_t i = 0; i < snpInfo.size() && i < keep.size(); ++i)
      if (keep[i] != 0) rows.push_back(i);
    write_rows(outfile, std::min(rows.size(), sumStat.size()), [&](AssocLine &ln, size_t t) {
      const SNPINFO &s = snpInfo[rows[t]];
      const SUMSTAT &st = sumStat[t];
      ln.str(s.chr).tab().str(s.rs_number).tab().num(s.base_position).tab().unum(s.n_miss).tab().unum(ni_test - s.n_miss).tab()
          .str(s.a_minor).tab().str(s.a_major).tab().fix3(s.maf).tab();
      switch (a_mode) {
      case 51: ln.sci(st.beta).tab().sci(st.se).tab().sci(st.p_wald); break;
      case 52: ln.sci(st.p_lrt); break;
      case 53: ln.sci(st.beta).tab().sci(st.se).tab().sci(st.p_score); break;
      case 54: ln.sci(st.beta).tab().sci(st.se).tab().sci(st.p_wald).tab().sci(st.p_lrt).tab().sci(st.p_score); break;
      }
      ln.endl();
    });
  }

private:
  void setup(const Matrix *W, const Vector *y) {
    if (W->tda != W->size2 || y->stride != 1 || y->size != W->size1)
      throw HipError(GEMMA_HIP_EINVAL, "LM: contiguous W (n x c) and y (n) required");
    ni_test = W->size1;
    enforce_hip(gemma_hip_lm_setup(a_mode, W->size1, W->size2, W->data, y->data), "LM::Analyze");
    sumStat.clear();
  }
  void batch(int kind, const void *geno, size_t l, size_t ld, std::vector<gemma_sumstat> &out) {
    enforce_hip(gemma_hip_lm_batch(kind, geno, l, ld, out.data()), "LM::Analyze");
    for (size_t i = 0; i < l; ++i) {
      SUMSTAT s = {out[i].beta, out[i].se, out[i].lambda_remle, out[i].lambda_mle,
                   out[i].p_wald, out[i].p_lrt, out[i].p_score, out[i].logl_H1};
      sumStat.push_back(s);
    }
  }
};

// class MVLMM, src/mvlmm.h:32-104 -- multivariate LMM (`-lmm m -n a b c ...`): the members CopyFromParam fills
// (src/mvlmm.cpp:51-90), AnalyzePlink / AnalyzeBimbam rows (:3418-3899 / :2972-3416; crt as -crt sets it) and WriteFiles (:117-210).
// sumStat holds MPHSUMSTAT (src/param.h:68-77) flat: per SNP beta[d], Vbeta[v], Vg[v], Ve[v], p_wald, p_lrt, p_score
// with v = d (d + 1) / 2 -- the record gemma_hip_mvlmm_batch writes.
class MVLMM {
public:
  int a_mode = 1;
  std::string file_bfile, file_geno, file_out, path_out = "./output/";
  double l_min = 1e-5, l_max = 1e5;
  size_t n_region = 10;
  size_t em_iter = 10000, nr_iter = 100; // src/param.cpp:94-107
  double em_prec = 1e-4, nr_prec = 1e-4, p_nr = 1e-3;
  size_t crt = 0; // PARAM::crt (-crt, src/gemma.cpp:1398-1399)
  size_t ni_total = 0, ni_test = 0, n_cvt = 1, n_ph = 0;
  double logl_remle_H0 = 0.0, logl_mle_H0 = 0.0, time_UtX = 0.0, time_opt = 0.0;
  gemma_mvlmm_null null_fit;
  std::vector<int> indicator_idv, indicator_snp;
  std::vector<SNPINFO> snpInfo;
  std::vector<double> sumStat;
  int shard_rank = 0, shard_world = 1; // as in class LMM
  bool kept_U = false;                 // as in class LMM
  std::vector<int> analysed_snps() const {
    std::vector<int> keep(indicator_snp);
    shard_keep(keep, shard_rank, shard_world);
    return keep;
  }
  size_t stride() const { return n_ph + 3 * (n_ph * (n_ph + 1) / 2) + 3; }

  // UtY: ni_test x n_ph row-major (the gsl_matrix the reference passes)
  void AnalyzePlink(const Matrix *U, const Vector *eval, const Matrix *UtW, const Matrix *UtY) {
    const std::string file_bed = file_bfile + ".bed";
    std::ifstream infile(file_bed.c_str(), std::ios::binary);
    if (!infile) throw std::runtime_error("error reading genotype (.bed) file");
    setup(U, eval, UtW, UtY, 1);
    const size_t n_bit = (ni_total + 3) / 4, B = io_block_rows(LMM_BATCH_SIZE);
    std::vector<double> out(B * stride());
    size_t t_next = 0;
    const std::vector<int> keep = analysed_snps();
    BlockPrefetch pf(B * n_bit, [&](void *slot, int) {
      return read_bed_rows(infile, keep, t_next, n_bit, static_cast<unsigned char *>(slot), B);
    });
    for (;;) {
      void *slot = nullptr;
      const size_t l = pf.next(slot);
      if (l == (size_t)-1) throw std::runtime_error("error reading genotype (.bed) file (truncated)");
      if (l == 0) break;
      enforce_hip(gemma_hip_mvlmm_batch(GEMMA_GENO_PLINK_2BIT, slot, l, n_bit, out.data()), "MVLMM::AnalyzePlink");
      sumStat.insert(sumStat.end(), out.begin(), out.begin() + l * stride());
    }
    enforce_hip(gemma_hip_lmm_finish(&time_UtX, &time_opt), "MVLMM::AnalyzePlink");
  }

  // MVLMM::AnalyzePlinkGXE, src/mvlmm.cpp:4416-4870 (`-gxe` with several phenotypes, src/gemma.cpp:2840-2846): the null model is
  // fitted on (W, env) -- UtWe = UtW with U^T env appended as its last column (:4492-4516) --, per SNP x o env is tested with
  // (W, env, x) as covariates.  env over the analysed individuals.
  void AnalyzePlinkGXE(const Matrix *U, const Vector *eval, const Matrix *UtW, const Matrix *UtWe, const Matrix *UtY, const Vector *env) {
    const std::string file_bed = file_bfile + ".bed";
    std::ifstream infile(file_bed.c_str(), std::ios::binary);
    if (!infile) throw std::runtime_error("error reading genotype (.bed) file");
    if (UtWe->size2 != UtW->size2 + 1 || UtWe->tda != UtWe->size2 || env->size != U->size1)
      throw HipError(GEMMA_HIP_EINVAL, "MVLMM::AnalyzePlinkGXE: UtWe must be UtW plus the U^T env column, env over the analysed individuals");
    std::vector<double> e(env->size);
    for (size_t i = 0; i < env->size; ++i) e[i] = env->data[i * env->stride];
    setup(U, eval, UtW, UtY, 1, UtWe, e.data());
    const size_t n_bit = (ni_total + 3) / 4, B = io_block_rows(LMM_BATCH_SIZE);
    std::vector<double> out(B * stride());
    size_t t_next = 0;
    const std::vector<int> keep = analysed_snps();
    BlockPrefetch pf(B * n_bit, [&](void *slot, int) {
      return read_bed_rows(infile, keep, t_next, n_bit, static_cast<unsigned char *>(slot), B);
    });
    for (;;) {
      void *slot = nullptr;
      const size_t l = pf.next(slot);
      if (l == (size_t)-1) throw std::runtime_error("error reading genotype (.bed) file (truncated)");
      if (l == 0) break;
      enforce_hip(gemma_hip_mvlmm_batch(GEMMA_GENO_PLINK_2BIT, slot, l, n_bit, out.data()), "MVLMM::AnalyzePlinkGXE");
      sumStat.insert(sumStat.end(), out.begin(), out.begin() + l * stride());
    }
    enforce_hip(gemma_hip_lmm_finish(&time_UtX, &time_opt), "MVLMM::AnalyzePlinkGXE");
  }

  // pull-style block source as in LMM::AnalyzeFeed (rows = analysed SNPs over the analysed individuals, NaN = missing):
  // MVLMM::AnalyzeBimbam, src/mvlmm.cpp:2972-3416, with the file reader factored out
  void AnalyzeFeed(const Matrix *U, const Vector *eval, const Matrix *UtW, const Matrix *UtY, LMM::RowFeeder &feed,
                   size_t max_rows, size_t ld) {
    setup(U, eval, UtW, UtY, 0);
    std::vector<double> out(max_rows * stride());
    for (;;) {
      const double *X = nullptr;
      const size_t l = feed(X);
      if (l == 0) break;
      if (l > max_rows) throw HipError(GEMMA_HIP_EINVAL, "MVLMM::AnalyzeFeed: block larger than max_rows");
      enforce_hip(gemma_hip_mvlmm_batch(GEMMA_GENO_F64_SNP_MAJOR, X, l, ld, out.data()), "MVLMM::AnalyzeBimbam");
      sumStat.insert(sumStat.end(), out.begin(), out.begin() + l * stride());
    }
    enforce_hip(gemma_hip_lmm_finish(&time_UtX, &time_opt), "MVLMM::AnalyzeBimbam");
  }

  // the same with the rows already in memory
  void AnalyzeRows(const Matrix *U, const Vector *eval, const Matrix *UtW, const Matrix *UtY, const double *X, size_t n_snps,
                   size_t ld) {
    setup(U, eval, UtW, UtY, 0);
    std::vector<double> out(std::min(n_snps, LMM_BATCH_SIZE) * stride());
    for (size_t s0 = 0; s0 < n_snps; s0 += LMM_BATCH_SIZE) {
      const size_t l = std::min(LMM_BATCH_SIZE, n_snps - s0);
      enforce_hip(gemma_hip_mvlmm_batch(GEMMA_GENO_F64_SNP_MAJOR, X + s0 * ld, l, ld, out.data()), "MVLMM::AnalyzeBimbam");
      sumStat.insert(sumStat.end(), out.begin(), out.begin() + l * stride());
    }
    enforce_hip(gemma_hip_lmm_finish(&time_UtX, &time_opt), "MVLMM::AnalyzeBimbam");
  }

  // MVLMM::WriteFiles, src/mvlmm.cpp:117-210
  void WriteFiles() {
    const std::string file_str = path_out + "/" + file_out + ".assoc.txt";
    std::ofstream outfile(file_str.c_str(), std::ofstream::out);
    if (!outfile) {
      std::cout << "error writing file: " << file_str << std::endl;
      return;
    }
    if (shard_rank == 0) {
      outfile << "chr\trs\tps\tn_miss\tallele1\tallele0\taf\t";
      for (size_t i = 0; i < n_ph; i++) outfile << "beta_" << i + 1 << "\t";
      for (size_t i = 0; i < n_ph; i++)
        for (size_t j = i; j < n_ph; j++) outfile << "Vbeta_" << i + 1 << "_" << j + 1 << "\t";
    }
    switch (shard_rank == 0 ? a_mode : -1) {
    case 1: outfile << "p_wald" << std::endl; break;
    case 2: outfile << "p_lrt" << std::endl; break;
    case 3: outfile << "p_score" << std::endl; break;
    case 4: outfile << "p_wald\tp_lrt\tp_score" << std::endl; break;
    }
    const size_t d = n_ph, v = d * (d + 1) / 2, st = stride();
    std::vector<size_t> rows;
    const std::vector<int> keep = analysed_snps();
    for (size_t i = 0; i < snpInfo.size() && i < keep.size(); ++i)
      if (keep[i] != 0) rows.push_back(i);
    write_rows(outfile, std::min(rows.size(), sumStat.size() / st), [&](AssocLine &ln, size_t t) {
      const SNPINFO &s = snpInfo[rows[t]];
      const double *r = &sumStat[t * st];
      ln.str(s.chr).tab().str(s.rs_number).tab().num(s.base_position).tab().unum(s.n_miss).tab().str(s.a_minor).tab()
          .str(s.a_major).tab().fix3(s.maf).tab();
      for (size_t k = 0; k < d + v; ++k) ln.sci(r[k]).tab(); // beta then Vbeta (upper triangle, row by row)
      const double p_wald = r[d + 3 * v], p_lrt = r[d + 3 * v + 1], p_score = r[d + 3 * v + 2];
      switch (a_mode) {
      case 1: ln.sci(p_wald); break;
      case 2: ln.sci(p_lrt); break;
      case 3: ln.sci(p_score); break;
      case 4: ln.sci(p_wald).tab().sci(p_lrt).tab().sci(p_score); break;
      }
      ln.endl();
    });
  }

private:
  // the null block (src/mvlmm.cpp:3056-3208) and the per-SNP loop's state
  // UtWe, env: the interaction test (the null model then has UtWe's covariates)
  void setup(const Matrix *U, const Vector *eval, const Matrix *UtW, const Matrix *UtY, int plink, const Matrix *UtWe = nullptr,
             const double *env = nullptr) {
    if ((!kept_U && U->tda != U->size2) || UtW->tda != UtW->size2 || UtY->tda != UtY->size2 || eval->stride != 1)
      throw HipError(GEMMA_HIP_EINVAL, "MVLMM: contiguous U/UtW/UtY/eval required");
    ni_test = U->size1;
    n_cvt = UtW->size2;
    n_ph = UtY->size2;
    const gemma_mvlmm_opt opt = {em_iter, nr_iter, em_prec, nr_prec, p_nr, crt, (size_t)(env ? 1 : 0)};
    const Matrix *Wn = env ? UtWe : UtW;
    enforce_hip(gemma_hip_mvlmm_null(ni_test, Wn->size2, n_ph, eval->data, Wn->data, UtY->data, l_min, l_max, n_region, &opt,
                                     &null_fit),
                "MVLMM (null model)");
    logl_remle_H0 = null_fit.logl_remle_H0;
    logl_mle_H0 = null_fit.logl_mle_H0;
    gemma_lmm_cfg cfg;
    cfg.a_mode = a_mode; cfg.n = ni_test; cfg.n_cvt = n_cvt; cfg.l_min = l_min; cfg.l_max = l_max;
    cfg.n_region = n_region; cfg.l_mle_null = 0; cfg.logl_mle_H0 = 0; cfg.plink_nan_rule = plink;
    std::vector<double> y0(ni_test); // the univariate Uty slot is not read by the multivariate path
    for (size_t i = 0; i < ni_test; ++i) y0[i] = UtY->data[i * UtY->tda];
    if (kept_U)
      enforce_hip(gemma_hip_lmm_setup_kept(&cfg, UtW->data, y0.data()), "MVLMM::Analyze");
    else
      enforce_hip(gemma_hip_lmm_setup(&cfg, U->data, eval->data, UtW->data, y0.data()), "MVLMM::Analyze");
    if (plink) enforce_hip(gemma_hip_lmm_set_indicator(indicator_idv.data(), indicator_idv.size()), "MVLMM::Analyze");
    if (env) enforce_hip(gemma_hip_lmm_set_env(env), "MVLMM::Analyze (gxe)");
    enforce_hip(gemma_hip_mvlmm_set(n_ph, UtY->data, &null_fit, &opt), "MVLMM::Analyze");
    sumStat.clear();
  }
};

} // namespace gemma_amd
#endif
