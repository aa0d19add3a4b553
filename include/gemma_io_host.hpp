// gemma_io_host.hpp -- the data formats either side of the hot path: the
// readers that decide WHICH individuals and SNPs the device sees, the genotype feeders, and the artefacts the
// reference leaves between runs.  Header-only C++11 over include/gemma_host.hpp and the C ABI.
//
//   phenotypes / covariates / annotation : ReadFile_pheno, ReadFile_fam, ReadFile_cvt, ReadFile_bim, ReadFile_anno
//                                           (src/gemma_io.cpp:280-637), ProcessCvtPhen / CheckCvt / CopyCvtPhen
//                                           (src/param.cpp:1937-2198)
//   first pass over the genotypes         : ReadFile_bed  (src/gemma_io.cpp:876-1064)  -- .bed rows go to the device
//                                           ReadFile_geno (src/gemma_io.cpp:639-873)   -- text rows are parsed by a
//                                           pool of host threads (BimbamReader), statistics and filters on the device
//                                           (gemma_hip_snp_qc)
//   feeders                               : BimbamKin / AnalyzeBimbam over BimbamReader (src/gemma_io.cpp:1418-1597,
//                                           src/lmm.cpp:1660-1706)
//   -eigen / -d -u artefacts              : WriteEigen (src/gemma.cpp:1779-1800 -> PARAM::WriteMatrix / WriteVector),
//                                           ReadFile_eigenU / ReadFile_eigenD (src/gemma_io.cpp:1323-1416)
//
// Nothing here computes on the host what the reference computes per SNP: counting, imputation and the filters run
// in gemma_hip_snp_qc; the host side only tokenises text and moves bytes.  Text -> double conversion must give the
// very double atof() gives (the reference's readers call atof on every token): parse_double() takes the exact
// fast path (<= 19 digits collected in an integer below 2^53, power of ten <= 10^22: one correctly rounded
// multiplication or division, Clinger 1990) and hands everything else to strtod.
#ifndef GEMMA_IO_HOST_HPP
#define GEMMA_IO_HOST_HPP

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <set>
#include <thread>

#include <zlib.h>

#include "gemma_host.hpp"

namespace gemma_amd {

// ---------------------------------------------------------------------------------------------------------------
// text -> double, identical to atof
// ---------------------------------------------------------------------------------------------------------------
inline double parse_double(const char *s, const char *end) {
  static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char *p = s;
  bool neg = false;
  if (p < end && (*p == '-' || *p == '+')) neg = (*p++ == '-');
  uint64_t w = 0;
  int nd = 0, frac = 0;
  bool any = false;
  while (p < end && *p >= '0' && *p <= '9') {
    if (nd < 19) { w = w * 10 + (uint64_t)(*p - '0'); if (w) ++nd; } else goto slow;
    ++p; any = true;
  }
  if (p < end && *p == '.') {
    ++p;
    while (p < end && *p >= '0' && *p <= '9') {
      if (nd < 19) { w = w * 10 + (uint64_t)(*p - '0'); if (w) ++nd; ++frac; } else goto slow;
      ++p; any = true;
    }
  }
  if (!any) goto slow;
  {
    int e10 = 0;
    if (p < end && (*p == 'e' || *p == 'E')) {
      const char *q = p + 1;
      bool eneg = false;
      if (q < end && (*q == '-' || *q == '+')) eneg = (*q++ == '-');
      if (q >= end || *q < '0' || *q > '9') goto slow;
      int ev = 0;
      while (q < end && *q >= '0' && *q <= '9') { if (ev < 10000) ev = ev * 10 + (*q - '0'); ++q; }
      e10 = eneg ? -ev : ev;
      p = q;
    }
    if (p != end) goto slow; // trailing characters: let strtod decide where the number ends
    e10 -= frac;
    if (w > (uint64_t(1) << 53) || e10 < -22 || e10 > 22) goto slow;
    double v = (double)w;
    if (e10 < 0) v /= p10[-e10]; else v *= p10[e10];
    return neg ? -v : v;
  }
slow : {
  char buf[64];
  const size_t len = (size_t)(end - s);
  if (len < sizeof(buf)) {
    memcpy(buf, s, len);
    buf[len] = 0;
    return strtod(buf, nullptr);
  }
  return strtod(std::string(s, end).c_str(), nullptr);
}
}

// ---------------------------------------------------------------------------------------------------------------
// plain or gzip-compressed text, line by line (the reference's igzstream + safeGetline, src/gemma_io.cpp:118-151:
// "\n", "\r\n" and "\r" all end a line)
// ---------------------------------------------------------------------------------------------------------------
class TextFile {
public:
  explicit TextFile(const std::string &path) : f_(gzopen(path.c_str(), "rb")), buf_(1 << 20), pos_(0), len_(0) {
    if (f_) gzbuffer(f_, 1 << 20);
  }
  ~TextFile() { if (f_) gzclose(f_); }
  TextFile(const TextFile &) = delete;
  TextFile &operator=(const TextFile &) = delete;
  bool ok() const { return f_ != nullptr; }
  // false at end of file with nothing read (safeGetline(...).eof() on an empty tail)
  bool getline(std::string &line) {
    line.clear();
    bool got = false;
    for (;;) {
      if (pos_ == len_) {
        const int r = f_ ? gzread(f_, buf_.data(), (unsigned)buf_.size()) : 0;
        if (r <= 0) return got;
        len_ = (size_t)r;
        pos_ = 0;
      }
      const char *b = buf_.data() + pos_, *e = buf_.data() + len_;
      const char *q = b;
      while (q < e && *q != '\n' && *q != '\r') ++q;
      line.append(b, q);
      got = got || q > b;
      pos_ = (size_t)(q - buf_.data());
      if (q == e) continue;
      ++pos_;
      if (*q == '\r') { // swallow the "\n" of "\r\n"
        if (pos_ == len_) {
          const int r = gzread(f_, buf_.data(), (unsigned)buf_.size());
          len_ = r > 0 ? (size_t)r : 0;
          pos_ = 0;
        }
        if (pos_ < len_ && buf_[pos_] == '\n') ++pos_;
      }
      return true;
    }
  }

private:
  gzFile f_;
  std::vector<char> buf_;
  size_t pos_, len_;
};

namespace detail {
inline bool is_delim(char c, bool comma) { return c == ' ' || c == '\t' || (comma && c == ','); }
// strtok(line, " ,\t") without touching the buffer: next token [b, e) at or after p
inline bool next_token(const char *&p, const char *end, const char *&b, const char *&e, bool comma = true) {
  while (p < end && is_delim(*p, comma)) ++p;
  if (p >= end) return false;
  b = p;
  while (p < end && !is_delim(*p, comma)) ++p;
  e = p;
  return true;
}
inline bool tok_is_na(const char *b, const char *e) { return e - b == 2 && b[0] == 'N' && b[1] == 'A'; }
} // namespace detail

// ---------------------------------------------------------------------------------------------------------------
// phenotypes, covariates, annotation
// ---------------------------------------------------------------------------------------------------------------
// ReadFile_snps, src/gemma_io.cpp:153-178 (`-snps`): one SNP id per line (first token)
inline bool ReadFile_snps(const std::string &file_snps, std::set<std::string> &setSnps) {
  setSnps.clear();
  TextFile infile(file_snps);
  if (!infile.ok()) {
    std::cout << "error! fail to open snps file: " << file_snps << std::endl;
    return false;
  }
  std::string line;
  while (infile.getline(line)) {
    const char *p = line.data(), *end = p + line.size(), *b, *e;
    if (!detail::next_token(p, end, b, e)) {
      std::cout << "Problem reading SNP file" << std::endl;
      return false;
    }
    setSnps.insert(std::string(b, e));
  }
  return true;
}

// ReadFile_column, src/gemma_io.cpp:344-383: one column of a text file (the -gxe / -widv inputs); "NA" -> indicator 0
inline bool ReadFile_column(const std::string &file_pheno, std::vector<int> &indicator_idv, std::vector<double> &pheno,
                            const int &p_column) {
  indicator_idv.clear();
  pheno.clear();
  TextFile infile(file_pheno);
  if (!infile.ok()) {
    std::cout << "error! fail to open phenotype file: " << file_pheno << std::endl;
    return false;
  }
  std::string line;
  while (infile.getline(line)) {
    const char *p = line.data(), *end = p + line.size(), *b = nullptr, *e = nullptr;
    bool have = false;
    for (int i = 0; i < p_column; ++i) have = detail::next_token(p, end, b, e);
    if (!have) {
      std::cout << "Problem reading PHENO column" << std::endl;
      return false;
    }
    if (detail::tok_is_na(b, e)) {
      indicator_idv.push_back(0);
      pheno.push_back(-9);
    } else {
      indicator_idv.push_back(1);
      pheno.push_back(parse_double(b, e));
    }
  }
  return true;
}

// ReadFile_pheno, src/gemma_io.cpp:386-444: BIMBAM phenotype file, p_column = 1-based columns; "NA" -> indicator 0, -9
inline bool ReadFile_pheno(const std::string &file_pheno, std::vector<std::vector<int>> &indicator_pheno,
                           std::vector<std::vector<double>> &pheno, const std::vector<size_t> &p_column) {
  indicator_pheno.clear();
  pheno.clear();
  TextFile infile(file_pheno);
  if (!infile.ok()) {
    std::cout << "error! fail to open phenotype file: " << file_pheno << std::endl;
    return false;
  }
  const size_t p_max = *std::max_element(p_column.begin(), p_column.end());
  std::map<size_t, size_t> mapP2c;
  for (size_t i = 0; i < p_column.size(); i++) mapP2c[p_column[i]] = i;
  std::vector<double> pheno_row(p_column.size(), -9);
  std::vector<int> ind_pheno_row(p_column.size(), 0);
  std::string line;
  while (infile.getline(line)) {
    const char *p = line.data(), *end = p + line.size(), *b, *e;
    for (size_t i = 0; i < p_max; ++i) {
      if (!detail::next_token(p, end, b, e)) {
        std::cout << "Number of phenotypes in pheno file do not match phenotypes in geno file" << std::endl;
        return false;
      }
      std::map<size_t, size_t>::const_iterator it = mapP2c.find(i + 1);
      if (it == mapP2c.end()) continue;
      if (detail::tok_is_na(b, e)) {
        ind_pheno_row[it->second] = 0;
        pheno_row[it->second] = -9;
      } else {
        ind_pheno_row[it->second] = 1;
        pheno_row[it->second] = parse_double(b, e);
      }
    }
    indicator_pheno.push_back(ind_pheno_row);
    pheno.push_back(pheno_row);
  }
  return true;
}

// ReadFile_fam, src/gemma_io.cpp:559-635: phenotype = column 6 (+ further columns); "NA" and -9 are missing
inline bool ReadFile_fam(const std::string &file_fam, std::vector<std::vector<int>> &indicator_pheno,
                         std::vector<std::vector<double>> &pheno, std::map<std::string, int> &mapID2num,
                         const std::vector<size_t> &p_column) {
  indicator_pheno.clear();
  pheno.clear();
  mapID2num.clear();
  TextFile infile(file_fam);
  if (!infile.ok()) {
    std::cout << "error opening .fam file: " << file_fam << std::endl;
    return false;
  }
  const size_t p_max = *std::max_element(p_column.begin(), p_column.end());
  std::map<size_t, size_t> mapP2c;
  for (size_t i = 0; i < p_column.size(); i++) mapP2c[p_column[i]] = i;
  std::vector<double> pheno_row(p_column.size(), -9);
  std::vector<int> ind_pheno_row(p_column.size(), 0);
  std::string line;
  int c = 0;
  while (infile.getline(line)) {
    const char *p = line.data(), *end = p + line.size(), *b, *e;
    std::string id;
    for (int k = 0; k < 5; ++k) { // family, individual, father, mother, sex: " \t" only
      if (!detail::next_token(p, end, b, e, false)) {
        std::cout << "Parsing input file '" << file_fam << "' failed in function ReadFile_fam" << std::endl;
        return false;
      }
      if (k == 1) id.assign(b, e);
    }
    bool have = detail::next_token(p, end, b, e, false); // first phenotype: " \t"; the following ones " ,\t"
    for (size_t i = 0; i < p_max; ++i) {
      std::map<size_t, size_t>::const_iterator it = mapP2c.find(i + 1);
      if (it != mapP2c.end()) {
        if (!have) {
          std::cout << "Problem reading FAM file (phenotypes do not match geno file)" << std::endl;
          return false;
        }
        const double v = detail::tok_is_na(b, e) ? -9.0 : parse_double(b, e);
        if (detail::tok_is_na(b, e) || v == -9) {
          ind_pheno_row[it->second] = 0;
          pheno_row[it->second] = -9;
        } else {
          ind_pheno_row[it->second] = 1;
          pheno_row[it->second] = v;
        }
      }
      have = detail::next_token(p, end, b, e, true);
    }
    indicator_pheno.push_back(ind_pheno_row);
    pheno.push_back(pheno_row);
    mapID2num[id] = c++;
  }
  return true;
}

// ReadFile_cvt, src/gemma_io.cpp:446-511: a row with any "NA" is flagged 0; all flagged-1 rows must have n_cvt columns
inline bool ReadFile_cvt(const std::string &file_cvt, std::vector<int> &indicator_cvt,
                         std::vector<std::vector<double>> &cvt, size_t &n_cvt) {
  indicator_cvt.clear();
  TextFile infile(file_cvt);
  if (!infile.ok()) {
    std::cout << "error! fail to open covariates file: " << file_cvt << std::endl;
    return false;
  }
  std::string line;
  while (infile.getline(line)) {
    std::vector<double> v_d;
    int flag_na = 0;
    const char *p = line.data(), *end = p + line.size(), *b, *e;
    while (detail::next_token(p, end, b, e)) {
      if (detail::tok_is_na(b, e)) {
        flag_na = 1;
        v_d.push_back(-9);
      } else {
        v_d.push_back(parse_double(b, e));
      }
    }
    indicator_cvt.push_back(flag_na == 0 ? 1 : 0);
    cvt.push_back(v_d);
  }
  if (indicator_cvt.empty()) {
    n_cvt = 0;
  } else {
    int first = 0;
    for (size_t i = 0; i < indicator_cvt.size(); ++i) {
      if (indicator_cvt[i] == 0) continue;
      if (first == 0) {
        first = 1;
        n_cvt = cvt[i].size();
      }
      if (n_cvt != cvt[i].size()) {
        std::cout << "error! number of covariates in row " << i << " do not match other rows." << std::endl;
        return false;
      }
    }
  }
  return true;
}

// ReadFile_bim, src/gemma_io.cpp:514-556: chr rs cM bp minor major (" \t" separated)
inline bool ReadFile_bim(const std::string &file_bim, std::vector<SNPINFO> &snpInfo) {
  snpInfo.clear();
  TextFile infile(file_bim);
  if (!infile.ok()) {
    std::cout << "error opening .bim file: " << file_bim << std::endl;
    return false;
  }
  std::string line;
  while (infile.getline(line)) {
    const char *p = line.data(), *end = p + line.size(), *b[6], *e[6];
    for (int k = 0; k < 6; ++k)
      if (!detail::next_token(p, end, b[k], e[k], false)) {
        std::cout << "Parsing input file '" << file_bim << "' failed in function ReadFile_bim" << std::endl;
        return false;
      }
    SNPINFO s;
    s.chr.assign(b[0], e[0]);
    s.rs_number.assign(b[1], e[1]);
    s.cM = parse_double(b[2], e[2]);
    s.base_position = atol(std::string(b[3], e[3]).c_str());
    s.a_minor.assign(b[4], e[4]);
    s.a_major.assign(b[5], e[5]);
    s.n_miss = 0;
    s.missingness = -9;
    s.maf = -9;
    s.n_idv = 0;
    s.n_nb = 0;
    s.file_position = 0;
    snpInfo.push_back(s);
  }
  return true;
}

// ReadFile_anno, src/gemma_io.cpp:280-341: rs, bp, chr, cM ("NA" or absent -> -9)
inline bool ReadFile_anno(const std::string &file_anno, std::map<std::string, std::string> &mapRS2chr,
                          std::map<std::string, long int> &mapRS2bp, std::map<std::string, double> &mapRS2cM) {
  mapRS2chr.clear();
  mapRS2bp.clear();
  TextFile infile(file_anno);
  if (!infile.ok()) {
    std::cout << "error opening annotation file: " << file_anno << std::endl;
    return false;
  }
  std::string line;
  while (infile.getline(line)) {
    const char *p = line.data(), *end = p + line.size(), *b, *e;
    if (!detail::next_token(p, end, b, e)) {
      std::cout << line << " Bad RS format" << std::endl;
      return false;
    }
    const std::string rs(b, e);
    if (!detail::next_token(p, end, b, e)) {
      std::cout << line << " Bad format" << std::endl;
      return false;
    }
    const long b_pos = detail::tok_is_na(b, e) ? -9 : atol(std::string(b, e).c_str());
    if (b_pos == 0) {
      std::cout << line << " Bad pos format (is zero)" << std::endl;
      return false;
    }
    std::string chr = "-9";
    double cM = -9;
    if (detail::next_token(p, end, b, e)) {
      if (!detail::tok_is_na(b, e)) chr.assign(b, e);
      if (detail::next_token(p, end, b, e) && !detail::tok_is_na(b, e)) cM = parse_double(b, e);
    }
    mapRS2chr[rs] = chr;
    mapRS2bp[rs] = b_pos;
    mapRS2cM[rs] = cM;
  }
  return true;
}

// The individual-selection state of PARAM (src/param.h:225-262) and the three functions that fill it
struct CvtPhen {
  std::vector<std::vector<int>> indicator_pheno;
  std::vector<std::vector<double>> pheno;
  std::vector<int> indicator_cvt;
  std::vector<std::vector<double>> cvt;
  std::vector<int> indicator_gxe; // -gxe: environment variable per individual (src/param.cpp:228-233)
  std::vector<double> gxe;
  std::vector<int> indicator_idv;
  size_t n_cvt = 0, ni_test = 0;
  bool error = false;

  // PARAM::CheckCvt, src/param.cpp:1937-1990: constant columns count as intercepts; if every column is one the
  // covariates are dropped, if none is a column of 1s is appended
  void CheckCvt() {
    if (indicator_cvt.empty()) return;
    size_t flag_ipt = 0, n_remove = 0;
    for (size_t j = 0; j < n_cvt; ++j) {
      double v_min = std::numeric_limits<double>::infinity(), v_max = -v_min;
      for (size_t i = 0; i < indicator_idv.size(); ++i) {
        if (indicator_idv[i] == 0 || indicator_cvt[i] == 0) continue;
        v_min = std::min(v_min, cvt[i][j]);
        v_max = std::max(v_max, cvt[i][j]);
      }
      if (v_min == v_max) {
        flag_ipt = 1;
        ++n_remove;
      }
    }
    if (n_cvt == n_remove) {
      // the reference keeps the rows of cvt and later copies their FIRST column as the single covariate
      // (src/param.cpp:1974-1976, 2146-2170): a constant column other than 1 becomes the intercept as it stands
      indicator_cvt.clear();
      n_cvt = 1;
    } else if (flag_ipt == 0) {
      std::cerr << "**** INFO: no intercept term is found in the cvt file: a column of 1s is added." << std::endl;
      for (size_t i = 0; i < indicator_idv.size(); ++i) {
        if (indicator_idv[i] == 0 || indicator_cvt[i] == 0) continue;
        cvt[i].push_back(1.0);
      }
      n_cvt++;
    }
  }

  // PARAM::ProcessCvtPhen, src/param.cpp:1993-2098 (no -widv / subsampling here)
  void ProcessCvtPhen() {
    indicator_idv.clear();
    for (size_t i = 0; i < indicator_pheno.size(); i++) {
      int k = 1;
      for (size_t j = 0; j < indicator_pheno[i].size(); j++)
        if (indicator_pheno[i][j] == 0) k = 0;
      indicator_idv.push_back(k);
    }
    if (!indicator_cvt.empty())
      for (size_t i = 0; i < indicator_idv.size(); ++i) indicator_idv[i] *= indicator_cvt[i];
    if (!indicator_gxe.empty()) // individuals with a missing environment value leave the analysis (:2016-2020)
      for (size_t i = 0; i < indicator_idv.size(); ++i) indicator_idv[i] *= indicator_gxe[i];
    ni_test = 0;
    for (size_t i = 0; i < indicator_idv.size(); ++i) ni_test += indicator_idv[i] != 0;
    if (ni_test == 0) {
      error = true;
      std::cout << "error! number of analyzed individuals equals 0. " << std::endl;
    }
    if (!indicator_cvt.empty()) {
      CheckCvt();
    } else {
      cvt.assign(indicator_idv.size(), std::vector<double>(1, 1.0)); // no -c file: the intercept alone
      indicator_cvt.assign(indicator_idv.size(), 1);
      n_cvt = 1;
    }
  }

  // PARAM::CopyGxe, src/param.cpp:2116-2128
  void CopyGxe(std::vector<double> &env) const {
    env.clear();
    for (size_t i = 0; i < indicator_idv.size(); ++i)
      if (indicator_idv[i] != 0 && indicator_gxe[i] != 0) env.push_back(gxe[i]);
  }

  // PARAM::CopyCvtPhen (flag 0), src/param.cpp:2146-2198: W (ni_test x n_cvt), Y (ni_test x n_ph) of the analysed
  void CopyCvtPhen(std::vector<double> &W, std::vector<double> &Y) const {
    const size_t n_ph = pheno.empty() ? 0 : pheno[0].size();
    W.assign(ni_test * n_cvt, 0.0);
    Y.assign(ni_test * n_ph, 0.0);
    size_t ci = 0;
    for (size_t i = 0; i < indicator_idv.size(); ++i) {
      if (indicator_idv[i] == 0) continue;
      for (size_t j = 0; j < n_ph; ++j) Y[ci * n_ph + j] = pheno[i][j];
      for (size_t j = 0; j < n_cvt; ++j) W[ci * n_cvt + j] = cvt[i][j];
      ci++;
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// BIMBAM mean-genotype text -> fp64 SNP-major blocks on a pool of host threads
// ---------------------------------------------------------------------------------------------------------------
// One line = rs, allele, allele, then ni_total values ("NA" = missing -> NaN); separators " ,\t".  read_block() pulls
// the text of up to max_snps lines through zlib in large reads on the calling thread (decompression is sequential by
// nature), finds the line ends with memchr, and splits the lines over n_threads parsers that work in place on the
// text buffer and write their own rows of the block: every value costs one token scan and -- for the usual
// "0.123"-style dosages -- one integer accumulation and one division; no locale, no allocation, no copy of the line.
class BimbamReader {
public:
  struct Row {
    std::string rs, minor, major;
  };
  // skip_fields: leading tokens that are not values (3 for BIMBAM: rs, allele, allele; 0 for a plain matrix such as the
  // kinship file); strict_columns: a line with more than skip_fields + ni_total tokens is malformed
  BimbamReader(const std::string &path, size_t ni_total, unsigned n_threads = 0, int skip_fields = 3,
               bool strict_columns = false)
      : f_(gzopen(path.c_str(), "rb")), ni_total_(ni_total), n_threads_(n_threads ? n_threads : default_threads()),
        skip_(skip_fields), strict_(strict_columns), line_no_(0), beg_(0), end_(0), eof_(false), buf_(nullptr), cap_(0) {
    if (f_) gzbuffer(f_, 1 << 20);
  }
  ~BimbamReader() {
    if (f_) gzclose(f_);
    free(buf_);
  }
  BimbamReader(const BimbamReader &) = delete;
  BimbamReader &operator=(const BimbamReader &) = delete;
  bool ok() const { return f_ != nullptr; }
  size_t lines_read() const { return line_no_; }
  size_t text_buffer_bytes() const { return cap_; } // capacity of the text buffer (tests: stays O(chunk) over dropped lines)
  // consumes one line without counting it (the header of a gene-expression file); false at end of file
  bool skip_line() {
    if (!f_) return false;
    for (;;) {
      size_t b, e, next;
      if (next_line(beg_, b, e, next)) {
        beg_ = next;
        return true;
      }
      if (eof_) {
        const bool had = beg_ < end_;
        beg_ = end_;
        return had;
      }
      if (beg_ > 0) {
        memmove(buf_, buf_ + beg_, end_ - beg_);
        end_ -= beg_;
        beg_ = 0;
      }
      fill();
    }
  }
  static unsigned default_threads() {
    const char *env = getenv("GEMMA_HIP_IO_THREADS");
    if (env && atoi(env) > 0) return (unsigned)atoi(env);
    const unsigned hc = std::thread::hardware_concurrency();
    return hc ? std::min(hc, 32u) : 4u;
  }
  // Reads up to max_snps further lines (fewer when their text would pass TEXT_CAP bytes); lines whose `keep` entry
  // (indexed by file line) is 0 are consumed and dropped without being parsed.  X gets rows x ld doubles (ld >= number
  // of columns kept); cols (optional, ni_total ints) selects the individuals copied into a row, in order.  Returns the
  // number of rows produced, 0 at end of file, or (size_t)-1 on a malformed line (message printed, as the reference's
  // enforce would).
  size_t read_block(size_t max_snps, double *X, size_t ld, std::vector<Row> *rows = nullptr,
                    const std::vector<int> *keep = nullptr, const int *cols = nullptr) {
    if (!f_) return 0;
    // drop the text of the previous block, keep the unconsumed tail at the front
    if (beg_ > 0) {
      memmove(buf_, buf_ + beg_, end_ - beg_);
      end_ -= beg_;
      beg_ = 0;
    }
    spans_.clear();
    size_t scan = 0; // everything before `scan` has been split into lines
    while (spans_.size() < max_snps) {
      size_t b, e, next;
      if (!next_line(scan, b, e, next)) {
        if (eof_) {
          if (scan < end_) { // last line without a line end
            b = scan; e = end_; next = end_;
          } else {
            break;
          }
        } else {
          if (!spans_.empty() && end_ >= text_cap()) break;
          // nothing kept yet: the text before `scan` belongs to dropped lines only (a shard's skipped prefix, -loco /
          // -snps with the kept SNPs late in the file) -- discard it so that the buffer tracks one chunk, not the prefix
          if (spans_.empty() && scan > 0) {
            memmove(buf_, buf_ + scan, end_ - scan);
            end_ -= scan;
            scan = 0;
          }
          fill();
          continue;
        }
      }
      scan = next;
      if (e == b && eof_ && scan == end_) break; // empty tail
      const size_t t = line_no_++;
      if (keep && t < keep->size() && (*keep)[t] == 0) continue;
      spans_.push_back(std::make_pair(b, e));
    }
    beg_ = scan;
    const size_t l = spans_.size();
    if (l == 0) return 0;
    if (rows) rows->assign(l, Row());
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(n_threads_, (l + 7) / 8));
    std::vector<int> bad(nt, 0);
    auto work = [&](unsigned w, size_t r0, size_t r1) {
      for (size_t r = r0; r < r1; ++r)
        if (!parse_line(buf_ + spans_[r].first, buf_ + spans_[r].second, X + r * ld,
                        rows ? &(*rows)[r] : nullptr, cols))
          bad[w] = 1;
    };
    if (nt == 1) {
      work(0, 0, l);
    } else {
      std::vector<std::thread> pool;
      for (unsigned w = 1; w < nt; ++w) pool.emplace_back(work, w, l * w / nt, l * (w + 1) / nt);
      work(0, 0, l / nt);
      for (std::thread &th : pool) th.join();
    }
    for (int b : bad)
      if (b) {
        if (skip_ == 3) std::cout << "Problem reading geno file (not enough genotypes in line)" << std::endl;
        return (size_t)-1;
      }
    return l;
  }

private:
  static constexpr size_t TEXT_CAP = size_t(1) << 30, CHUNK = size_t(16) << 20;
  // bytes of text one block may hold before read_block returns what it has (GEMMA_HIP_IO_TEXT_CAP: test hook)
  static size_t text_cap() {
    const char *env = getenv("GEMMA_HIP_IO_TEXT_CAP");
    return env && atol(env) > 0 ? (size_t)atol(env) : TEXT_CAP;
  }
  // appends up to CHUNK further bytes of text
  void fill() {
    const size_t chunk = std::min(CHUNK, text_cap()); // (a small test cap also makes the reads small)
    if (cap_ < end_ + chunk) { // plain realloc: no zero fill of text that is about to be overwritten
      cap_ = std::max(cap_ + cap_ / 2, end_ + chunk);
      char *nb = static_cast<char *>(realloc(buf_, cap_));
      if (!nb) throw std::bad_alloc();
      buf_ = nb;
    }
    const int r = gzread(f_, buf_ + end_, (unsigned)chunk);
    if (r <= 0) eof_ = true; else end_ += (size_t)r;
  }
  // next complete line [b, e) at or after `from`; "\n", "\r\n" and "\r" end a line (src/gemma_io.cpp:118-151)
  bool next_line(size_t from, size_t &b, size_t &e, size_t &next) const {
    if (from >= end_) return false;
    const char *base = buf_;
    const char *nl = (const char *)memchr(base + from, '\n', end_ - from);
    const size_t lim = nl ? (size_t)(nl - base) : end_;
    const char *cr = (const char *)memchr(base + from, '\r', lim - from);
    if (cr) {
      const size_t q = (size_t)(cr - base);
      if (q + 1 == end_ && !eof_) return false; // cannot tell "\r" from "\r\n" yet
      b = from; e = q;
      next = (q + 1 < end_ && base[q + 1] == '\n') ? q + 2 : q + 1;
      return true;
    }
    if (!nl) return false;
    b = from; e = lim; next = lim + 1;
    return true;
  }
  bool parse_line(const char *p, const char *end, double *x, Row *row, const int *cols) const {
    const char *b, *e;
    for (int k = 0; k < skip_; ++k) {
      if (!detail::next_token(p, end, b, e)) return false;
      if (row) (k == 0 ? row->rs : k == 1 ? row->minor : row->major).assign(b, e);
    }
    size_t o = 0;
    for (size_t i = 0; i < ni_total_; ++i) {
      if (!detail::next_token(p, end, b, e)) return false;
      if (cols && cols[i] == 0) continue;
      // only the genotype reader knows "NA"; the kinship and gene-expression readers call atof on every token
      x[o++] = (skip_ == 3 && detail::tok_is_na(b, e)) ? std::numeric_limits<double>::quiet_NaN() : parse_double(b, e);
    }
    return !(strict_ && detail::next_token(p, end, b, e));
  }
  gzFile f_;
  size_t ni_total_;
  unsigned n_threads_;
  int skip_;
  bool strict_;
  size_t line_no_, beg_, end_;
  bool eof_;
  char *buf_;
  size_t cap_;
  std::vector<std::pair<size_t, size_t>> spans_;
};

// rows per block of the text feeders: at most `cap` rows and about 256 MiB of doubles (two such slots are in flight)
inline size_t bimbam_block_rows(size_t row_len, size_t cap) {
  return io_block_rows(std::max<size_t>(64, std::min<size_t>(cap, (size_t(256) << 20) / (8 * std::max<size_t>(row_len, 1)))));
}

// ---------------------------------------------------------------------------------------------------------------
// first pass: which SNPs are analysed
// ---------------------------------------------------------------------------------------------------------------
struct QcLevels {
  double maf_level = 0.01, miss_level = 0.05, hwe_level = 0, r2_level = 0.9999; // src/param.cpp:94-107
};

// ReadFile_bed, src/gemma_io.cpp:876-1064.  W: ni_test x n_cvt covariates of the analysed individuals.
inline bool ReadFile_bed(const std::string &file_bed, const std::set<std::string> &setSnps, const Matrix *W,
                         std::vector<int> &indicator_idv, std::vector<int> &indicator_snp,
                         std::vector<SNPINFO> &snpInfo, const double &maf_level, const double &miss_level,
                         const double &hwe_level, const double &r2_level, size_t &ns_test) {
  indicator_snp.clear();
  const size_t ns_total = snpInfo.size();
  std::ifstream infile(file_bed.c_str(), std::ios::binary);
  if (!infile) {
    std::cout << "error reading bed file:" << file_bed << std::endl;
    return false;
  }
  if (W->tda != W->size2) return false;
  const size_t ni_total = indicator_idv.size(), n_bit = (ni_total + 3) / 4;
  size_t ni_test = 0;
  for (int v : indicator_idv) ni_test += v;
  ns_test = 0;
  const gemma_qc_cfg cfg = {maf_level, miss_level, hwe_level, r2_level};
  const size_t B = io_block_rows(std::min<size_t>(std::max<size_t>(ns_total, 1), K_BATCH_SIZE));
  std::vector<int> ind(B);
  std::vector<double> maf(B);
  std::vector<size_t> n_miss(B);
  const std::vector<int> all(ns_total, 1); // the first pass looks at every SNP of the file
  size_t t_next = 0;
  BlockPrefetch pf(B * n_bit, [&](void *slot, int) {
    return read_bed_rows(infile, all, t_next, n_bit, static_cast<unsigned char *>(slot), B);
  });
  for (size_t t0 = 0; t0 < ns_total;) {
    void *slot = nullptr;
    const size_t l = pf.next(slot);
    if (l == 0 || l == (size_t)-1) {
      std::cout << "error reading bed file:" << file_bed << " (truncated)" << std::endl;
      return false;
    }
    enforce_hip(gemma_hip_snp_qc(GEMMA_GENO_PLINK_2BIT, slot, l, n_bit, indicator_idv.data(), ni_total, W->data, W->size1,
                                 W->size2, &cfg, ind.data(), maf.data(), n_miss.data()),
                "ReadFile_bed");
    for (size_t i = 0; i < l; ++i) {
      SNPINFO &s = snpInfo[t0 + i];
      s.file_position = t0 + i;
      if (!setSnps.empty() && setSnps.count(s.rs_number) == 0) {
        s.n_miss = (size_t)-9;
        s.missingness = -9;
        s.maf = -9;
        indicator_snp.push_back(0);
        continue;
      }
      s.n_miss = n_miss[i];
      s.missingness = (double)n_miss[i] / (double)ni_test;
      s.maf = maf[i];
      s.n_idv = ni_test - n_miss[i];
      s.n_nb = 0;
      indicator_snp.push_back(ind[i]);
      ns_test += ind[i] != 0;
    }
    t0 += l;
  }
  return true;
}

// ReadFile_geno (first pass), src/gemma_io.cpp:639-873
inline bool ReadFile_geno(const std::string &file_geno, const std::set<std::string> &setSnps, const Matrix *W,
                          std::vector<int> &indicator_idv, std::vector<int> &indicator_snp, const double &maf_level,
                          const double &miss_level, const double &hwe_level, const double &r2_level,
                          std::map<std::string, std::string> &mapRS2chr, std::map<std::string, long int> &mapRS2bp,
                          std::map<std::string, double> &mapRS2cM, std::vector<SNPINFO> &snpInfo, size_t &ns_test) {
  indicator_snp.clear();
  snpInfo.clear();
  const size_t ni_total = indicator_idv.size();
  BimbamReader rd(file_geno, ni_total);
  if (!rd.ok()) {
    std::cout << "error reading genotype file:" << file_geno << std::endl;
    return false;
  }
  if (W->tda != W->size2) return false;
  size_t ni_test = 0;
  for (int v : indicator_idv) ni_test += v;
  ns_test = 0;
  const gemma_qc_cfg cfg = {maf_level, miss_level, hwe_level, r2_level};
  const size_t B = bimbam_block_rows(ni_total, 8192);
  std::vector<BimbamReader::Row> names[2];
  std::vector<int> ind(B);
  std::vector<double> maf(B);
  std::vector<size_t> n_miss(B);
  size_t file_pos = 0;
  // block k+1 is read and parsed while the device filters block k
  BlockPrefetch pf(B * ni_total * sizeof(double), [&](void *slot, int k) {
    return rd.read_block(B, static_cast<double *>(slot), ni_total, &names[k]);
  });
  for (;;) {
    void *slot = nullptr;
    int k = 0;
    const size_t l = pf.next(slot, &k);
    if (l == (size_t)-1) return false;
    if (l == 0) break;
    const std::vector<BimbamReader::Row> &rows = names[k];
    enforce_hip(gemma_hip_snp_qc(GEMMA_GENO_F64_SNP_MAJOR, slot, l, ni_total, indicator_idv.data(), ni_total, W->data,
                                 W->size1, W->size2, &cfg, ind.data(), maf.data(), n_miss.data()),
                "ReadFile_geno");
    for (size_t i = 0; i < l; ++i, ++file_pos) {
      const BimbamReader::Row &r = rows[i];
      SNPINFO s;
      s.rs_number = r.rs;
      s.a_minor = r.minor;
      s.a_major = r.major;
      s.n_nb = 0;
      s.file_position = file_pos;
      if (!setSnps.empty() && setSnps.count(r.rs) == 0) { // src/gemma_io.cpp:722-731
        s.chr = "-9"; s.cM = -9; s.base_position = -9; s.n_miss = 0; s.missingness = -9; s.maf = -9; s.n_idv = 0;
        snpInfo.push_back(s);
        indicator_snp.push_back(0);
        continue;
      }
      std::map<std::string, long int>::const_iterator it = mapRS2bp.find(r.rs);
      if (it == mapRS2bp.end()) {
        s.chr = "-9"; s.base_position = -9; s.cM = -9;
      } else {
        s.base_position = it->second; s.chr = mapRS2chr[r.rs]; s.cM = mapRS2cM[r.rs];
      }
      s.n_miss = n_miss[i];
      s.missingness = (double)n_miss[i] / (double)ni_test;
      s.maf = maf[i];
      s.n_idv = ni_test - n_miss[i];
      snpInfo.push_back(s);
      indicator_snp.push_back(ind[i]);
      ns_test += ind[i] != 0;
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// feeders over the threaded reader
// ---------------------------------------------------------------------------------------------------------------
// LOCO_set_Snps, src/param.cpp:52-66: with `-loco C` the kinship uses the annotated SNPs NOT on chromosome C and the
// association tests the ones on it (SNPs without annotation are in neither set)
inline void LOCO_set_Snps(std::set<std::string> &ksnps, std::set<std::string> &gwasnps,
                          const std::map<std::string, std::string> &mapchr, const std::string &loco) {
  for (std::map<std::string, std::string>::const_iterator kv = mapchr.begin(); kv != mapchr.end(); ++kv)
    (kv->second != loco ? ksnps : gwasnps).insert(kv->first);
}

// BimbamKin, src/gemma_io.cpp:1418-1597 (kinship over ALL ni_total individuals, analysed SNPs only; with a non-empty
// ksnps only its members, :1478-1480 -- snpInfo supplies the rs of every file line from the first pass)
inline bool BimbamKinThreaded(const std::string &file_geno, const std::vector<int> &indicator_snp_in, const int k_mode,
                              Matrix *matrix_kin, const std::set<std::string> &ksnps = std::set<std::string>(),
                              const std::vector<SNPINFO> *snpInfo = nullptr, const KinKeep &kk = KinKeep()) {
  std::vector<int> indicator_snp(indicator_snp_in);
  if (!ksnps.empty() && snpInfo)
    for (size_t t = 0; t < indicator_snp.size() && t < snpInfo->size(); ++t)
      if (indicator_snp[t] && ksnps.count((*snpInfo)[t].rs_number) == 0) indicator_snp[t] = 0;
  if (kk.keep) shard_keep(indicator_snp, kk.rank, kk.world);
  const size_t ni_total = matrix_kin->size1;
  BimbamReader rd(file_geno, ni_total);
  if (!rd.ok()) {
    std::cout << "error reading genotype file:" << file_geno << std::endl;
    return false;
  }
  if (matrix_kin->tda != matrix_kin->size2) return false;
  enforce_hip(gemma_hip_kin_begin(ni_total, k_mode), "BimbamKin");
  const size_t B = bimbam_block_rows(ni_total, 8192);
  BlockPrefetch pf(B * ni_total * sizeof(double), [&](void *slot, int) -> size_t {
    if (rd.lines_read() >= indicator_snp.size()) return 0;
    return rd.read_block(B, static_cast<double *>(slot), ni_total, nullptr, &indicator_snp);
  });
  for (;;) {
    void *slot = nullptr;
    const size_t l = pf.next(slot);
    if (l == (size_t)-1) return false;
    if (l == 0) break;
    enforce_hip(gemma_hip_kin_add(GEMMA_GENO_F64_SNP_MAJOR, slot, l, ni_total), "BimbamKin");
  }
  kin_finish(matrix_kin, kk, "BimbamKin");
  return true;
}

// LMM::AnalyzeBimbam, src/lmm.cpp:1660-1706: rows of the analysed SNPs over the analysed individuals, NaN = missing,
// in blocks of LMM_BATCH_SIZE rows at most (about 256 MiB of doubles per block), parsed one block ahead of the device
inline void AnalyzeBimbam(LMM &lmm, const Matrix *U, const Vector *eval, const Matrix *UtW, const Vector *Uty) {
  const size_t ni_total = lmm.indicator_idv.size(), n = U->size1;
  BimbamReader rd(lmm.file_geno, ni_total);
  if (!rd.ok()) throw std::runtime_error("error reading genotype file");
  const size_t B = bimbam_block_rows(n, LMM_BATCH_SIZE);
  const std::vector<int> keep = lmm.analysed_snps();
  BlockPrefetch pf(B * n * sizeof(double), [&](void *slot, int) -> size_t {
    if (rd.lines_read() >= keep.size()) return 0;
    return rd.read_block(B, static_cast<double *>(slot), n, nullptr, &keep, lmm.indicator_idv.data());
  });
  LMM::RowFeeder feed = [&](const double *&X) -> size_t {
    void *slot = nullptr;
    const size_t l = pf.next(slot);
    if (l == (size_t)-1) throw std::runtime_error("Problem reading geno file (not enough genotypes in line)");
    X = static_cast<const double *>(slot);
    return l;
  };
  lmm.AnalyzeFeed(U, eval, UtW, Uty, feed, B, n);
}

// MVLMM::AnalyzeBimbam, src/mvlmm.cpp:2972-3416, over the same threaded reader (mv.file_geno, mv.indicator_idv / _snp)
inline void AnalyzeBimbam(MVLMM &mv, const Matrix *U, const Vector *eval, const Matrix *UtW, const Matrix *UtY) {
  const size_t ni_total = mv.indicator_idv.size(), n = U->size1;
  BimbamReader rd(mv.file_geno, ni_total);
  if (!rd.ok()) throw std::runtime_error("error reading genotype file");
  const size_t B = bimbam_block_rows(n, LMM_BATCH_SIZE);
  const std::vector<int> keep = mv.analysed_snps();
  BlockPrefetch pf(B * n * sizeof(double), [&](void *slot, int) -> size_t {
    if (rd.lines_read() >= keep.size()) return 0;
    return rd.read_block(B, static_cast<double *>(slot), n, nullptr, &keep, mv.indicator_idv.data());
  });
  LMM::RowFeeder feed = [&](const double *&X) -> size_t {
    void *slot = nullptr;
    const size_t l = pf.next(slot);
    if (l == (size_t)-1) throw std::runtime_error("Problem reading geno file (not enough genotypes in line)");
    X = static_cast<const double *>(slot);
    return l;
  };
  mv.AnalyzeFeed(U, eval, UtW, UtY, feed, B, n);
}

// ReadFile_gene, src/gemma_io.cpp:2307-2364 (`-gene`): a header line, then one row per gene: id and ni_total expression
// values.  Only the ids (into snpInfo.rs_number) and the row count matter to the LMM path.
inline bool ReadFile_gene(const std::string &file_gene, std::vector<SNPINFO> &snpInfo, size_t &ng_total) {
  ng_total = 0;
  TextFile infile(file_gene);
  if (!infile.ok()) {
    std::cout << "error! fail to open gene expression file: " << file_gene << std::endl;
    return false;
  }
  std::string line;
  infile.getline(line); // header
  size_t n_idv = 0;
  while (infile.getline(line)) {
    const char *p = line.data(), *end = p + line.size(), *b, *e;
    if (!detail::next_token(p, end, b, e)) {
      std::cout << "Parsing input file '" << file_gene << "' failed in function ReadFile_gene" << std::endl;
      return false;
    }
    SNPINFO s;
    s.chr = "-9"; s.rs_number.assign(b, e); s.cM = -9; s.base_position = -9; s.a_minor = "-9"; s.a_major = "-9";
    s.n_miss = 0; s.missingness = -9; s.maf = -9; s.n_idv = 0; s.n_nb = 0; s.file_position = 0;
    size_t t = 0;
    while (detail::next_token(p, end, b, e)) ++t;
    if (ng_total == 0) n_idv = t;
    if (t != n_idv) {
      std::cout << "error! number of columns doesn't match in row: " << ng_total << std::endl;
      return false;
    }
    snpInfo.push_back(s);
    ng_total++;
  }
  return true;
}

// LMM::AnalyzeGene, src/lmm.cpp:1365-1471, from the file: rows = genes over the analysed individuals (atof on every
// token), the tested variable is the -p phenotype whose rotation arrives in Utx; ng_total rows after the header
inline void AnalyzeGene(LMM &lmm, const Matrix *U, const Vector *eval, const Matrix *UtW, const Vector *Utx, size_t ng_total) {
  const size_t ni_total = lmm.indicator_idv.size(), n = U->size1;
  BimbamReader rd(lmm.file_gene, ni_total, 0, 1);
  if (!rd.ok() || !rd.skip_line()) throw std::runtime_error("error reading gene expression file");
  const size_t B = bimbam_block_rows(n, LMM_BATCH_SIZE);
  BlockPrefetch pf(B * n * sizeof(double), [&](void *slot, int) -> size_t {
    if (rd.lines_read() >= ng_total) return 0;
    return rd.read_block(std::min(B, ng_total - rd.lines_read()), static_cast<double *>(slot), n, nullptr, nullptr,
                         lmm.indicator_idv.data());
  });
  LMM::RowFeeder feed = [&](const double *&Y) -> size_t {
    void *slot = nullptr;
    const size_t l = pf.next(slot);
    if (l == (size_t)-1) throw std::runtime_error("Parsing the gene expression file failed (not enough columns)");
    Y = static_cast<const double *>(slot);
    return l;
  };
  lmm.AnalyzeGeneFeed(U, eval, UtW, Utx, feed, B, n);
}

// LM::AnalyzeBimbam, src/lm.cpp:382-503, over the same threaded reader (lm.file_geno, lm.indicator_idv / _snp)
inline void AnalyzeBimbam(LM &lm, const Matrix *W, const Vector *y) {
  const size_t ni_total = lm.indicator_idv.size(), n = W->size1;
  BimbamReader rd(lm.file_geno, ni_total);
  if (!rd.ok()) throw std::runtime_error("error reading genotype file");
  const size_t B = bimbam_block_rows(n, LMM_BATCH_SIZE);
  const std::vector<int> keep = lm.analysed_snps();
  BlockPrefetch pf(B * n * sizeof(double), [&](void *slot, int) -> size_t {
    if (rd.lines_read() >= keep.size()) return 0;
    return rd.read_block(B, static_cast<double *>(slot), n, nullptr, &keep, lm.indicator_idv.data());
  });
  LMM::RowFeeder feed = [&](const double *&X) -> size_t {
    void *slot = nullptr;
    const size_t l = pf.next(slot);
    if (l == (size_t)-1) throw std::runtime_error("Problem reading geno file (not enough genotypes in line)");
    X = static_cast<const double *>(slot);
    return l;
  };
  lm.AnalyzeFeed(W, y, feed, B, n);
}

// ReadFile_kin (k_mode == 1), src/gemma_io.cpp:1186-1243, on the thread pool: the same dense ni_total x ni_total text,
// rows and columns of non-analysed individuals dropped while parsing, straight into G (the serial twin is
// gemma_host.hpp's ReadFile_kin; at n = 20 000 the file holds 4e8 numbers)
inline void ReadFile_kin_threaded(const std::string &file_kin, std::vector<int> &indicator_idv, bool &error, Matrix *G) {
  const size_t ni_total = indicator_idv.size();
  BimbamReader rd(file_kin, ni_total, 0, 0, true);
  if (!rd.ok()) {
    std::cout << "error! fail to open kinship file: " << file_kin << std::endl;
    error = true;
    return;
  }
  size_t i_test = 0;
  while (i_test < G->size1) {
    const size_t l = rd.read_block(G->size1 - i_test, G->data + i_test * G->tda, G->tda, nullptr, &indicator_idv,
                                   indicator_idv.data());
    if (l == (size_t)-1) {
      std::cout << "number of columns in the kinship file does not match the number of individuals" << std::endl;
      error = true;
      return;
    }
    if (l == 0) break;
    i_test += l;
  }
  // whatever is left must be rows of non-analysed individuals, and the file must end with row ni_total
  std::vector<double> rest(G->size2 ? G->size2 : 1);
  while (rd.lines_read() < ni_total) {
    const size_t before = rd.lines_read();
    const size_t l = rd.read_block(1, rest.data(), rest.size(), nullptr, &indicator_idv, indicator_idv.data());
    if (l == (size_t)-1 || (l == 0 && rd.lines_read() == before)) break;
  }
  std::vector<double> extra(ni_total ? ni_total : 1);
  if (i_test != G->size1 || rd.lines_read() != ni_total || rd.read_block(1, extra.data(), extra.size()) != 0) {
    std::cout << "number of rows in the kinship file does not match the number of individuals." << std::endl;
    error = true;
  }
}

// ReadFile_kin with k_mode == 2 (`-km 2`), src/gemma_io.cpp:1244-1288: "id1 id2 value" triples over the .fam ids
// (mapID2num from ReadFile_fam); pairs with an unknown or non-analysed id are skipped, the matrix is filled symmetrically,
// a pair given twice with different values is an error
inline void ReadFile_kin_km2(const std::string &file_kin, const std::vector<int> &indicator_idv,
                             const std::map<std::string, int> &mapID2num, bool &error, Matrix *G) {
  TextFile infile(file_kin);
  if (!infile.ok()) {
    std::cout << "error! fail to open kinship file: " << file_kin << std::endl;
    error = true;
    return;
  }
  for (size_t i = 0; i < G->size1; ++i)
    for (size_t j = 0; j < G->size2; ++j) G->data[i * G->tda + j] = 0.0;
  std::vector<long> id2id(indicator_idv.size(), -1);
  long c = 0;
  for (size_t i = 0; i < indicator_idv.size(); ++i)
    if (indicator_idv[i] == 1) id2id[i] = c++;
  std::string line;
  while (infile.getline(line)) {
    const char *p = line.data(), *end = p + line.size(), *b[3], *e[3];
    for (int k = 0; k < 3; ++k)
      if (!detail::next_token(p, end, b[k], e[k])) {
        std::cout << "Parsing input file '" << file_kin << "' failed in function ReadFile_kin" << std::endl;
        error = true;
        return;
      }
    const std::map<std::string, int>::const_iterator i1 = mapID2num.find(std::string(b[0], e[0])),
                                                     i2 = mapID2num.find(std::string(b[1], e[1]));
    if (i1 == mapID2num.end() || i2 == mapID2num.end()) continue;
    const long r = id2id[(size_t)i1->second], q = id2id[(size_t)i2->second];
    if (r < 0 || q < 0) continue;
    const double d = parse_double(b[2], e[2]);
    const double have = G->data[(size_t)r * G->tda + (size_t)q];
    if (have != 0 && have != d) {
      std::cout << "error! redundant and unequal terms in the kinship file, for id1 = " << std::string(b[0], e[0])
                << " and id2 = " << std::string(b[1], e[1]) << std::endl;
      error = true;
      return;
    }
    G->data[(size_t)r * G->tda + (size_t)q] = d;
    G->data[(size_t)q * G->tda + (size_t)r] = d;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// -eigen artefacts and their readers
// ---------------------------------------------------------------------------------------------------------------
// `gemma -k K -eigen` (src/gemma.cpp:1779-1800): <out>.eigenU.txt and <out>.eigenD.txt through PARAM::WriteMatrix /
// WriteVector (src/param.cpp:1886-1935: precision(10), tab separated)
inline bool WriteEigen(const Matrix *U, const Vector *eval, const std::string &path_out, const std::string &file_out) {
  return WriteMatrix(U, path_out + "/" + file_out + ".eigenU.txt") &&
         WriteVector(eval, path_out + "/" + file_out + ".eigenD.txt");
}

// ReadFile_eigenU, src/gemma_io.cpp:1323-1369: n_row lines of n_col numbers; U must be n_row x n_col
inline void ReadFile_eigenU(const std::string &file_ku, bool &error, Matrix *U) {
  TextFile infile(file_ku);
  if (!infile.ok()) {
    std::cout << "error! fail to open the U file: " << file_ku << std::endl;
    error = true;
    return;
  }
  const size_t n_row = U->size1, n_col = U->size2;
  for (size_t i = 0; i < n_row; ++i)
    for (size_t j = 0; j < n_col; ++j) U->data[i * U->tda + j] = 0.0;
  std::string line;
  size_t i_row = 0;
  while (infile.getline(line)) {
    if (i_row == n_row) {
      std::cout << "error! number of rows in the U file is larger than expected." << std::endl;
      error = true;
      return;
    }
    size_t i_col = 0;
    const char *p = line.data(), *end = p + line.size(), *b, *e;
    while (detail::next_token(p, end, b, e)) {
      if (i_col == n_col) {
        std::cout << "error! number of columns in the U file is larger than expected, for row = " << i_row << std::endl;
        error = true;
        return;
      }
      U->data[i_row * U->tda + i_col++] = parse_double(b, e);
    }
    i_row++;
  }
}

// ReadFile_eigenU on the thread pool (at n = 20 000 the file holds 4e8 numbers): the same doubles for a well-formed
// file; stricter on a malformed one -- the reference zero-fills rows / columns a short file lacks and only rejects a surplus
// (src/gemma_io.cpp:1338-1362), here both are errors
inline void ReadFile_eigenU_threaded(const std::string &file_ku, bool &error, Matrix *U) {
  BimbamReader rd(file_ku, U->size2, 0, 0, true);
  if (!rd.ok()) {
    std::cout << "error! fail to open the U file: " << file_ku << std::endl;
    error = true;
    return;
  }
  size_t i_row = 0;
  while (i_row < U->size1) {
    const size_t l = rd.read_block(U->size1 - i_row, U->data + i_row * U->tda, U->tda);
    if (l == (size_t)-1) {
      std::cout << "error! number of columns in the U file does not match, near row = " << i_row << std::endl;
      error = true;
      return;
    }
    if (l == 0) break;
    i_row += l;
  }
  std::vector<double> extra(U->size2 ? U->size2 : 1);
  if (i_row != U->size1 || rd.read_block(1, extra.data(), extra.size()) != 0) {
    std::cout << "error! number of rows in the U file does not match." << std::endl;
    error = true;
  }
}

// ReadFile_eigenD, src/gemma_io.cpp:1372-1416: one number per line
inline void ReadFile_eigenD(const std::string &file_kd, bool &error, Vector *eval) {
  TextFile infile(file_kd);
  if (!infile.ok()) {
    std::cout << "error! fail to open the D file: " << file_kd << std::endl;
    error = true;
    return;
  }
  const size_t n_row = eval->size;
  for (size_t i = 0; i < n_row; ++i) eval->data[i * eval->stride] = 0.0;
  std::string line;
  size_t i_row = 0;
  while (infile.getline(line)) {
    if (i_row == n_row) {
      std::cout << "error! number of rows in the D file is larger than expected." << std::endl;
      error = true;
      return;
    }
    const char *p = line.data(), *end = p + line.size(), *b, *e;
    if (!detail::next_token(p, end, b, e)) {
      std::cout << "Parsing input file '" << file_kd << "' failed in function ReadFile_eigenD" << std::endl;
      error = true;
      return;
    }
    eval->data[i_row * eval->stride] = parse_double(b, e);
    if (detail::next_token(p, end, b, e)) {
      std::cout << "error! number of columns in the D file is larger than expected, for row = " << i_row << std::endl;
      error = true;
      return;
    }
    i_row++;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// <o>.log.txt -- the lines of GEMMA::WriteLog (src/gemma.cpp:3159-3590) that belong to this path, in its format
// ("## key = value", ostream default precision): the summary counts, the null-model estimates of the univariate LMM and
// the five timers (minutes), plus one line naming the device.  The null model's beta / se(beta) lines are not written
// (gemma_hip_lmm_null returns the variance components only).
// ---------------------------------------------------------------------------------------------------------------
struct RunLog {
  std::string command_line;
  int a_mode = 0; // 1-4, 9: -lmm; 21 / 22: -gk; 51-54: -lm (src/gemma.h:39-43)
  size_t ni_total = 0, ni_test = 0, n_cvt = 1, n_ph = 1, ns_total = 0, ns_test = 0;
  bool have_null = false;
  NullModel null;
  double time_total = 0, time_G = 0, time_eigen = 0, time_UtX = 0, time_opt = 0; // minutes

  bool Write(const std::string &path_out, const std::string &file_out) const {
    const std::string file_str = path_out + "/" + file_out + ".log.txt";
    std::ofstream outfile(file_str.c_str(), std::ofstream::out);
    if (!outfile) {
      std::cout << "error writing log file: " << file_str << std::endl;
      return false;
    }
    char name[256] = "no device";
    int n_cu = 0;
    size_t hbm = 0;
    gemma_hip_device_info(name, sizeof name, &n_cu, &hbm);
    outfile << "##" << std::endl;
    outfile << "## GEMMA path on HIP, C ABI version = " << gemma_hip_abi_version() << std::endl;
    outfile << "## device = " << name << ", " << n_cu << " CUs, " << (double)hbm / 1e9 << " GB" << std::endl;
    outfile << "##" << std::endl;
    outfile << "## Command Line Input = " << command_line << std::endl;
    outfile << "##" << std::endl;
    outfile << "## Summary Statistics:" << std::endl;
    outfile << "## number of total individuals = " << ni_total << std::endl;
    outfile << "## number of analyzed individuals = " << ni_test << std::endl;
    outfile << "## number of covariates = " << n_cvt << std::endl;
    outfile << "## number of phenotypes = " << n_ph << std::endl;
    outfile << "## number of total SNPs/var = " << ns_total << std::endl;
    outfile << "## number of analyzed SNPs/var = " << ns_test << std::endl;
    if (have_null) {
      outfile << "## REMLE log-likelihood in the null model = " << null.logl_remle_H0 << std::endl;
      outfile << "## MLE log-likelihood in the null model = " << null.logl_mle_H0 << std::endl;
      if (n_ph == 1) {
        outfile << "## pve estimate in the null model = " << null.pve_null << std::endl;
        outfile << "## se(pve) in the null model = " << null.pve_se_null << std::endl;
        outfile << "## vg estimate in the null model = " << null.vg_remle_null << std::endl;
        outfile << "## ve estimate in the null model = " << null.ve_remle_null << std::endl;
      }
    }
    outfile << "##" << std::endl;
    outfile << "## Computation Time:" << std::endl;
    outfile << "## total computation time = " << time_total << " min " << std::endl;
    outfile << "## computation time break down: " << std::endl;
    if (a_mode == 21 || a_mode == 22)
      outfile << "##      time on calculating relatedness matrix = " << time_G << " min " << std::endl;
    if ((a_mode >= 1 && a_mode <= 4) || a_mode == 9) {
      outfile << "##      time on eigen-decomposition = " << time_eigen << " min " << std::endl;
      outfile << "##      time on calculating UtX = " << time_UtX << " min " << std::endl;
    }
    if ((a_mode >= 1 && a_mode <= 4) || a_mode == 9 || (a_mode >= 51 && a_mode <= 54))
      outfile << "##      time on optimization = " << time_opt << " min " << std::endl;
    outfile << "##" << std::endl;
    return true;
  }
};

} // namespace gemma_amd
#endif
