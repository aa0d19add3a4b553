// CPU harness for include/gemma_io_host.hpp (tests/test_io_host.py builds and drives it): every sub-command runs one
// reader / writer of the host layer and prints what it produced in a form the Python side can compare exactly
// (doubles as %.17g or raw bytes).  No device call is made here.
//
//   parse                          tokens on stdin, one per line -> the 8 bytes of parse_double() as hex
//   pheno|fam <file> <col>...      indicator and value per row and column
//   cvt <file>                     n_cvt, then indicator and values per row
//   bim <file> | anno <file>
//   cvtphen <pheno|fam:file> <cvt|-> <col>...   ProcessCvtPhen + CopyCvtPhen: indicator_idv, n_cvt, W, Y
//   geno <file> <ni_total> <threads> <block> <out.bin> [keep-file] [cols-file]
//                                  BimbamReader blocks appended to out.bin (raw doubles), "rs minor major" per row
//   eigen <U-in> <D-in> <n> <outdir> <name>     ReadFile_eigenU/D -> WriteEigen
//   assoc <assoc-in> <a_mode> <outdir> <name>   parse a reference .assoc.txt, LMM::WriteFiles it again
//   assocbench <a_mode> <n_snps> <outdir>       wall time of LMM::WriteFiles on n_snps synthetic records
//   prefetch                                    BlockPrefetch scenarios: order of blocks, an exception in the producer,
//                                               a consumer that stops early (no deadlock), an immediately empty source
//   kinbench <n> <file>                         wall time of WriteMatrix + ReadFile_kin on an n x n matrix
//   kin <cXX-in> <n> <out>                      ReadFile_kin (all individuals) -> WriteMatrix
//   plinkgen <prefix> <ni> <ns> [threads [n_ph]]  synthetic PLINK set (n_ph correlated traits in .fam columns 6..): two sub-populations, maf ~ U(0.1, 0.45) +- 0.075, 1 % missing calls,
//                                               y = 0.3 * (first 20 SNPs) + 0.8 * population + N(0,1), 2 % of the phenotypes -9
//   genogen <file> <ni_total> <n_snps>          synthetic BIMBAM file ("0.123"-style dosages, hard calls, 1 % NA)
//   genobench <file> <ni_total> <threads>       wall time of BimbamReader over the whole file (threads = 0: the
//                                               reference's own way, one thread of strtok + atof, for comparison)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "gemma_io_host.hpp"

using namespace gemma_amd;

static std::vector<int> read_ints(const char *path) {
  std::vector<int> v;
  std::ifstream f(path);
  int x;
  while (f >> x) v.push_back(x);
  return v;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const std::string cmd = argv[1];
  if (cmd == "parse") {
    std::string tok;
    while (std::getline(std::cin, tok)) {
      const double d = parse_double(tok.data(), tok.data() + tok.size());
      unsigned long long u;
      memcpy(&u, &d, 8);
      printf("%016llx\n", u);
    }
    return 0;
  }
  if (cmd == "pheno" || cmd == "fam") {
    std::vector<std::vector<int>> ind;
    std::vector<std::vector<double>> ph;
    std::vector<size_t> cols;
    for (int i = 3; i < argc; ++i) cols.push_back(strtoul(argv[i], nullptr, 10));
    std::map<std::string, int> ids;
    const bool ok = cmd == "pheno" ? ReadFile_pheno(argv[2], ind, ph, cols) : ReadFile_fam(argv[2], ind, ph, ids, cols);
    if (!ok) return 1;
    for (size_t i = 0; i < ph.size(); ++i) {
      for (size_t j = 0; j < ph[i].size(); ++j) printf("%d %.17g ", ind[i][j], ph[i][j]);
      printf("\n");
    }
    if (cmd == "fam") printf("ids %zu\n", ids.size());
    return 0;
  }
  if (cmd == "cvt") {
    std::vector<int> ind;
    std::vector<std::vector<double>> cvt;
    size_t n_cvt = 0;
    if (!ReadFile_cvt(argv[2], ind, cvt, n_cvt)) return 1;
    printf("%zu\n", n_cvt);
    for (size_t i = 0; i < cvt.size(); ++i) {
      printf("%d", ind[i]);
      for (double v : cvt[i]) printf(" %.17g", v);
      printf("\n");
    }
    return 0;
  }
  if (cmd == "bim") {
    std::vector<SNPINFO> info;
    if (!ReadFile_bim(argv[2], info)) return 1;
    for (const SNPINFO &s : info)
      printf("%s %s %.17g %ld %s %s\n", s.chr.c_str(), s.rs_number.c_str(), s.cM, s.base_position, s.a_minor.c_str(),
             s.a_major.c_str());
    return 0;
  }
  if (cmd == "anno") {
    std::map<std::string, std::string> chr;
    std::map<std::string, long int> bp;
    std::map<std::string, double> cm;
    if (!ReadFile_anno(argv[2], chr, bp, cm)) return 1;
    for (const auto &kv : bp) printf("%s %ld %s %.17g\n", kv.first.c_str(), kv.second, chr[kv.first].c_str(), cm[kv.first]);
    return 0;
  }
  if (cmd == "cvtphen") {
    CvtPhen cp;
    std::vector<size_t> cols;
    for (int i = 4; i < argc; ++i) cols.push_back(strtoul(argv[i], nullptr, 10));
    const std::string src = argv[2];
    std::map<std::string, int> ids;
    const bool ok = src.compare(0, 4, "fam:") == 0 ? ReadFile_fam(src.substr(4), cp.indicator_pheno, cp.pheno, ids, cols)
                                                   : ReadFile_pheno(src, cp.indicator_pheno, cp.pheno, cols);
    if (!ok) return 1;
    if (std::string(argv[3]) != "-" && !ReadFile_cvt(argv[3], cp.indicator_cvt, cp.cvt, cp.n_cvt)) return 1;
    cp.ProcessCvtPhen();
    if (cp.error) return 1;
    std::vector<double> W, Y;
    cp.CopyCvtPhen(W, Y);
    printf("%zu %zu\n", cp.ni_test, cp.n_cvt);
    for (int v : cp.indicator_idv) printf("%d ", v);
    printf("\n");
    for (double v : W) printf("%.17g ", v);
    printf("\n");
    for (double v : Y) printf("%.17g ", v);
    printf("\n");
    return 0;
  }
  if (cmd == "geno") {
    const size_t ni_total = strtoul(argv[3], nullptr, 10), block = strtoul(argv[5], nullptr, 10);
    BimbamReader rd(argv[2], ni_total, (unsigned)atoi(argv[4]));
    if (!rd.ok()) return 1;
    std::vector<int> keep, cols;
    if (argc > 7 && std::string(argv[7]) != "-") keep = read_ints(argv[7]);
    if (argc > 8) cols = read_ints(argv[8]);
    size_t ld = ni_total;
    if (!cols.empty()) {
      ld = 0;
      for (int c : cols) ld += c != 0;
    }
    std::vector<double> X(block * ld);
    std::vector<BimbamReader::Row> rows;
    FILE *out = fopen(argv[6], "wb");
    for (;;) {
      const size_t l = rd.read_block(block, X.data(), ld, &rows, keep.empty() ? nullptr : &keep,
                                     cols.empty() ? nullptr : cols.data());
      if (l == (size_t)-1) return 1;
      if (l == 0) break;
      fwrite(X.data(), 8, l * ld, out);
      for (size_t i = 0; i < l; ++i) printf("%s %s %s\n", rows[i].rs.c_str(), rows[i].minor.c_str(), rows[i].major.c_str());
    }
    fclose(out);
    fprintf(stderr, "text_buffer_bytes %zu\n", rd.text_buffer_bytes());
    return 0;
  }
  if (cmd == "plinkgen") {
    const std::string prefix = argv[2];
    const size_t ni = strtoul(argv[3], nullptr, 10), ns = strtoul(argv[4], nullptr, 10), nb = (ni + 3) / 4;
    const unsigned nt = argc > 5 ? (unsigned)atoi(argv[5]) : 8u;
    std::vector<unsigned char> bed(ns * nb, 0);
    auto rnd = [](unsigned long long &st) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    auto gen = [&](size_t s0, size_t s1) {
      for (size_t s = s0; s < s1; ++s) {
        unsigned long long st = 0x9E3779B97F4A7C15ull * (s + 1) + 12345;
        for (int k = 0; k < 4; ++k) rnd(st);
        // two sub-populations (first / second half of the individuals) whose allele frequencies differ by up to 0.15
        const double maf = 0.1 + 0.35 * (double)(rnd(st) >> 11) / 9007199254740992.0;
        const double dlt = 0.15 * ((double)(rnd(st) >> 11) / 9007199254740992.0 - 0.5);
        const unsigned long long thr2[2] = {(unsigned long long)((maf + dlt) * 4294967296.0),
                                            (unsigned long long)((maf - dlt) * 4294967296.0)};
        unsigned char *row = &bed[s * nb];
        for (size_t i = 0; i < ni; ++i) {
          const unsigned long long thr = thr2[i >= ni / 2];
          const unsigned long long r = rnd(st);
          unsigned code;
          if ((r >> 54) < 10) code = 1; // ~1 % missing (bits 01)
          else {
            const unsigned g = ((r & 0xffffffffull) < thr) + (((r >> 20) & 0xffffffffull) < thr); // minor-allele count
            code = g == 2 ? 0u : g == 1 ? 2u : 3u; // 00 -> 2, 10 -> 1, 11 -> 0 (src/lmm.cpp:1797-1812)
          }
          row[i >> 2] |= (unsigned char)(code << (2 * (i & 3)));
        }
      }
    };
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < nt; ++w) pool.emplace_back(gen, ns * w / nt, ns * (w + 1) / nt);
    for (std::thread &t : pool) t.join();
    FILE *f = fopen((prefix + ".bed").c_str(), "wb");
    const unsigned char magic[3] = {0x6C, 0x1B, 0x01};
    fwrite(magic, 1, 3, f);
    fwrite(bed.data(), 1, bed.size(), f);
    fclose(f);
    f = fopen((prefix + ".bim").c_str(), "w");
    for (size_t s = 0; s < ns; ++s) fprintf(f, "%zu\trs%zu\t0\t%zu\tA\tG\n", 1 + s * 20 / ns, s, s + 1);
    fclose(f);
    const size_t n_ph = argc > 6 ? strtoul(argv[6], nullptr, 10) : 1;
    unsigned long long st = 424242;
    auto gauss = [&]() { // Irwin-Hall ~ N(0, 1)
      double y = 0;
      for (int k = 0; k < 12; ++k) y += (double)(rnd(st) >> 11) / 9007199254740992.0;
      return y - 6.0;
    };
    f = fopen((prefix + ".fam").c_str(), "w");
    for (size_t i = 0; i < ni; ++i) {
      double g = 0;
      for (size_t s = 0; s < 20 && s < ns; ++s) {
        const unsigned code = (bed[s * nb + (i >> 2)] >> (2 * (i & 3))) & 3u;
        g += 0.3 * (code == 0 ? 2.0 : code == 2 ? 1.0 : 0.0);
      }
      if (i >= ni / 2) g += 0.8; // population effect: the kinship's structure carries phenotypic variance
      const double shared = n_ph > 1 ? 0.6 * gauss() : 0.0;
      fprintf(f, "f%zu i%zu 0 0 1", i, i);
      const bool miss = rnd(st) % 50 == 0;
      for (size_t k = 0; k < n_ph; ++k) {
        const double y = g * (1.0 - 0.3 * (double)k) + shared + (n_ph > 1 ? 0.8 : 1.0) * gauss();
        if (miss && k == 0) fprintf(f, " -9");
        else fprintf(f, " %.6f", y);
      }
      fprintf(f, "\n");
    }
    fclose(f);
    return 0;
  }
  if (cmd == "genogen") {
    const size_t ni_total = strtoul(argv[3], nullptr, 10), ns = strtoul(argv[4], nullptr, 10);
    FILE *out = fopen(argv[2], "wb");
    std::vector<char> line;
    unsigned long long st = 88172645463325252ull;
    for (size_t s = 0; s < ns; ++s) {
      line.clear();
      char tmp[32];
      int k = snprintf(tmp, sizeof tmp, "rs%zu, A, G", s);
      line.insert(line.end(), tmp, tmp + k);
      for (size_t i = 0; i < ni_total; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17; // xorshift64
        const unsigned r = (unsigned)(st >> 33) % 3000u;
        if (r < 30) k = snprintf(tmp, sizeof tmp, ", NA");
        else if (r < 1000) k = snprintf(tmp, sizeof tmp, ", %u", r % 3);
        else k = snprintf(tmp, sizeof tmp, ", %u.%03u", (r - 1000) / 1000, (r - 1000) % 1000);
        line.insert(line.end(), tmp, tmp + k);
      }
      line.push_back('\n');
      fwrite(line.data(), 1, line.size(), out);
    }
    fclose(out);
    return 0;
  }
  if (cmd == "genobench") {
    const size_t ni_total = strtoul(argv[3], nullptr, 10);
    const unsigned threads = (unsigned)atoi(argv[4]);
    const size_t block = 2048;
    std::vector<double> X(block * ni_total);
    const auto t0 = std::chrono::steady_clock::now();
    size_t rows = 0;
    double sum = 0;
    if (threads == 0) { // src/lmm.cpp:1675-1700 / src/gemma_io.cpp:1487-1509: getline, strtok, strcmp "NA", atof
      TextFile in(argv[2]);
      std::string line;
      while (in.getline(line)) {
        char *save = nullptr;
        char *tok = strtok_r(&line[0], " ,\t", &save);
        tok = strtok_r(nullptr, " ,\t", &save);
        tok = strtok_r(nullptr, " ,\t", &save);
        double *x = X.data() + (rows % block) * ni_total;
        for (size_t i = 0; i < ni_total; ++i) {
          tok = strtok_r(nullptr, " ,\t", &save);
          if (!tok) return 1;
          x[i] = strcmp(tok, "NA") == 0 ? std::numeric_limits<double>::quiet_NaN() : atof(tok);
        }
        sum += x[ni_total / 2] == x[ni_total / 2] ? x[ni_total / 2] : 0;
        ++rows;
      }
    } else {
      BimbamReader rd(argv[2], ni_total, threads);
      if (!rd.ok()) return 1;
      for (;;) {
        const size_t l = rd.read_block(block, X.data(), ni_total);
        if (l == (size_t)-1) return 1;
        if (l == 0) break;
        for (size_t r = 0; r < l; ++r) {
          const double v = X[r * ni_total + ni_total / 2];
          sum += v == v ? v : 0;
        }
        rows += l;
      }
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("rows %zu values %zu seconds %.3f values_per_s %.3e checksum %.6f\n", rows, rows * ni_total, sec,
           (double)(rows * ni_total) / sec, sum);
    return 0;
  }
  if (cmd == "eigen") {
    const size_t n = strtoul(argv[4], nullptr, 10);
    std::vector<double> Ub(n * n), Db(n);
    Matrix U = matrix_view(Ub.data(), n, n);
    Vector D = vector_view(Db.data(), n);
    bool error = false;
    ReadFile_eigenU(argv[2], error, &U);
    ReadFile_eigenD(argv[3], error, &D);
    if (error) return 1;
    std::vector<double> Ut(n * n);
    Matrix U2 = matrix_view(Ut.data(), n, n);
    ReadFile_eigenU_threaded(argv[2], error, &U2); // the threaded reader must give the same bits
    if (error || memcmp(Ut.data(), Ub.data(), n * n * 8) != 0) return 3;
    return WriteEigen(&U, &D, argv[5], argv[6]) ? 0 : 1;
  }
  if (cmd == "prefetch") {
    // (a) 37 blocks arrive in order, each slot intact while it is held
    {
      int produced = 0;
      BlockPrefetch pf(1024, [&](void *slot, int) -> size_t {
        if (produced == 37) return 0;
        memset(slot, produced & 0xff, 1024);
        return (size_t)(++produced);
      });
      for (int k = 1; k <= 37; ++k) {
        void *slot = nullptr;
        const size_t n = pf.next(slot);
        if (n != (size_t)k) return 10;
        std::this_thread::sleep_for(std::chrono::microseconds(200 * (k % 3))); // the producer runs ahead meanwhile
        const unsigned char *b = static_cast<unsigned char *>(slot);
        for (int i = 0; i < 1024; ++i)
          if (b[i] != (unsigned char)((k - 1) & 0xff)) return 11;
      }
      void *slot = nullptr;
      if (pf.next(slot) != 0 || pf.next(slot) != 0) return 12; // the end is sticky
    }
    // (b) an exception in the producer surfaces in next()
    {
      int produced = 0;
      BlockPrefetch pf(64, [&](void *, int) -> size_t {
        if (++produced == 3) throw std::runtime_error("producer failed");
        return 1;
      });
      void *slot = nullptr;
      bool threw = false;
      try {
        for (int k = 0; k < 5; ++k) pf.next(slot);
      } catch (const std::runtime_error &e) {
        threw = std::string(e.what()) == "producer failed";
      }
      if (!threw) return 13;
    }
    // (c) the consumer walks away after one block while the producer has more: the destructor must not hang
    {
      BlockPrefetch pf(64, [&](void *, int) -> size_t { return 1; });
      void *slot = nullptr;
      if (pf.next(slot) != 1) return 14;
    }
    // (d) empty source, malformed source
    {
      BlockPrefetch pf(64, [&](void *, int) -> size_t { return 0; });
      void *slot = nullptr;
      if (pf.next(slot) != 0) return 15;
      BlockPrefetch bad(64, [&](void *, int) -> size_t { return (size_t)-1; });
      if (bad.next(slot) != (size_t)-1 || bad.next(slot) != 0) return 16;
    }
    printf("prefetch ok\n");
    return 0;
  }
  if (cmd == "kinbench") {
    const size_t n = strtoul(argv[2], nullptr, 10);
    std::vector<double> Gb(n * n), Hb(n * n);
    unsigned long long st = 99;
    for (double &v : Gb) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = ((double)(st >> 11) / 9007199254740992.0 - 0.5) * 0.3; }
    Matrix G = matrix_view(Gb.data(), n, n), H = matrix_view(Hb.data(), n, n);
    auto t0 = std::chrono::steady_clock::now();
    if (!WriteMatrix(&G, argv[3])) return 1;
    const double tw = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<int> ind(n, 1);
    bool error = false;
    t0 = std::chrono::steady_clock::now();
    ReadFile_kin(argv[3], ind, error, &H);
    const double tr1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<double> H1(Hb);
    t0 = std::chrono::steady_clock::now();
    ReadFile_kin_threaded(argv[3], ind, error, &H);
    const double tr = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (error || memcmp(H1.data(), Hb.data(), n * n * 8) != 0) return 1;
    printf("ReadFile_kin serial %.3f s; ", tr1);
    double worst = 0;
    for (size_t i = 0; i < n * n; ++i) worst = std::max(worst, std::fabs(Hb[i] - Gb[i]));
    printf("n %zu WriteMatrix %.3f s ReadFile_kin %.3f s max |read - written| %.3e\n", n, tw, tr, worst);
    return 0;
  }
  if (cmd == "kin") {
    const size_t n = strtoul(argv[3], nullptr, 10);
    std::vector<double> Gb(n * n);
    Matrix G = matrix_view(Gb.data(), n, n);
    std::vector<int> ind(n, 1);
    bool error = false;
    if (argc > 5) { // kin <in> <n_total> <out> <indicator-file>: threaded reader with individuals dropped
      ind = read_ints(argv[5]);
      size_t nt = 0;
      for (int v : ind) nt += v != 0;
      std::vector<double> Sb(nt * nt), S2(nt * nt);
      Matrix S = matrix_view(Sb.data(), nt, nt), T = matrix_view(S2.data(), nt, nt);
      ReadFile_kin_threaded(argv[2], ind, error, &S);
      if (error) return 1;
      ReadFile_kin(argv[2], ind, error, &T);
      if (error || memcmp(Sb.data(), S2.data(), nt * nt * 8) != 0) return 3;
      return WriteMatrix(&S, argv[4]) ? 0 : 1;
    }
    ReadFile_kin_threaded(argv[2], ind, error, &G);
    if (error) return 1;
    return WriteMatrix(&G, argv[4]) ? 0 : 1;
  }
  if (cmd == "assocbench") {
    LMM lmm;
    lmm.a_mode = atoi(argv[2]);
    const size_t ns = strtoul(argv[3], nullptr, 10);
    lmm.path_out = argv[4];
    lmm.file_out = "bench";
    for (size_t t = 0; t < ns; ++t) {
      SNPINFO s;
      s.chr = "1"; s.rs_number = "rs" + std::to_string(t); s.cM = 0; s.base_position = (long)t + 1; s.a_minor = "A";
      s.a_major = "G"; s.n_miss = t % 300; s.missingness = 0; s.maf = 0.05 + 0.45 * (double)(t % 1000) / 1000.0; s.n_idv = 0;
      s.n_nb = 0; s.file_position = t;
      lmm.snpInfo.push_back(s);
      lmm.indicator_snp.push_back(1);
      const double x = 1e-3 * (double)(t % 7919) + 1e-7;
      SUMSTAT st = {0.3 - x, 0.02 + x * 1e-2, 1.5 + x, 1.4 + x, x * 1e-20, x * 1e-3, x * 1e-2, -32000.0 - x};
      lmm.sumStat.push_back(st);
    }
    const auto t0 = std::chrono::steady_clock::now();
    lmm.WriteFiles();
    printf("WriteFiles %zu records: %.3f s\n", ns, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return 0;
  }
  if (cmd == "assoc") {
    LMM lmm;
    lmm.a_mode = atoi(argv[3]);
    lmm.path_out = argv[4];
    lmm.file_out = argv[5];
    std::ifstream f(argv[2]);
    std::string line;
    std::getline(f, line); // header
    while (std::getline(f, line)) {
      std::vector<std::string> t;
      size_t p = 0;
      while (p <= line.size()) {
        const size_t q = line.find('\t', p);
        t.push_back(line.substr(p, q == std::string::npos ? std::string::npos : q - p));
        if (q == std::string::npos) break;
        p = q + 1;
      }
      SNPINFO s;
      s.chr = t[0]; s.rs_number = t[1]; s.base_position = atol(t[2].c_str()); s.n_miss = strtoul(t[3].c_str(), nullptr, 10);
      s.a_minor = t[4]; s.a_major = t[5]; s.maf = atof(t[6].c_str());
      s.cM = 0; s.missingness = 0; s.n_idv = 0; s.n_nb = 0; s.file_position = 0;
      lmm.snpInfo.push_back(s);
      lmm.indicator_snp.push_back(1);
      std::vector<double> v;
      for (size_t k = 7; k < t.size(); ++k) v.push_back(atof(t[k].c_str()));
      SUMSTAT st = {0, 0, 0, 0, 0, 0, 0, 0};
      switch (lmm.a_mode) {
      case 1: st.beta = v[0]; st.se = v[1]; st.logl_H1 = v[2]; st.lambda_remle = v[3]; st.p_wald = v[4]; break;
      case 2: st.logl_H1 = v[0]; st.lambda_mle = v[1]; st.p_lrt = v[2]; break;
      case 3: st.beta = v[0]; st.se = v[1]; st.p_score = v[2]; break;
      case 4: st.beta = v[0]; st.se = v[1]; st.logl_H1 = v[2]; st.lambda_remle = v[3]; st.lambda_mle = v[4];
              st.p_wald = v[5]; st.p_lrt = v[6]; st.p_score = v[7]; break;
      case 9: st.beta = v[0]; st.se = v[1]; st.lambda_mle = v[2]; st.p_lrt = v[3]; break;
      }
      lmm.sumStat.push_back(st);
    }
    lmm.WriteFiles();
    return 0;
  }
  return 2;
}
