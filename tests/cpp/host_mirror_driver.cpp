// End-to-end driver over include/gemma_host.hpp -- the sequence of GEMMA's BatchRun for
//   gemma -bfile P -gk 1   followed by   gemma -bfile P -k K -lmm <mode>
// (src/gemma.cpp:1900-1926 and :2557-2830) with every numeric step going through the C ABI.
// Test harness only (tests/test_gpu_host_mirror.py builds and runs it); not a CLI replacement.
//
// usage: host_mirror_driver <bfile-prefix> <ni_total> <n_snps> <pheno.txt> <a_mode> <outdir> <outname>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "gemma_host.hpp"

using namespace gemma_amd;

int main(int argc, char **argv) {
  if (argc < 8) return 2;
  const std::string prefix = argv[1];
  const size_t ni_total = std::strtoul(argv[2], nullptr, 10), ns = std::strtoul(argv[3], nullptr, 10);
  const std::string pheno = argv[4];
  const int a_mode = std::atoi(argv[5]);
  try {
    enforce_hip(gemma_hip_init(0, 0), "init");
    // phenotypes: "NA" -> individual not analysed (indicator_idv, src/param.cpp:1993-2036)
    std::vector<int> indicator_idv;
    std::vector<double> yall;
    {
      std::ifstream f(pheno.c_str());
      std::string tok;
      while (f >> tok) {
        if (tok == "NA") { indicator_idv.push_back(0); yall.push_back(-9); }
        else { indicator_idv.push_back(1); yall.push_back(std::atof(tok.c_str())); }
      }
    }
    if (indicator_idv.size() != ni_total) { std::cerr << "pheno rows != ni_total\n"; return 2; }
    size_t ni_test = 0;
    for (int v : indicator_idv) ni_test += v;
    std::vector<int> indicator_snp(ns, 1);

    // -gk: K over all individuals (src/gemma.cpp:1903-1908)
    std::vector<double> Kbuf(ni_total * ni_total, 0.0);
    Matrix K = matrix_view(Kbuf.data(), ni_total, ni_total);
    if (!PlinkKin(prefix + ".bed", indicator_snp, 1, 0, &K)) return 3;

    // the two-run hand-off: cXX.txt at 10 significant digits (src/gemma.cpp:1919), read back with the
    // non-analysed individuals dropped (ReadFile_kin, src/gemma_io.cpp:1205-1243); then centre, eigen
    const std::string cxx = std::string(argv[6]) + "/" + argv[7] + ".cXX.txt";
    if (!WriteMatrix(&K, cxx)) return 3;
    std::vector<double> Gbuf(ni_test * ni_test), Ubuf(ni_test * ni_test), evalbuf(ni_test);
    Matrix G = matrix_view(Gbuf.data(), ni_test, ni_test), U = matrix_view(Ubuf.data(), ni_test, ni_test);
    bool error = false;
    ReadFile_kin(cxx, indicator_idv, error, &G);
    if (error) return 3;
    Vector eval = vector_view(evalbuf.data(), ni_test);
    CenterMatrix(&G);
    const double trace_G = EigenDecomp_Zeroed(&G, &U, &eval, 0);

    std::vector<double> Wbuf(ni_test, 1.0), ybuf, UtWbuf(ni_test), Utybuf(ni_test);
    for (size_t i = 0; i < ni_total; ++i) if (indicator_idv[i]) ybuf.push_back(yall[i]);
    Matrix W = matrix_view(Wbuf.data(), ni_test, 1), Y = matrix_view(ybuf.data(), ni_test, 1);
    Matrix UtW = matrix_view(UtWbuf.data(), ni_test, 1), UtY = matrix_view(Utybuf.data(), ni_test, 1);
    CalcUtX(&U, &W, &UtW);
    CalcUtX(&U, &Y, &UtY);
    Vector Uty = vector_view(Utybuf.data(), ni_test);
    const NullModel nm = CalcLambdaNull(&eval, &UtW, &Uty, 1e-5, 1e5, 10, trace_G);

    LMM cLmm;
    cLmm.a_mode = a_mode;
    cLmm.file_bfile = prefix;
    cLmm.path_out = argv[6];
    cLmm.file_out = argv[7];
    cLmm.ni_total = ni_total;
    cLmm.indicator_idv = indicator_idv;
    cLmm.indicator_snp = indicator_snp;
    cLmm.l_mle_null = nm.l_mle_null;
    cLmm.logl_mle_H0 = nm.logl_mle_H0;
    for (size_t t = 0; t < ns; ++t) {
      SNPINFO s;
      s.chr = "1"; s.rs_number = "rs" + std::to_string(t); s.cM = 0; s.base_position = (long)t + 1;
      s.a_minor = "A"; s.a_major = "G"; s.n_miss = 0; s.missingness = 0; s.maf = 0.25; s.n_idv = ni_test;
      s.n_nb = 0; s.file_position = t;
      cLmm.snpInfo.push_back(s);
    }
    cLmm.AnalyzePlink(&U, &eval, &UtW, &Uty);
    cLmm.WriteFiles();
    std::cout << "trace_G " << std::setprecision(12) << trace_G << " l_remle_null " << nm.l_remle_null << " pve "
              << nm.pve_null << " snps " << cLmm.sumStat.size() << " time_UtX(min) " << cLmm.time_UtX << std::endl;
    gemma_hip_shutdown();
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
