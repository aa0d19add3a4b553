// File-driven run of the path, the way `gemma` is invoked (test harness for include/gemma_io_host.hpp +
// include/gemma_host.hpp; NOT a replacement of GEMMA's CLI -- INTEGRATION.md binds the C ABI inside GEMMA itself):
//
//   gemma_file_driver (-g geno[.gz] -p pheno [-a anno] | -bfile prefix) [-c cvt] [-n col [col ...]]
//                     (-gk [1|2] | -k kin (-eigen | -lmm [1|2|3|4|9]) | -d eigenD -u eigenU -lmm m | -lm [1|2|3|4])
//                     [-maf x] [-miss x] [-hwe x] [-r2 x] [-loco chr] [-gxe env] [-snps list] [-notsnp] [-km 2] [-o name] [-outdir dir]
//   gemma_file_driver ... -lmm m -gpus N [-samegpu]    one process per GPU over RCCL.  Every rank reads the
//                     small files and makes the first pass; with -inproc the kinship is SNP-SHARDED (each rank accumulates its
//                     share of the SNPs, ONE ncclAllReduce of the n^2 sums); the eigendecomposition runs on rank 0 ONLY and
//                     (U, eval) reach the other ranks in ONE ncclBroadcast (with -k likewise: rank 0 alone reads the kinship
//                     file); every rank analyses its contiguous share of the SNPs; the parent concatenates the per-rank parts
//                     in rank order into <o>.assoc.txt.  The communicator id travels over a pipe opened before the fork.
//                     (-samegpu puts every rank on device 0 and selects the library's shared-memory test transport,
//                     GEMMA_HIP_COMM=shm -- RCCL refuses two ranks on one device: a test hook for 1-GPU boxes)
//   gemma_file_driver -gene expr.txt -p pheno -k kin -lmm m        every row of expr.txt is a phenotype (LMM::AnalyzeGene)
//   gemma_file_driver -bfile prefix -inproc [1|2] -lmm m ...   kinship, eigendecomposition and association in ONE
//                     process: K never becomes text; wall seconds of every stage on the log line
//
// following PARAM::ReadFiles (src/param.cpp:115-300) and BatchRun (src/gemma.cpp:1900-1926 `-gk`, :1779-1800 `-eigen`,
// :2557-2830 `-lmm`): first pass over the genotypes (device QC) -> kinship over all individuals -> <o>.cXX.txt / .sXX.txt;
// or kinship file -> rows of the analysed individuals -> centre -> eigendecomposition (-> <o>.eigenU/D.txt) -> U^T W,
// U^T y -> null model -> per-SNP association -> <o>.assoc.txt.  One line of key=value pairs on stdout is the log.
// Several phenotype columns (-n 1 2 3) take the multivariate LMM (src/gemma.cpp:2796-2830 -> class MVLMM).
#include <chrono>
#include <cstdlib>
#include <iostream>

#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "gemma_io_host.hpp"

using namespace gemma_amd;

int main(int argc, char **argv) {
  std::string loco, file_gxe, file_gene, file_snps;
  int km = 1, gpus = 1, rank = 0, device = 0;
  bool samegpu = false;
  size_t crt = 0;
  std::string file_geno, file_pheno, file_anno, file_bfile, file_cvt, file_kin, file_kd, file_ku, file_out = "result",
                                                                                                  path_out = "./output";
  std::vector<size_t> p_column;
  int k_mode = 0, a_mode = 0, inproc = 0, lm_mode = 0;
  bool do_eigen = false;
  const auto t_start = std::chrono::steady_clock::now();
  auto lap = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
  QcLevels qc;
  RunLog log;
  for (int i = 0; i < argc; ++i) log.command_line += std::string(i ? " " : "") + argv[i];
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    const bool has = i + 1 < argc && argv[i + 1][0] != '-';
    if (a == "-g" && has) file_geno = argv[++i];
    else if (a == "-p" && has) file_pheno = argv[++i];
    else if (a == "-a" && has) file_anno = argv[++i];
    else if (a == "-bfile" && has) file_bfile = argv[++i];
    else if (a == "-c" && has) file_cvt = argv[++i];
    else if (a == "-k" && has) file_kin = argv[++i];
    else if (a == "-d" && has) file_kd = argv[++i];
    else if (a == "-u" && has) file_ku = argv[++i];
    else if (a == "-n" && has) {
      while (i + 1 < argc && argv[i + 1][0] != '-') p_column.push_back(strtoul(argv[++i], nullptr, 10));
    }
    else if (a == "-o" && has) file_out = argv[++i];
    else if (a == "-outdir" && has) path_out = argv[++i];
    else if (a == "-gk") k_mode = has ? atoi(argv[++i]) : 1;
    else if (a == "-inproc") inproc = has ? atoi(argv[++i]) : 1;
    else if (a == "-lmm") a_mode = has ? atoi(argv[++i]) : 1;
    else if (a == "-lm") lm_mode = has ? atoi(argv[++i]) : 1;
    else if (a == "-eigen") do_eigen = true;
    else if (a == "-loco" && has) loco = argv[++i];
    else if (a == "-gxe" && has) file_gxe = argv[++i];
    else if (a == "-gene" && has) file_gene = argv[++i];
    else if (a == "-snps" && has) file_snps = argv[++i];
    else if (a == "-km" && has) km = atoi(argv[++i]);
    else if (a == "-gpus" && has) gpus = atoi(argv[++i]);
    else if (a == "-samegpu") samegpu = true;
    else if (a == "-crt") crt = 1; // src/gemma.cpp:1398-1399
    else if (a == "-notsnp") qc.maf_level = -1; // src/gemma.cpp:1116-1117
    else if (a == "-maf" && has) qc.maf_level = atof(argv[++i]);
    else if (a == "-miss" && has) qc.miss_level = atof(argv[++i]);
    else if (a == "-hwe" && has) qc.hwe_level = atof(argv[++i]);
    else if (a == "-r2" && has) qc.r2_level = atof(argv[++i]);
    else {
      std::cerr << "unknown or incomplete option " << a << std::endl;
      return 2;
    }
  }
  std::vector<int> id_pipe_r, id_pipe_w; // rank 0 -> rank r: the 128-byte communicator id
  if (gpus > 1 && (a_mode || lm_mode) && !k_mode && !do_eigen && file_gene.empty()) {
    // one process per GPU, forked before anything touches the device; rank r writes <o>.rank<r>.assoc.txt
    if (samegpu) setenv("GEMMA_HIP_COMM", "shm", 1);
    id_pipe_r.assign(gpus, -1);
    id_pipe_w.assign(gpus, -1);
    for (int r = 1; r < gpus; ++r) {
      int fd[2];
      if (pipe(fd) != 0) return 6;
      id_pipe_r[r] = fd[0];
      id_pipe_w[r] = fd[1];
    }
    std::vector<pid_t> kids;
    for (int r = 0; r < gpus; ++r) {
      const pid_t pid = fork();
      if (pid < 0) return 6;
      if (pid == 0) {
        rank = r;
        device = samegpu ? 0 : r;
        file_out += ".rank" + std::to_string(r);
        // keep only the pipe ends this rank uses (rank 0: every write end; rank r: its read end), so that a rank whose peer
        // died sees EOF instead of blocking on a descriptor a sibling still holds open
        for (int q = 1; q < gpus; ++q) {
          if (r != 0) close(id_pipe_w[q]);
          if (q != r) close(id_pipe_r[q]);
        }
        break;
      }
      kids.push_back(pid);
    }
    if ((int)kids.size() == gpus) { // the parent: wait, then concatenate in rank order
      for (int q = 1; q < gpus; ++q) { close(id_pipe_r[q]); close(id_pipe_w[q]); }
      // The ranks meet in collectives that have no time-out (ncclBroadcast, the shm transport's barrier): when one of them
      // leaves early -- an unreadable kinship file on rank 0, a failed eigendecomposition -- the others would wait for ever.
      // So: reap in completion order, and on the first failure end the siblings.
      int bad = 0;
      for (size_t left = kids.size(); left > 0; --left) {
        int st = 0;
        const pid_t done = waitpid(-1, &st, 0);
        if (done < 0) { bad = 1; break; }
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
          bad = 1;
          // a failure every rank meets (no device, unreadable input) ends them all by itself: two seconds of grace so that
          // each can say why, then the ones still waiting in a collective are ended
          std::vector<pid_t> left_kids;
          for (pid_t k : kids)
            if (k != done) left_kids.push_back(k);
          for (int tick = 0; tick < 100 && !left_kids.empty(); ++tick) {
            for (size_t q = 0; q < left_kids.size();) {
              int s2 = 0;
              const pid_t r = waitpid(left_kids[q], &s2, WNOHANG);
              if (r == left_kids[q] || r < 0) left_kids.erase(left_kids.begin() + (long)q);
              else ++q;
            }
            if (!left_kids.empty()) usleep(20000);
          }
          for (pid_t k : left_kids) kill(k, SIGKILL);
          for (pid_t k : left_kids) {
            int s2 = 0;
            waitpid(k, &s2, 0);
          }
          break;
        }
      }
      if (bad) return 7;
      std::ofstream all((path_out + "/" + file_out + ".assoc.txt").c_str(), std::ios::binary);
      for (int r = 0; r < gpus; ++r) {
        const std::string part = path_out + "/" + file_out + ".rank" + std::to_string(r) + ".assoc.txt";
        std::ifstream in(part.c_str(), std::ios::binary);
        all << in.rdbuf();
        in.close();
        remove(part.c_str());
      }
      std::cout << "ranks=" << gpus << std::endl;
      return 0;
    }
  } else {
    gpus = 1;
  }
  try {
    enforce_hip(gemma_hip_init(device, 0), "init");
    if (gpus > 1) { // the communicator: rank 0 makes the id, the pipes carry it
      unsigned char id[GEMMA_HIP_COMM_ID_BYTES];
      if (rank == 0) {
        enforce_hip(gemma_hip_comm_unique_id(id), "comm_unique_id");
        for (int r = 1; r < gpus; ++r)
          if (write(id_pipe_w[r], id, sizeof id) != (ssize_t)sizeof id) return 6;
      } else if (read(id_pipe_r[rank], id, sizeof id) != (ssize_t)sizeof id) {
        return 6;
      }
      enforce_hip(gemma_hip_comm_init(id, rank, gpus), "comm_init");
    }
    // ---- PARAM::ReadFiles ---------------------------------------------------------------------------------------
    CvtPhen cp;
    std::vector<SNPINFO> snpInfo;
    std::vector<int> indicator_snp;
    std::map<std::string, int> mapID2num;
    std::map<std::string, std::string> mapRS2chr;
    std::map<std::string, long int> mapRS2bp;
    std::map<std::string, double> mapRS2cM;
    std::set<std::string> setSnps;
    if (!file_snps.empty() && !ReadFile_snps(file_snps, setSnps)) return 3; // src/param.cpp:147-153
    if (p_column.empty()) p_column.push_back(1);
    const std::vector<size_t> &cols = p_column;
    const size_t n_ph = cols.size();
    if (!file_cvt.empty() && !ReadFile_cvt(file_cvt, cp.indicator_cvt, cp.cvt, cp.n_cvt)) return 3;
    if (cp.indicator_cvt.empty()) cp.n_cvt = 1;
    if (!file_gxe.empty() && !ReadFile_column(file_gxe, cp.indicator_gxe, cp.gxe, 1)) return 3; // src/param.cpp:232-236
    size_t ns_test = 0;
    std::vector<double> Wb, Yb;
    if (!file_bfile.empty()) {
      if (!ReadFile_bim(file_bfile + ".bim", snpInfo)) return 3;
      if (!(file_pheno.empty() ? ReadFile_fam(file_bfile + ".fam", cp.indicator_pheno, cp.pheno, mapID2num, cols)
                               : ReadFile_pheno(file_pheno, cp.indicator_pheno, cp.pheno, cols)))
        return 3;
    } else if (!file_geno.empty()) {
      if (!file_anno.empty() && !ReadFile_anno(file_anno, mapRS2chr, mapRS2bp, mapRS2cM)) return 3;
      if (!ReadFile_pheno(file_pheno, cp.indicator_pheno, cp.pheno, cols)) return 3;
    } else if (!file_gene.empty()) { // src/param.cpp:441-470: phenotypes, then the gene ids
      if (!ReadFile_pheno(file_pheno, cp.indicator_pheno, cp.pheno, cols)) return 3;
    } else {
      std::cerr << "need -g/-p, -gene/-p or -bfile" << std::endl;
      return 2;
    }
    cp.ProcessCvtPhen();
    if (cp.error) return 3;
    cp.CopyCvtPhen(Wb, Yb);
    const size_t ni_total = cp.indicator_idv.size(), ni_test = cp.ni_test, n_cvt = cp.n_cvt;
    Matrix W = matrix_view(Wb.data(), ni_test, n_cvt);
    // -inproc: the order of the eigendecomposition is known from here on.  Its device workspace (~5 n^2 doubles; hipMalloc of such sizes
    // costs 25-50 ms per GB) is reserved on a helper thread while the first pass reads the genotype file (gemma_hip_eigh_reserve touches
    // only the solver's own pool: the one entry point of the library that may run beside another one); joined before the kinship stage.
    std::thread reserve_thr;
    if (inproc && k_mode == 0 && a_mode != 0 && lm_mode == 0 && ni_test >= 8000 && getenv("GEMMA_DRIVER_NO_RESERVE") == nullptr)
      reserve_thr = std::thread([ni_test] { (void)gemma_hip_eigh_reserve(ni_test); });
    struct JoinGuard {
      std::thread &t;
      ~JoinGuard() { if (t.joinable()) t.join(); }
    } reserve_guard{reserve_thr};
    size_t ng_total = 0;
    if (!file_gene.empty()) {
      if (!ReadFile_gene(file_gene, snpInfo, ng_total)) return 3;
    } else if (!file_bfile.empty()) {
      if (!ReadFile_bed(file_bfile + ".bed", setSnps, &W, cp.indicator_idv, indicator_snp, snpInfo, qc.maf_level,
                        qc.miss_level, qc.hwe_level, qc.r2_level, ns_test))
        return 3;
    } else {
      if (!ReadFile_geno(file_geno, setSnps, &W, cp.indicator_idv, indicator_snp, qc.maf_level, qc.miss_level,
                         qc.hwe_level, qc.r2_level, mapRS2chr, mapRS2bp, mapRS2cM, snpInfo, ns_test))
        return 3;
    }
    std::cout << "ni_total=" << ni_total << " ni_test=" << ni_test << " n_cvt=" << n_cvt
              << " ns_total=" << indicator_snp.size() << " ns_test=" << ns_test;
    std::set<std::string> setKSnps, setGWASnps; // src/param.cpp:497-500
    if (!loco.empty()) {
      LOCO_set_Snps(setKSnps, setGWASnps, mapRS2chr, loco);
      std::cout << " ksnps=" << setKSnps.size() << " gwasnps=" << setGWASnps.size();
    }
    if (reserve_thr.joinable()) reserve_thr.join();
    if (inproc) std::cout << " t_first_pass=" << lap();
    log.ni_total = ni_total; log.ni_test = ni_test; log.n_cvt = n_cvt; log.n_ph = n_ph;
    log.ns_total = indicator_snp.size(); log.ns_test = ns_test;

    // ---- -lm (src/gemma.cpp:2475-2555): no kinship ------------------------------------------------------------------
    if (lm_mode) {
      Vector yv = vector_view(Yb.data(), ni_test);
      LM cLm;
      cLm.a_mode = 50 + lm_mode;
      cLm.file_bfile = file_bfile;
      cLm.file_geno = file_geno;
      cLm.path_out = path_out;
      cLm.file_out = file_out;
      cLm.ni_total = ni_total;
      cLm.indicator_idv = cp.indicator_idv;
      cLm.indicator_snp = indicator_snp;
      cLm.snpInfo = std::move(snpInfo);
      cLm.shard_rank = rank;
      cLm.shard_world = gpus;
      if (!file_bfile.empty()) cLm.AnalyzePlink(&W, &yv);
      else AnalyzeBimbam(cLm, &W, &yv);
      cLm.WriteFiles();
      log.a_mode = 50 + lm_mode;
      log.time_total = lap() / 60.0;
      if (rank == 0) log.Write(path_out, gpus > 1 ? file_out.substr(0, file_out.rfind(".rank")) : file_out);
      std::cout << " snps=" << cLm.sumStat.size() << std::endl;
      gemma_hip_shutdown();
      return 0;
    }

    // ---- -gk (src/gemma.cpp:1900-1926) ----------------------------------------------------------------------------
    if (k_mode) {
      std::vector<double> Kb(ni_total * ni_total, 0.0);
      Matrix K = matrix_view(Kb.data(), ni_total, ni_total);
      const bool ok = file_bfile.empty() ? BimbamKinThreaded(file_geno, indicator_snp, k_mode, &K, setKSnps, &snpInfo)
                                         : PlinkKin(file_bfile + ".bed", indicator_snp, k_mode, 0, &K);
      if (!ok) return 4;
      log.time_G = lap() / 60.0;
      if (!WriteMatrix(&K, path_out + "/" + file_out + (k_mode == 1 ? ".cXX.txt" : ".sXX.txt"))) return 4;
      log.a_mode = 20 + k_mode;
      log.time_total = lap() / 60.0;
      log.Write(path_out, file_out);
      std::cout << std::endl;
      gemma_hip_shutdown();
      return 0;
    }

    // ---- eigen pairs: from -k (centre + decompose) or from -d / -u ------------------------------------------------
    // With -inproc, and with -k on several GPUs, U and eval stay on the device (the library's kept chain): U is then an
    // empty view and cLmm.kept_U says so; only -eigen copies them out.
    std::vector<double> Ub, evalb(ni_test);
    bool kept = false, eig_sharded = false;
    double trace_G = 0.0;
    bool error = false;
    Vector eval = vector_view(evalb.data(), ni_test);
    if (inproc) { // -gk and -lmm in one process: K stays binary AND on the device (changes K at the 1e-10 level)
      Matrix K = matrix_view(nullptr, ni_total, ni_total);
      KinKeep kk;
      kk.keep = true; kk.rank = rank; kk.world = gpus;
      const bool ok = file_bfile.empty() ? BimbamKinThreaded(file_geno, indicator_snp, inproc, &K, setKSnps, &snpInfo, kk)
                                         : PlinkKin(file_bfile + ".bed", indicator_snp, inproc, 0, &K, kk);
      if (!ok) return 4;
      std::cout << " t_kinship=" << lap();
      // sub-selection, centring, eigendecomposition.  Several ranks: every one holds the all-reduced K, so the decomposition is
      // a collective whose back-transformations are shared out (GEMMA_HIP_EIGH_SHARD=0: rank 0 alone, then the broadcast below)
      const char *esh = getenv("GEMMA_HIP_EIGH_SHARD");
      eig_sharded = gpus > 1 && !(esh && esh[0] == '0');
      if (eig_sharded) trace_G = EigenDecompKeptSharded(cp.indicator_idv, &eval);
      else if (rank == 0) trace_G = EigenDecompKept(cp.indicator_idv, &eval);
      kept = true;
      std::cout << " t_eigen=" << lap();
    } else if (!file_kin.empty() && gpus > 1) {
      if (rank == 0) { // rank 0 alone reads the kinship file and decomposes
        std::vector<double> Gb(ni_test * ni_test);
        Matrix G = matrix_view(Gb.data(), ni_test, ni_test);
        if (km == 2) ReadFile_kin_km2(file_kin, cp.indicator_idv, mapID2num, error, &G);
        else ReadFile_kin_threaded(file_kin, cp.indicator_idv, error, &G);
        if (error) return 5;
        CenterMatrix(&G);
        enforce_hip(gemma_hip_eigh_keep(G.data, ni_test, evalb.data(), &trace_G), "EigenDecomp_Zeroed (kept)");
      }
      kept = true;
    } else if (!file_kin.empty()) {
      Ub.resize(ni_test * ni_test);
      Matrix U1 = matrix_view(Ub.data(), ni_test, ni_test);
      std::vector<double> Gb(ni_test * ni_test);
      Matrix G = matrix_view(Gb.data(), ni_test, ni_test);
      if (km == 2) ReadFile_kin_km2(file_kin, cp.indicator_idv, mapID2num, error, &G);
      else ReadFile_kin_threaded(file_kin, cp.indicator_idv, error, &G);
      if (error) return 5;
      CenterMatrix(&G);
      const double t_e0 = lap();
      trace_G = EigenDecomp_Zeroed(&G, &U1, &eval, 0);
      log.time_eigen = (lap() - t_e0) / 60.0;
    } else if (!file_kd.empty() && !file_ku.empty()) {
      Ub.resize(ni_test * ni_test);
      Matrix U1 = matrix_view(Ub.data(), ni_test, ni_test);
      ReadFile_eigenU_threaded(file_ku, error, &U1);
      ReadFile_eigenD(file_kd, error, &eval);
      if (error) return 5;
      for (size_t i = 0; i < ni_test; ++i) { // src/gemma.cpp:2640-2647
        if (evalb[i] < 1e-10) evalb[i] = 0;
        trace_G += evalb[i];
      }
      trace_G /= (double)ni_test;
    } else {
      std::cerr << "need -gk, -inproc, -k or -d/-u" << std::endl;
      return 2;
    }
    if (kept && gpus > 1 && !eig_sharded) { // the ONE broadcast of (U, eval); the other ranks then fetch eval (n doubles) for the null model
      enforce_hip(gemma_hip_kept_bcast(0, &trace_G), "kept_bcast");
      if (rank != 0) enforce_hip(gemma_hip_kept_U_get(nullptr, evalb.data()), "kept_U_get");
    }
    if (kept && do_eigen) { // -eigen wants the artefacts on disk
      Ub.resize(ni_test * ni_test);
      enforce_hip(gemma_hip_kept_U_get(Ub.data(), nullptr), "kept_U_get");
    }
    Matrix U = matrix_view(Ub.empty() ? nullptr : Ub.data(), ni_test, ni_test);
    std::cout << " trace_G=" << std::setprecision(12) << trace_G;
    if (do_eigen) { // src/gemma.cpp:1779-1800
      if (!WriteEigen(&U, &eval, path_out, file_out)) return 5;
      std::cout << std::endl;
      gemma_hip_shutdown();
      return 0;
    }
    if (!a_mode) {
      std::cerr << "nothing to do" << std::endl;
      return 2;
    }

    // ---- -lmm (src/gemma.cpp:2699-2830) ---------------------------------------------------------------------------
    std::vector<double> UtWb(ni_test * n_cvt), Utyb(ni_test * n_ph);
    Matrix Y = matrix_view(Yb.data(), ni_test, n_ph), UtW = matrix_view(UtWb.data(), ni_test, n_cvt),
           UtY = matrix_view(Utyb.data(), ni_test, n_ph);
    if (kept) {
      CalcUtXKept(&W, &UtW);
      CalcUtXKept(&Y, &UtY);
    } else {
      CalcUtX(&U, &W, &UtW);
      CalcUtX(&U, &Y, &UtY);
    }
    if (n_ph > 1) { // src/gemma.cpp:2796-2830: MVLMM
      MVLMM cMv;
      cMv.file_geno = file_geno;
      cMv.a_mode = a_mode;
      cMv.crt = crt;
      cMv.file_bfile = file_bfile;
      cMv.path_out = path_out;
      cMv.file_out = file_out;
      cMv.ni_total = ni_total;
      cMv.indicator_idv = cp.indicator_idv;
      cMv.indicator_snp = indicator_snp;
      cMv.snpInfo = std::move(snpInfo);
      cMv.shard_rank = rank;
      cMv.shard_world = gpus;
      cMv.kept_U = kept;
      const double t_a0 = lap();
      if (!file_gxe.empty()) { // src/gemma.cpp:2840-2851
        if (file_bfile.empty()) {
          std::cerr << "-gxe takes -bfile input" << std::endl;
          return 2;
        }
        std::vector<double> envb;
        cp.CopyGxe(envb);
        Vector env = vector_view(envb.data(), envb.size());
        // U^T env as covariate n_cvt + 1 of the null model (src/mvlmm.cpp:4492-4494)
        std::vector<double> Uteb(ni_test), UtWeb(ni_test * (n_cvt + 1));
        Matrix E = matrix_view(envb.data(), ni_test, 1), UtE = matrix_view(Uteb.data(), ni_test, 1);
        if (kept) CalcUtXKept(&E, &UtE);
        else CalcUtX(&U, &E, &UtE);
        for (size_t i = 0; i < ni_test; ++i) {
          for (size_t j = 0; j < n_cvt; ++j) UtWeb[i * (n_cvt + 1) + j] = UtWb[i * n_cvt + j];
          UtWeb[i * (n_cvt + 1) + n_cvt] = Uteb[i];
        }
        Matrix UtWe = matrix_view(UtWeb.data(), ni_test, n_cvt + 1);
        cMv.AnalyzePlinkGXE(&U, &eval, &UtW, &UtWe, &UtY, &env);
      } else if (!file_bfile.empty()) cMv.AnalyzePlink(&U, &eval, &UtW, &UtY);
      else AnalyzeBimbam(cMv, &U, &eval, &UtW, &UtY);
      const double t_a1 = lap();
      cMv.WriteFiles();
      std::cout << " logl_remle_H0=" << cMv.logl_remle_H0 << " logl_mle_H0=" << cMv.logl_mle_H0;
      if (inproc)
        std::cout << " t_assoc=" << t_a1 << " t_written=" << lap() << " assoc_seconds=" << t_a1 - t_a0
                  << " assoc_snps_per_s=" << (double)(cMv.sumStat.size() / cMv.stride()) / (t_a1 - t_a0);
      std::cout << " snps=" << cMv.sumStat.size() / cMv.stride() << std::endl;
      gemma_hip_shutdown();
      return 0;
    }
    Vector Uty = vector_view(Utyb.data(), ni_test);
    const NullModel nm = CalcLambdaNull(&eval, &UtW, &Uty, 1e-5, 1e5, 10, trace_G);
    std::cout << " l_mle_null=" << nm.l_mle_null << " logl_mle_H0=" << nm.logl_mle_H0 << " l_remle_null=" << nm.l_remle_null
              << " logl_remle_H0=" << nm.logl_remle_H0 << " pve=" << nm.pve_null << " se_pve=" << nm.pve_se_null
              << " vg=" << nm.vg_remle_null << " ve=" << nm.ve_remle_null;
    LMM cLmm;
    cLmm.a_mode = a_mode;
    cLmm.file_bfile = file_bfile;
    cLmm.file_geno = file_geno;
    cLmm.file_gene = file_gene;
    cLmm.path_out = path_out;
    cLmm.file_out = file_out;
    cLmm.ni_total = ni_total;
    cLmm.indicator_idv = cp.indicator_idv;
    cLmm.indicator_snp = indicator_snp;
    cLmm.snpInfo = std::move(snpInfo); // not read again below
    cLmm.setGWASnps = setGWASnps;
    cLmm.shard_rank = rank;
    cLmm.shard_world = gpus;
    cLmm.kept_U = kept;
    cLmm.l_mle_null = nm.l_mle_null;
    cLmm.logl_mle_H0 = nm.logl_mle_H0;
    const double t_a0 = lap();
    if (!file_gene.empty()) { // src/gemma.cpp:2675-2690: y is the predictor, the genes are the phenotypes
      AnalyzeGene(cLmm, &U, &eval, &UtW, &Uty, ng_total);
    } else if (!file_gxe.empty()) { // src/gemma.cpp:2809-2827
      if (file_bfile.empty()) {
        std::cerr << "-gxe takes -bfile input (the reference's BIMBAM GXE reader cannot open its file, src/lmm.cpp:2289)" << std::endl;
        return 2;
      }
      std::vector<double> envb;
      cp.CopyGxe(envb);
      Vector env = vector_view(envb.data(), envb.size());
      cLmm.AnalyzePlinkGXE(&U, &eval, &UtW, &Uty, &env);
    } else if (!file_bfile.empty()) cLmm.AnalyzePlink(&U, &eval, &UtW, &Uty);
    else AnalyzeBimbam(cLmm, &U, &eval, &UtW, &Uty);
    const double t_a1 = lap();
    cLmm.WriteFiles();
    log.a_mode = a_mode;
    log.have_null = true;
    log.null = nm;
    log.time_UtX = cLmm.time_UtX;
    log.time_opt = cLmm.time_opt;
    log.time_total = lap() / 60.0;
    if (rank == 0) log.Write(path_out, gpus > 1 ? file_out.substr(0, file_out.rfind(".rank")) : file_out);
    if (inproc)
      std::cout << " t_null=" << t_a0 << " t_assoc=" << t_a1 << " t_written=" << lap() << " assoc_seconds=" << t_a1 - t_a0
                << " assoc_snps_per_s=" << (double)cLmm.sumStat.size() / (t_a1 - t_a0)
                << " gpu_min_UtX=" << cLmm.time_UtX << " gpu_min_opt=" << cLmm.time_opt;
    std::cout << " snps=" << cLmm.sumStat.size() << std::endl;
    gemma_hip_shutdown();
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
