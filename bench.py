#!/usr/bin/env python
"""bench.py -- SNPs/s of `-lmm 1` (Wald) at n = 20 000 on N x MI355X  (BASELINE.json metric).

A step = one pass of the hot path over one block of B SNPs (B = LMM_BATCH_SIZE = 20000, the
reference's own batch, src/lmm.h:33) with the block's PLINK 2-bit genotypes already resident in
HBM: ingest (2-bit decode + mean imputation) -> UtX = X U (exact int8-digit MFMA products for hard
calls; fp64 MFMA GEMM for dosages) -> per-SNP lambda search + Wald test -> 64 B/SNP SUMSTAT left in HBM.  Nothing is skipped or cached between steps
(every step gets its own genotype block).

Untimed setup on rank 0: synthetic genotypes -> kinship K (this library's SYRK path) -> centring
-> eigendecomposition -> U^T W, U^T y -> null model; then ONE broadcast of (U, eval, UtW, Uty) to
the other ranks (torch.distributed, backend nccl = RCCL).  Ranks then work on disjoint SNP blocks
with no collective in the timed region (weak scaling: B SNPs per rank per step).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (UtX GEMM): algorithmic flops / hipEvent-measured duration
  cpu_baseline -- the reference's own LMM::Analyze (kind "reference": oracle/_ref/libgemma_ref.so, the reference's
                  sources compiled unchanged) timed on this box's host cores on a bounded SNP sample; the oracle
                  (kind "port") beside it, and alone when that library is absent.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (AMD CDNA4 datasheet; v_mfma_f64_16x16x4 = 64 clk)
INT8_MFMA_PEAK_TOPS = 5000.0  # dense int8 = 2 x the 2.5 PFLOP/s bf16 dense peak (MI355X_MICROARCH.md: "i8 ~ 2x bf16
                              # (2xK)"; micro-benchmark ceiling there 4404 TOP/s for 32x32x32)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--individuals", dest="n", type=int, default=20000, help="analysed individuals (n)")
    ap.add_argument("--batch", type=int, default=20000, help="SNPs per step and per rank")
    ap.add_argument("--kin-snps", type=int, default=100000, help="SNPs used for the kinship matrix (setup; SURVEY 8d: >= 100 000)")
    ap.add_argument("--eigen", default="auto", choices=["auto", "gemma", "torch"])
    ap.add_argument("--cpu-sample", type=int, default=2560, help="SNPs in the CPU baseline sample (0 = skip)")
    ap.add_argument("--a-mode", type=int, default=1)
    ap.add_argument("--fp64-steps", type=int, default=-1,
                    help="steps of the fp64 MFMA GEMM path beside the timed region, on the same blocks (-1, default: min(--steps, 10); 0 = skip)")
    ap.add_argument("--dosage-steps", type=int, default=2,
                    help="extra untimed-region steps on BIMBAM-style fixed-point dosages (k/100, fp64 input): the int8-digit "
                         "dosage path, checked against the fp64 GEMM path on the same block (0 = skip)")
    ap.add_argument("--miss", type=float, default=0.01, help="missing-call rate of the synthetic PLINK blocks (timed region)")
    ap.add_argument("--miss-leg", type=float, default=0.05,
                    help="extra untimed-region steps at this missing-call rate (GEMMA's default -miss ceiling is 0.05): every row then "
                         "has groups of four with 3-4 missing calls, which the sparse mask operand hands to the fp64 fix-up (0 = skip)")
    ap.add_argument("--complete-steps", type=int, default=4,
                    help="extra untimed-region steps on blocks WITHOUT a missing call (hard calls of an imputed panel): the library "
                         "notices on the device and takes the genotype product alone (include/gemma_hip.h: "
                         "gemma_hip_dbg_last_block_missing); 0 = skip")
    ap.add_argument("--lowh2-leg", type=float, default=3e-4,
                    help="extra untimed-region steps on a second phenotype whose variance ratio lambda is this value (next to no "
                         "heritability: lambda-hat in the decades below 1e-3, the common case in human GWAS) -- the per-SNP stage must "
                         "not fall off the table path there (0 = skip)")
    ap.add_argument("--ref-procs", type=int, default=8,
                    help="independent processes of the reference's LMM::Analyze in the cpu_baseline leg (each on its own slice of the "
                         "last timed block; 1 = one in-process call as in rounds 1-2)")
    ap.add_argument("--cpu-setup", type=int, default=1,
                    help="1: also time the reference's setup stages on the host cores -- EigenDecomp_Zeroed (dsyevr) at this n and "
                         "PlinkKin on one batch -- in a child process beside the GPU legs, reported as cpu_baseline.setup (0 = skip)")
    ap.add_argument("--cpu-setup-n", type=int, default=8192,
                    help="order of the leading principal block of the centred kinship the host eigendecomposition runs on (0 = this "
                         "run's n: 334 s at n = 20000 on the GPU box, profiles/r03_cpu_setup_baseline.json, which is why the default "
                         "run measures a block and cites the full-n measurement)")
    ap.add_argument("--cpu-setup-budget", type=float, default=600.0,
                    help="seconds after which the host eigendecomposition is abandoned and reported as unfinished")
    ap.add_argument("--setup-parity", type=int, default=1,
                    help="1: verify the setup stages at the bench's own size (setup_parity in the JSON line): residual and orthogonality of "
                         "the n x n eigendecomposition through the library's GEMM, its eigenvalues against rocSOLVER's, and -- with "
                         "--cpu-setup -- the kinship of one .bed batch and the eigenvalues of the --cpu-setup-n block against the reference's "
                         "PlinkKin / dsyevr outputs (0 = skip)")
    ap.add_argument("--digits7-steps", type=int, default=-1,
                    help="steps of the STRICT-precision leg beside the timed region's 6 digits at n >= 16384: GEMMA_HIP_I8_FORM=7g6m -- seven "
                         "base-256 digits of U for the genotype product (2^-56 of each column's maximum), the mask product on the upper six: "
                         "U^T x at or below the fp64 GEMM's own rounding error (tests/test_gpu_at_size.py).  -1 (default): as many steps "
                         "as the timed region, the same blocks; reported as value_strict at equal standing (0 = skip)")
    ap.add_argument("--c4-leg", type=int, default=1,
                    help="1: after everything else, BASELINE config 4's per-GPU piece at its real size in a child process (n = 50000, one "
                         "20000-SNP block per step, 2 steps, kinship from 40000 SNPs, its own eigendecomposition, 64 SNPs against the "
                         "oracle / the reference) -- ~80 s, reported under \"c4_leg\", never part of `value` (0 = skip)")
    ap.add_argument("--child", default="", help=argparse.SUPPRESS)
    ap.add_argument("--child-spec", default="", help=argparse.SUPPRESS)
    ap.add_argument("--pipeline", type=int, default=0,
                    help="0 (default): gemma_hip_lmm_batch_d, one stream, one block at a time.  1: the timed steps go through "
                         "gemma_hip_lmm_batch_pipe_d -- the int8 product of block i + 1 on one CU partition beside the digit combine and per-SNP "
                         "stage of block i on the other (GEMMA_HIP_PIPE_CUS, default 64); the timed region ends with the flush.  Measured in round "
                         "5: no gain on a power-limited product (profiles/r05_pipeline_partition.txt)")
    ap.add_argument("--config", type=int, default=0,
                    help="4: BASELINE config 4 with one flag -- n = 50000, p = 500000 SNPs, -gk + -lmm 1, the SNPs (kinship AND association) "
                         "split over the ranks: steps / batch / kin-snps are derived (ceil(p / ranks / 20000) steps of equal blocks per rank), "
                         "scaling is 'strong', the optional legs are off.  0 (default): the headline config (configs[2])")
    ap.add_argument("--seed", type=int, default=20000)
    ap.add_argument("--state-file", default="",
                    help="measurement aid: keep the setup's result (U, eval, UtW, Uty, null scalars) in this file -- written "
                         "when absent, loaded when present -- so that a profiler run (rocprofv3 --pmc crashes inside the "
                         "eigensolver's ~80 000 launches) can start at the timed region")
    ap.add_argument("--e2e-snps", type=int, default=1000000,
                    help="end-to-end leg after the timed region (0 = skip; default 1000000 = BASELINE config 3 in full since round 5): a synthetic PLINK set of this many SNPs on disk -> "
                         "tests/cpp/gemma_file_driver -inproc (first pass, kinship, eigen, -lmm, .assoc.txt), wall seconds "
                         "per stage reported under \"e2e\" (0 = skip)")
    return ap.parse_args()


def ctypes_digits(L, n):
    """base-256 digits of U the int8 product uses at this n (csrc/i8gemm.hip.h: 7, or 6 from n = 16384 up)"""
    import ctypes
    dg = ctypes.c_int(7)
    L.lib().gemma_hip_dbg_i8_digits(n, ctypes.byref(dg))
    return dg.value


def synth_block(torch, n, l, gen, dev, miss=0.01, fst=0.05):
    """l SNPs x n individuals, Balding-Nichols two-population genotypes, 1 % missing, packed as PLINK
    2-bit rows (codes: 2 -> 00, 1 -> 10, 0 -> 11, missing -> 01; low bits first)."""
    maf = torch.empty(l, device=dev).uniform_(0.05, 0.5, generator=gen)
    a = maf * (1 - fst) / fst
    b = (1 - maf) * (1 - fst) / fst
    # Beta(a,b) via two gammas (torch._standard_gamma takes no generator: seeded globally)
    ga = torch._standard_gamma(a.repeat(2, 1))
    gb = torch._standard_gamma(b.repeat(2, 1))
    psub = ga / (ga + gb)  # 2 x l
    half = n // 2
    pfreq = torch.cat([psub[0].unsqueeze(1).expand(l, half), psub[1].unsqueeze(1).expand(l, n - half)], dim=1)
    u1 = torch.rand((l, n), device=dev, generator=gen)
    u2 = torch.rand((l, n), device=dev, generator=gen)
    g = (u1 < pfreq).to(torch.uint8) + (u2 < pfreq).to(torch.uint8)  # 0/1/2
    del u1, u2, pfreq
    code = torch.where(g == 2, torch.zeros_like(g), torch.where(g == 1, torch.full_like(g, 2), torch.full_like(g, 3)))
    m = torch.rand((l, n), device=dev, generator=gen) < miss
    code = torch.where(m, torch.ones_like(code), code)
    nb = (n + 3) // 4
    pad = torch.zeros((l, nb * 4), dtype=torch.uint8, device=dev)
    pad[:, :n] = code
    packed = pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)
    return packed.contiguous()


def run_child(args):
    """Helper processes of the CPU legs (no torch, no GPU): `ref_lmm` = the reference's LMM::Analyze on one slice of SNP rows;
    `ref_setup` = the reference's PlinkKin on one .bed batch and its EigenDecomp_Zeroed on the centred kinship."""
    import numpy as np
    from oracle import oracle as O
    spec = json.load(open(args.child_spec))
    if args.child == "ref_lmm":
        U = np.load(spec["U"], mmap_mode="r")
        ev, UtW, Uty, X = (np.load(spec[k]) for k in ("ev", "UtW", "Uty", "X"))
        t0 = time.perf_counter()
        r = O.ref_lmm_analyze(spec["a_mode"], U, ev, UtW, Uty, X, l_mle_null=spec["l_mle_null"], logl_mle_H0=spec["logl_mle_H0"])
        dt = time.perf_counter() - t0
        np.save(spec["out"], np.stack([r[k] for k in r.dtype.names], axis=1))
        json.dump({"seconds": dt, "snps": int(X.shape[0]), "threads": O.ref_blas_threads()}, open(spec["out"] + ".json", "w"))
    elif args.child == "ref_setup":
        res = {"threads": O.ref_blas_threads()}
        n = spec["n"]
        if spec.get("bed"):
            t0 = time.perf_counter()
            K = O.ref_plink_kin(spec["bed"], n, spec["bed_snps"], 1)
            res["plink_kin"] = {"seconds": round(time.perf_counter() - t0, 2), "snps": spec["bed_snps"],
                                "checksum": float(np.abs(K).sum())}
            if spec.get("K_gpu") and os.path.exists(spec["K_gpu"]):
                # the GPU's kinship of the SAME .bed batch (written by the parent): every entry against the reference's
                Kg = np.load(spec["K_gpu"], mmap_mode="r")
                kmax, dmax, rmax = float(np.abs(K).max()), 0.0, 0.0
                for r0 in range(0, n, 1024):
                    a, b = K[r0:r0 + 1024], np.asarray(Kg[r0:r0 + 1024])
                    d = np.abs(a - b)
                    dmax = max(dmax, float(d.max()))
                    big = np.abs(a) > 1e-3 * kmax  # entrywise relative error where the entry is not a cancellation residue
                    if big.any():
                        rmax = max(rmax, float((d[big] / np.abs(a[big])).max()))
                res["plink_kin"].update({"kin_max_abs_diff_over_max": dmax / kmax, "kin_max_rel_entries_above_1e-3_max": rmax,
                                         "kin_entries_compared": int(n) * int(n), "kin_max_abs": kmax})
            json.dump(res, open(spec["out"], "w"))
            del K
        G = np.load(spec["G"])
        t0 = time.perf_counter()
        U, ev, tr = O.ref_eigen_decomp_zeroed(G)
        res["eigen"] = {"seconds": round(time.perf_counter() - t0, 2), "n": n, "n_block": int(G.shape[0]), "eval_sum": float(ev.sum()),
                        "eval_max": float(ev.max())}
        np.save(spec["out"] + ".eval.npy", np.asarray(ev, dtype=np.float64))  # the parent compares the GPU's eigenvalues of the same block
        json.dump(res, open(spec["out"], "w"))
    else:
        raise SystemExit("unknown --child " + args.child)


def shm_dir():
    import tempfile
    for d in ("/dev/shm", "/tmp"):
        if os.path.isdir(d) and os.access(d, os.W_OK):
            return tempfile.mkdtemp(prefix="gemma_bench_", dir=d)
    return tempfile.mkdtemp(prefix="gemma_bench_")


def self_launch(args):
    """`python bench.py --gpus N` with no torch.distributed environment: launch the N ranks ourselves, exactly the way the
    driver does (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1), and pass rank 0's JSON line
    through.  With WORLD_SIZE already set (the driver's own torchrun launch) this is never taken."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def setup_multi_rank(args, np, torch, dist, api, L, gdist, rank, world, dev, gen, n, B, setup_info, null, UtW, Uty):
    """The setup of an N-rank run as the real flow has it (tests/cpp/gemma_file_driver.cpp -gpus N): every rank adds ITS
    share of the kinship SNPs, gemma_hip_kin_end_keep(allreduce) is the ONE ncclAllReduce of the n^2 sums, the decomposition is the
    collective gemma_hip_eigh_kept_K_sharded (K, U and eval never leave the device), every rank rotates the covariates and the
    phenotype on its kept U, and ONE small ncclBroadcast makes rank 0's (UtW, Uty, null model) everybody's.  Every clock is this
    rank's wall time around synchronised library calls; rank 0's go into the line, the slowest rank's beside them."""
    wa = torch.ones((64, 64), dtype=torch.float64, device=dev)
    api.fast_dgemm("N", "N", 1.0, wa, wa, 0.0, torch.empty_like(wa))  # the library's code object is loaded before any clock starts
    del wa
    lo, hi = gdist.shard_range(args.kin_snps, rank, world)
    torch.cuda.synchronize()
    t_kin = 0.0
    t0 = time.time()
    api.kin_begin(n, 1)
    torch.cuda.synchronize()
    t_kin += time.time() - t0
    done = lo
    while done < hi:
        l = min(B, hi - done)
        blk = synth_block(torch, n, l, gen, dev)
        torch.cuda.synchronize()
        t0 = time.time()
        api.kin_add(blk, L.GENO_PLINK_2BIT)
        torch.cuda.synchronize()
        t_kin += time.time() - t0
        done += l
        del blk
    gdist.ctl_barrier()
    t0 = time.time()
    ns = api.kin_end_keep(allreduce=True)
    torch.cuda.synchronize()
    t_all = time.time() - t0
    prev_t = os.environ.get("GEMMA_HIP_EIGH_TIMING")
    os.environ["GEMMA_HIP_EIGH_TIMING"] = "1"
    t0 = time.time()
    try:
        api.eigh_reserve(n)  # the solver's workspace ahead of its clock (a stage of its own: eigen_workspace_reserve_s)
    except L.GemmaHipError:
        pass
    torch.cuda.synchronize()
    t_res = time.time() - t0
    t0 = time.time()
    evh, trace = api.EigenDecomp_kept_K(n, None, sharded=True)
    torch.cuda.synchronize()
    t_eig = time.time() - t0
    t0 = time.time()
    api.eigh_release()
    torch.cuda.synchronize()
    t_rel = time.time() - t0
    if prev_t is None:
        os.environ.pop("GEMMA_HIP_EIGH_TIMING", None)
    else:
        os.environ["GEMMA_HIP_EIGH_TIMING"] = prev_t
    import ctypes
    t8 = (ctypes.c_double * 8)()
    L.lib().gemma_hip_dbg_eigh_last(t8)
    # phenotype: 50 causal SNPs + noise from a generator seeded alike on every rank; rank 0's rotation is the one that counts (below)
    gc = torch.Generator(device=dev).manual_seed(args.seed + 77)
    cb = synth_block(torch, n, 50, gc, dev)
    codes = (cb.unsqueeze(2) >> torch.tensor([0, 2, 4, 6], device=dev, dtype=torch.uint8)) & 3
    codes = codes.reshape(50, -1)[:, :n]
    gv = torch.where(codes == 0, 2.0, torch.where(codes == 2, 1.0, 0.0)).to(torch.float64)
    y = gv.T @ (torch.randn(50, dtype=torch.float64, device=dev, generator=gc) * 0.15)
    y += torch.randn(n, dtype=torch.float64, device=dev, generator=gc) * y.std().clamp_min(1e-3)
    UtWh = api.CalcUtX_kept(np.ones((n, 1)))
    Utyh = api.CalcUtX_kept(y.cpu().numpy())
    nm = api.CalcLambdaNull(evh, UtWh, Utyh, trace_G=trace)
    UtW.copy_(torch.from_numpy(UtWh)); Uty.copy_(torch.from_numpy(Utyh))
    null[0], null[1] = nm["l_mle_null"], nm["logl_mle_H0"]
    torch.cuda.synchronize()
    t0 = time.time()
    gdist.broadcast_state_native([UtW, Uty, null])  # ONE ncclBroadcast: rank 0's rotated covariates / phenotype / null model
    torch.cuda.synchronize()
    t_bc = time.time() - t0
    # the slowest rank's clocks beside rank 0's (one SUM all-reduce of a world x 4 table)
    flat = [0.0] * (4 * world)
    flat[4 * rank:4 * rank + 4] = [t_kin, t_all, t_eig, t_bc]
    tab = np.array(gdist.ctl_allreduce(flat, "sum")).reshape(world, 4)
    transport = api.comm_info()[2]
    two = int(t8[7]) == 2
    setup_info.update({
        "flow": "every rank: kin_add of its %d-SNP share -> gemma_hip_kin_end_keep(allreduce=1) -> gemma_hip_eigh_kept_K_sharded -> "
                "calc_utx_kept -> ONE broadcast of (UtW, Uty, null) -> lmm_setup_kept" % (hi - lo),
        "kinship_s": round(t_kin, 3), "kinship_snps_per_rank": [int(gdist.shard_range(args.kin_snps, r, world)[1] -
                                                                      gdist.shard_range(args.kin_snps, r, world)[0]) for r in range(world)],
        "kinship_snps_all_ranks": int(ns),
        "allreduce_s": round(t_all, 3),
        "allreduce": "%s of the n^2 kinship sums + the SNP count, issued by libgemma_hip.so (gemma_hip_kin_end_keep); %.2f GB per rank"
                     % ("ncclAllReduce" if transport == 1 else "shm test transport's all-reduce", 8.0 * n * n / 1e9),
        "eigen_workspace_reserve_s": round(t_res, 3), "eigen_workspace_release_s": round(t_rel, 3),
        "eigen_s": round(t_eig, 3),
        "eigen": "gemma_hip_eigh_kept_K_sharded (%s)" % (
            "two-stage; collective over %d ranks: reduction and divide & conquer on every rank, back-transformations shared out by "
            "eigenvector, slices exchanged once" % world if two else "one-stage: replicated on every rank, nothing is exchanged"),
        "eigen_stages_s": {"reduction": round(t8[0], 3), "bulge_chase": round(t8[1], 3), "divide_conquer": round(t8[2], 3),
                           "backtransform_q2": round(t8[3], 3), "backtransform_q1": round(t8[4], 3), "sort_transpose": round(t8[5], 3)},
        "broadcast_s": round(t_bc, 3),
        "slowest_rank_s": {"kinship_s": round(float(tab[:, 0].max()), 3), "allreduce_s": round(float(tab[:, 1].max()), 3),
                           "eigen_s": round(float(tab[:, 2].max()), 3), "broadcast_s": round(float(tab[:, 3].max()), 3)},
        "null": {k: nm[k] for k in ("l_remle_null", "pve")}})
    api.profile_read(L.STAGE_UTX_GEMM, reset=True)
    return {"broadcast": ("native: ncclBroadcast from libgemma_hip.so's communicator" if transport == 1 else
                          "native: libgemma_hip.so's communicator over its shm test transport (GEMMA_HIP_COMM=shm)") +
                         "; (U, eval) from the collective eigensolver on the all-reduced kept K, not broadcast"}


def apply_config(args, world):
    """--config 4 (BASELINE configs[3]): n = 50 000, p = 500 000 SNPs split over the ranks -- kinship shares and association blocks."""
    if args.config == 0:
        return None
    if args.config != 4:
        raise SystemExit("bench.py: --config %d is not a bench workload (0 = headline, 4 = BASELINE config 4)" % args.config)
    p_total = 500000
    args.n = 50000
    per_rank = (p_total + world - 1) // world
    args.steps = max(1, (per_rank + 19999) // 20000)
    args.batch = (per_rank + args.steps - 1) // args.steps
    args.kin_snps = p_total
    args.warmup = min(args.warmup, 1)
    args.fp64_steps = args.dosage_steps = args.digits7_steps = 0
    args.miss_leg = args.lowh2_leg = 0.0
    args.complete_steps = 0
    args.e2e_snps = 0
    args.c4_leg = 0
    args.cpu_setup = 0
    args.cpu_sample = min(args.cpu_sample, 64)
    args.ref_procs = 1
    return {"config": 4, "p_total": p_total, "snps_per_rank": args.batch * args.steps}


def main():
    args = parse()
    if args.child:
        return run_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is used" % (args.gpus, world), file=sys.stderr)
    # test hooks (1-GPU box): BENCH_FORCE_DEVICE pins every rank to one device, BENCH_DIST_BACKEND=gloo then
    # carries the collectives; the driver's multi-GPU runs use neither (one rank per GPU over RCCL)
    if os.environ.get("BENCH_FORCE_DEVICE"):
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    preset = apply_config(args, world)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # Control plane on gloo (CPU tensors: agreements, barriers, the clocks' exchange), device tensors on nccl = RCCL, whose
        # communicator torch creates lazily at the first device collective -- under the staged start below, not here.  A pure
        # "nccl" group (BENCH_DIST_BACKEND=nccl) initialises eagerly, a pure "gloo" one stages device tensors through the host
        # (the tests' N-ranks-on-one-device route).
        backend = os.environ.get("BENCH_DIST_BACKEND", "cpu:gloo,cuda:nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from gemma_amd import api
    from gemma_amd import _lib as L
    from gemma_amd import dist as gdist

    api.init(local, verbose=0)
    name, n_cu, hbm = api.device_info()
    # ---- staged start of the transports: this run must come back with a line whatever the fabric does.
    #   native      libgemma_hip.so's own communicator (librccl: ncclCommInitRank under a deadline, then ONE KiB through ncclAllReduce /
    #               ncclBroadcast under a deadline) -- the real flow: sharded kinship, one all-reduce, collective eigensolver
    #   torch       rank 0 runs the setup alone, (U, eval, UtW, Uty, null) travel over torch.distributed (its own first contact under a
    #               deadline as well) -- LABELLED as such
    #   replicated  no device transport at all: every rank runs the setup itself; the timed region is communication-free anyway
    # Every decision is an agreement on the control plane, so all ranks take the same branch.
    native = False
    comm_log = {"tried": []}
    setup_mode = "single"
    if world > 1:
        pure_gloo = os.environ.get("BENCH_DIST_BACKEND", "") == "gloo"
        want_native = (not pure_gloo) or os.environ.get("GEMMA_HIP_COMM", "") == "shm"
        if want_native:
            t0 = time.time()
            native = bool(gdist.native_comm_init(timeout=float(os.environ.get("BENCH_COMM_DEADLINE", "180"))))
            comm_log["tried"].append({"stage": "libgemma_hip.so communicator: init", "ok": native, "seconds": round(time.time() - t0, 2),
                                      "error": None if native else gdist.native_comm_error()})
            if native:
                t0 = time.time()
                native = bool(gdist.native_comm_selftest(timeout=float(os.environ.get("BENCH_COMM_DEADLINE", "180")) / 3))
                comm_log["tried"].append({"stage": "libgemma_hip.so communicator: 1 KiB all-reduce + broadcast", "ok": native,
                                          "seconds": round(time.time() - t0, 2), "error": None if native else gdist.native_comm_error()})
        if native:
            setup_mode = "native"
        else:
            t0 = time.time()
            if pure_gloo:
                ok_t, err_t = True, ""
            else:
                ok_t, err_t = gdist.torch_device_selftest(timeout=float(os.environ.get("BENCH_COMM_DEADLINE", "180")) / 3)
            comm_log["tried"].append({"stage": "torch.distributed device collectives (%s): 1 KiB all-reduce + broadcast"
                                               % os.environ.get("BENCH_DIST_BACKEND", "cpu:gloo,cuda:nccl"),
                                      "ok": bool(ok_t), "seconds": round(time.time() - t0, 2), "error": err_t or None})
            setup_mode = "torch" if ok_t else "replicated"
        if want_native and not native:
            comm_log["error"] = "; ".join("%s: %s" % (t["stage"], t["error"]) for t in comm_log["tried"] if not t["ok"])
        comm_log["setup_mode"] = setup_mode
    replicated = setup_mode == "replicated"
    n, B = args.n, args.batch
    torch.manual_seed(args.seed + rank)
    gen = torch.Generator(device=dev).manual_seed(args.seed + 1000 * rank)

    # ------------------------------------------------------------------ setup (untimed)
    t_bench0 = time.time()
    t_setup = time.time()
    U = torch.empty((n, n), dtype=torch.float64, device=dev)
    ev = torch.empty(n, dtype=torch.float64, device=dev)
    UtW = torch.empty((n, 1), dtype=torch.float64, device=dev)
    Uty = torch.empty(n, dtype=torch.float64, device=dev)
    setup_info = {}
    cpu_setup = None
    null = torch.zeros(2, dtype=torch.float64, device=dev)
    loaded = False
    if rank == 0 and args.state_file and os.path.exists(args.state_file):
        st = torch.load(args.state_file, map_location=dev)
        if st["U"].shape == U.shape:
            U.copy_(st["U"]); ev.copy_(st["ev"]); UtW.copy_(st["UtW"]); Uty.copy_(st["Uty"]); null.copy_(st["null"])
            setup_info["state"] = "loaded from " + args.state_file
            loaded = True
        del st
    sent_go = False
    shard_eig = False
    multi_real = world > 1 and native and args.eigen in ("auto", "gemma") and not args.state_file
    kept_state = None
    if multi_real:
        # the real flow, under a deadline and an agreement: a collective that FAILS after the passed self-test (or never returns) must
        # cost this run its native setup, not its line
        fin, res = gdist._with_deadline(lambda: setup_multi_rank(args, np, torch, dist, api, L, gdist, rank, world, dev, gen, n, B, setup_info,
                                                                 null, UtW, Uty),
                                        float(os.environ.get("BENCH_SETUP_DEADLINE", str(90 + 300 * (n / 50000.0) ** 3))))  # ~20 x what the stage takes
        ok_here = bool(fin and isinstance(res, dict))
        if gdist.agree(ok_here):
            kept_state = res
            shard_eig = True
            del U  # (U, eval) are the library's kept buffers in this flow (gemma_hip_eigh_kept_K_sharded); nothing of them is a torch tensor
            U = None
        else:
            why = ("no answer within the setup deadline" if not fin else str(res))
            comm_log["tried"].append({"stage": "native multi-rank setup (sharded kinship -> ncclAllReduce -> collective eigensolver -> ncclBroadcast)",
                                      "ok": False, "error": why if not ok_here else "another rank failed"})
            comm_log["error"] = (comm_log.get("error", "") + "; " if comm_log.get("error") else "") + "native setup: " + \
                                (why if not ok_here else "another rank failed")
            multi_real = False
            native = False
            replicated = True
            setup_mode = comm_log["setup_mode"] = "replicated"
            setup_info.clear()
            if not fin:
                # the helper thread is still inside the library, its stream may never drain: everything from here on runs on a new stream
                torch.cuda.set_stream(torch.cuda.Stream(device=dev))
            else:
                try:
                    api.kept_release()
                except Exception:  # noqa: BLE001
                    pass
                try:
                    L.lib().gemma_hip_comm_finalize()
                except Exception:  # noqa: BLE001
                    pass
            torch.manual_seed(args.seed + rank)
            gen = torch.Generator(device=dev).manual_seed(args.seed + 1000 * rank)
    if replicated:
        # every rank builds the SAME setup from the seeds rank 0 uses; its own blocks of the timed region are drawn afterwards
        torch.manual_seed(args.seed)
        gen = torch.Generator(device=dev).manual_seed(args.seed)
    if (rank == 0 or replicated) and not loaded and not multi_real:
        # kinship_s / eigen_s are the LIBRARY's stages: the synthetic blocks (torch kernels, loaded lazily on their first use: seconds
        # on a box with a cold page cache) and the load of the library's own code object (first launch) stay outside the clocks
        wa = torch.ones((64, 64), dtype=torch.float64, device=dev)
        api.fast_dgemm("N", "N", 1.0, wa, wa, 0.0, torch.empty_like(wa))
        del wa
        K = torch.empty((n, n), dtype=torch.float64, device=dev)
        api.profile_enable(True)
        torch.cuda.synchronize()
        t_kin = 0.0
        t0 = time.time()
        api.kin_begin(n, 1)
        torch.cuda.synchronize()
        t_kin += time.time() - t0
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        done = 0
        while done < args.kin_snps:
            l = min(B, args.kin_snps - done)
            blk = synth_block(torch, n, l, gen, dev)
            torch.cuda.synchronize()
            t0 = time.time()
            api.kin_add(blk, L.GENO_PLINK_2BIT)
            torch.cuda.synchronize()
            t_kin += time.time() - t0
            if done == 0:  # phenotype: 50 causal SNPs of the first block + noise
                codes = (blk[:50].unsqueeze(2) >> torch.tensor([0, 2, 4, 6], device=dev, dtype=torch.uint8)) & 3
                codes = codes.reshape(50, -1)[:, :n]
                gv = torch.where(codes == 0, 2.0, torch.where(codes == 2, 1.0, 0.0)).to(torch.float64)
                beta = torch.randn(50, dtype=torch.float64, device=dev, generator=gen) * 0.15
                y += gv.T @ beta
            done += l
            del blk
        torch.cuda.synchronize()
        t0 = time.time()
        api.kin_end(K)
        torch.cuda.synchronize()
        t_kin += time.time() - t0
        kin_ms, kin_n = api.profile_read(L.STAGE_KIN_GEMM, reset=True)
        setup_info["kinship_s"] = round(t_kin, 3)
        setup_info["kinship_s_what"] = "wall time inside kin_begin / kin_add / kin_end (synchronised); synthetic block generation excluded"
        setup_info["kinship_gemm_tflops"] = round(2.0 * n * n * args.kin_snps / (kin_ms * 1e-3) / 1e12, 2) if kin_ms else None
        kin_int = os.environ.get("GEMMA_HIP_KIN_I8", "1") != "0"  # PLINK hard calls: exact-integer G^T G + sparse correction
        if kin_ms and kin_int:
            # kin_i8.hip.h: G^T G on the int8 MFMA pipe (2 n^2 p integer ops, full square), fp64 accumulation of the int32
            # block product, and the missing-call correction (VALU-bound gather over the ~1 % missing calls).  There is no
            # single roofline for the stage; the per-kernel split is in profiles/r02_kin_i8_stats.csv.  "equiv" is the
            # GEMM-form fp64 rate the stage replaces (SURVEY 8d's 2 n^2 p), comparable with the fp64 SYRK's figure.
            setup_info["roofline_kinship"] = {
                "kernel": "kin_i8 stage: i8gemm_packed_kernel_t<false> (G^T G, tiles meeting the upper triangle) + kin_i8_accum_kernel + "
                          "lists of the missing calls (count / scan / fill) + kin_i8_corr2_kernel",
                "bound": "valu + latency (kin_i8_corr2_kernel, ~55 % of the stage: extract, convert, FMA per (missing call, individual) "
                         "pair from a 2-bit copy of the block; the both-missing term as integer LDS atomics over per-SNP lists) / "
                         "mfma int8 (G^T G, ~25 %)",
                "launch_ms_total": round(kin_ms, 3), "blocks": kin_n,
                "equiv_fp64_tflops": setup_info["kinship_gemm_tflops"],
                "equiv_fp64_ratio": round(setup_info["kinship_gemm_tflops"] / 78.6, 4),
                "equiv_note": "GEMM-form fp64 rate the stage replaces (SURVEY 8d: 2 n^2 p) over the fp64 MFMA peak -- a ratio of two "
                              "different arithmetics, NOT a roofline fraction (exact integers; K agrees with the fp64 SYRK to 1e-14)",
                "gtg_int8": "per 20000-SNP block at n = 20000 (profiles/r03_kin_i8_lists_kernel_stats.csv): G^T G 3.66 ms on the 6319 "
                            "of 12403 tiles that meet the upper triangle = 2.26 POP/s = 0.45 of the dense int8 peak; correction 7.9 ms "
                            "(round 2: 7.3 + 17.5 ms); the per-kernel split is not measured inside bench.py"}
        elif kin_ms:
            # K = Xc Xc^T as a SYRK: only the 128 x 128 tiles with tile_n >= tile_m are launched (dgemm_mfma.hip.h), so the
            # flops EXECUTED are tiles * 2 * 128^2 * p; SURVEY 8(d)'s GEMM-form figure 2 n^2 p (what the reference's
            # cblas_dgemm does) is the line above and is what "142" means -- it is not a rate of the matrix pipe
            tn = (n + 127) // 128
            executed = tn * (tn + 1) / 2 * 2.0 * 128 * 128 * args.kin_snps
            setup_info["roofline_kinship"] = {
                "kernel": "dgemm_mfma_glds_kernel, SYRK grid (K = Xc Xc^T)", "bound": "mfma", "unit": "TFLOP/s",
                "achieved": round(executed / (kin_ms * 1e-3) / 1e12, 2), "peak": 78.6,
                "frac": round(executed / (kin_ms * 1e-3) / 1e12 / 78.6, 4), "launch_ms_total": round(kin_ms, 3),
                "launches": kin_n, "flops": "executed (upper-triangle tiles); GEMM-form 2 n^2 p in kinship_gemm_tflops"}
        y += torch.randn(n, dtype=torch.float64, device=dev, generator=gen) * y.std().clamp_min(1e-3)
        api.CenterMatrix(K)
        # the reference's setup stages on the host cores, in a child process beside everything the GPU does from here on
        # (dsyevr at this n "measured once and reported separately", the reference's -gk beside the GPU kinship)
        if world == 1 and args.cpu_setup and args.cpu_sample > 0:
            cpu_setup = start_cpu_setup(args, np, torch, K, n, B, dev)
        shard_eig = (world > 1 and native and not replicated and args.eigen in ("auto", "gemma") and
                     os.environ.get("GEMMA_HIP_EIGH_SHARD", "1") != "0")
        if shard_eig:
            # every rank needs the centred kinship: in the real flow it HAS it (the SNP-sharded kinship ends in an all-reduce,
            # gemma_hip_kin_end_keep); here rank 0 built K alone from synthetic blocks, so K travels once
            t0 = time.time()
            gdist.broadcast_state_native([torch.ones(1, dtype=torch.float64, device=dev)])  # "a collective decomposition follows"
            sent_go = True
            gdist.broadcast_state_native([K])
            setup_info["kinship_bcast_s"] = round(time.time() - t0, 3)
        t0 = time.time()
        eig = args.eigen
        prev_t = os.environ.get("GEMMA_HIP_EIGH_TIMING")
        os.environ["GEMMA_HIP_EIGH_TIMING"] = "1"  # stage seconds for the Amdahl block (gemma_hip_dbg_eigh_last); prints to stderr
        if eig in ("auto", "gemma"):
            try:
                Kc = K.clone()  # the solver destroys its input; the copy is not part of the stage
                # the solver's ~5 n^2 doubles of workspace ahead of the clock (gemma_hip_eigh_reserve: a stage of its own, reported and
                # counted in amdahl.setup_once_s; the file driver issues it while it reads the genotype files)
                torch.cuda.synchronize()
                t0 = time.time()
                try:
                    api.eigh_reserve(n)
                except L.GemmaHipError:
                    pass  # the solve below reports the shortage itself
                torch.cuda.synchronize()
                setup_info["eigen_workspace_reserve_s"] = round(time.time() - t0, 3)
                t0 = time.time()
                if shard_eig:
                    api.EigenDecomp_Zeroed_sharded(Kc, U, ev)
                else:
                    api.EigenDecomp_Zeroed(Kc, U, ev)
                st = os.environ.get("GEMMA_HIP_EIGH_STAGES", "")
                ne = n + (n & 1) if (n >= 192 and os.environ.get("GEMMA_HIP_EIGH_PAD", "") != "0") else n  # odd n is padded
                two = (st == "2" and ne >= 384 or st != "1" and ne >= 8000) and ne % 2 == 0  # eigh.hip.h: eig_two_stage
                eig = "gemma_hip_eigh (%s)" % ("two-stage: dense -> band -> tridiagonal" if two else "one-stage tridiagonalisation")
                if shard_eig:
                    eig += ("; collective over %d ranks: reduction and divide & conquer on every rank, back-transformations shared out by "
                            "eigenvector, slices exchanged once" % world) if two else "; replicated on every rank (one-stage: nothing is exchanged)"
                del Kc
            except L.GemmaHipError:
                if eig == "gemma":
                    raise
                eig = "torch"
        if eig == "torch":
            w, V = torch.linalg.eigh(K)
            U.copy_(V)
            ev.copy_(torch.where(w < 1e-10, torch.zeros_like(w), w))
            eig = "torch.linalg.eigh (setup only; library eigensolver unavailable)"
            del w, V
        torch.cuda.synchronize()
        setup_info["eigen"] = eig
        setup_info["eigen_s"] = round(time.time() - t0, 3)
        t0r = time.time()
        setup_info["eigen_workspace_released_gb"] = round(api.eigh_release() / 1e9, 1)
        torch.cuda.synchronize()
        setup_info["eigen_workspace_release_s"] = round(time.time() - t0r, 3)
        if prev_t is None:
            os.environ.pop("GEMMA_HIP_EIGH_TIMING", None)
        else:
            os.environ["GEMMA_HIP_EIGH_TIMING"] = prev_t
        if eig.startswith("gemma_hip_eigh"):
            import ctypes
            t8 = (ctypes.c_double * 8)()
            L.lib().gemma_hip_dbg_eigh_last(t8)
            if int(t8[6]) == n + (n & 1 if n >= 192 else 0) or int(t8[6]) == n:
                setup_info["eigen_stages_s"] = {"reduction": round(t8[0], 3), "bulge_chase": round(t8[1], 3), "divide_conquer": round(t8[2], 3),
                                                "backtransform_q2": round(t8[3], 3), "backtransform_q1": round(t8[4], 3),
                                                "sort_transpose": round(t8[5], 3)}
        if args.setup_parity:
            # (U, eval) of THIS run, at this n, through the path the library chose: backward error and orthogonality through the
            # library's own fp64 GEMM, in units of n eps (LAPACK's dsyevr test ratios; bars of tests/test_gpu_eigh.py: 30), and
            # the eigenvalues against an independent solver (rocSOLVER behind torch.linalg.eigvalsh) -- src/lapack.cpp:260-291
            try:
                t1 = time.time()
                eps = 2.0 ** -52
                knorm = float(ev.abs().max())
                R = torch.empty_like(K)
                api.fast_dgemm("N", "N", 1.0, K, U, 0.0, R)
                R.sub_(U * ev[None, :])
                sp = {"n": n, "eigh_resid": round(float(torch.linalg.matrix_norm(R)) / (n * eps * knorm), 4)}
                api.fast_dgemm("T", "N", 1.0, U, U, 0.0, R)
                R.diagonal().sub_(1.0)
                sp["eigh_orth"] = round(float(torch.linalg.matrix_norm(R)) / (n * eps), 4)
                del R
                sp["eigh_what"] = ("||K U - U diag(eval)||_F / (n eps ||K||_2) and ||U^T U - I||_F / (n eps) of this run's n x n "
                                   "eigendecomposition (%s), products on the library's fp64 MFMA GEMM" % eig)
                try:
                    if n > 30000:
                        raise RuntimeError("skipped at n > 30000 (rocSOLVER's eigvalsh takes longer than this library's whole decomposition)")
                    wr = torch.linalg.eigvalsh(K)
                    wr = torch.where(wr < 1e-10, torch.zeros_like(wr), wr)
                    sp["eval_vs_rocsolver_max_rel"] = float((torch.sort(ev)[0] - wr).abs().max()) / knorm
                    del wr
                except Exception as e:  # an out-of-memory workspace at large n must not take the bench down
                    sp["eval_vs_rocsolver_error"] = repr(e)[:120]
                torch.cuda.synchronize()
                sp["seconds"] = round(time.time() - t1, 2)
                setup_info["setup_parity"] = sp
            except Exception as e:
                setup_info["setup_parity"] = {"error": repr(e)[:300]}
        del K
        # UtW = U^T 1, Uty = U^T y through the library's GEMM (CalcUtX, src/mathfunc.cpp:504-506)
        ones = torch.ones((n, 1), dtype=torch.float64, device=dev)
        api.fast_dgemm("T", "N", 1.0, U, ones, 0.0, UtW)
        ycol = y.reshape(n, 1).contiguous()
        Utyc = torch.empty((n, 1), dtype=torch.float64, device=dev)
        api.fast_dgemm("T", "N", 1.0, U, ycol, 0.0, Utyc)
        Uty.copy_(Utyc[:, 0])
        torch.cuda.synchronize()
        nm = api.CalcLambdaNull(ev.cpu().numpy(), UtW.cpu().numpy(), Uty.cpu().numpy(), trace_G=float(ev.mean()))
        null[0], null[1] = nm["l_mle_null"], nm["logl_mle_H0"]
        setup_info["null"] = {k: nm[k] for k in ("l_remle_null", "pve")}
        api.profile_read(L.STAGE_UTX_GEMM, reset=True)
        if args.state_file:
            torch.save({"U": U, "ev": ev, "UtW": UtW, "Uty": Uty, "null": null}, args.state_file)
    if replicated:
        torch.manual_seed(args.seed + rank)
        gen = torch.Generator(device=dev).manual_seed(args.seed + 1000 * rank + 1)
    if world > 1 and native and not multi_real and not replicated:
        # do the other ranks take part in a collective decomposition?  (rank 0 decides: a loaded state file or --eigen torch say no)
        if rank == 0:
            if not sent_go:
                gdist.broadcast_state_native([torch.zeros(1, dtype=torch.float64, device=dev)])
        else:
            go = torch.zeros(1, dtype=torch.float64, device=dev)
            gdist.broadcast_state_native([go])
            if float(go[0]) == 1.0:
                shard_eig = True
                Kr = torch.empty((n, n), dtype=torch.float64, device=dev)
                gdist.broadcast_state_native([Kr])
                api.EigenDecomp_Zeroed_sharded(Kr, U, ev)  # the same (U, eval) on every rank: nothing of it is broadcast below
                del Kr
    t0 = time.time()
    # the single broadcast round: ncclBroadcast issued by the library's own RCCL communicator (csrc/comm.hip.h) when every
    # rank could create it, otherwise the same two collectives through torch.distributed (nccl = RCCL as well)
    bpath = "none (1 rank)"
    if multi_real:
        bpath = kept_state["broadcast"]
    elif replicated:
        bpath = "none: no device transport came up -- every rank ran the whole setup itself (see config.comm.error)"
    elif world > 1:
        if native:
            # after a collective decomposition U and eval are everywhere already: only the rotated covariates / phenotype and the
            # null model's two scalars travel
            gdist.broadcast_state_native([UtW, Uty, null] if shard_eig else [U, ev, UtW, Uty, null])
            bpath = "native: ncclBroadcast from libgemma_hip.so's communicator" if os.environ.get("GEMMA_HIP_COMM", "") != "shm" \
                else "native: libgemma_hip.so's communicator over its shm test transport (GEMMA_HIP_COMM=shm)"
            if shard_eig:
                bpath += "; (U, eval) from the collective eigensolver (gemma_hip_eigh_sharded_d), not broadcast"
        else:
            gdist.broadcast_state([U, ev, UtW, Uty, null])
            bpath = "torch.distributed broadcast (%s)" % os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.synchronize()
    if not multi_real:
        setup_info["broadcast_s"] = round(time.time() - t0, 3)
    setup_info["broadcast"] = bpath

    blocks = [synth_block(torch, n, B, gen, dev, miss=args.miss) for _ in range(args.steps + args.warmup)]
    out = torch.empty((B, 8), dtype=torch.float64, device=dev)
    lmm = api.LMM(a_mode=args.a_mode, l_mle_null=float(null[0]), logl_mle_H0=float(null[1]))
    if multi_real:
        lmm.setup_kept(UtW.cpu().numpy(), Uty.cpu().numpy(), plink=True)  # lmm_setup on the kept (U, eval): nothing crosses PCIe
    else:
        lmm.setup(U, ev, UtW, Uty, plink=True)
    torch.cuda.synchronize()
    setup_info["setup_total_s"] = round(time.time() - t_setup, 1)

    # ------------------------------------------------------------------ warmup + timed steps
    api.profile_enable(False)
    piped = bool(args.pipeline) and os.environ.get("GEMMA_HIP_UTX_I8", "1") != "0"

    def step(blk):
        if piped:
            lmm.batch_pipe(blk, L.GENO_PLINK_2BIT, out)
        else:
            lmm.batch(blk, L.GENO_PLINK_2BIT, out=out)
    if piped:
        # setup, not a step: the pipeline's two buffer sets (planes, packed block) are allocated by its first two calls
        for _ in range(2):
            step(blocks[0])
        lmm.pipe_flush()
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(blocks[i])
    if piped:
        lmm.pipe_flush()
    torch.cuda.synchronize()
    api.profile_enable(True)
    for st in range(L.STAGE_UTX_POST + 1):
        api.profile_read(st, reset=True)
    if world > 1:
        gdist.ctl_barrier()  # the contract's barrier: an all-reduce on the control plane (gloo) -- no device transport in or around the timed region
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_own0 = t0
    for i in range(args.steps):
        step(blocks[args.warmup + i])
    if piped:
        lmm.pipe_flush()  # inside the timed region: the last block's combine and per-SNP stage are waited for
    torch.cuda.synchronize()
    own = time.perf_counter() - t_own0  # this rank's own steps, before it waits for the others
    if world > 1:
        gdist.ctl_barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank_s = [own]
    if world > 1:
        # every rank's own clock (one SUM all-reduce of a vector with one slot per rank), then the contract's MAX over ranks of the
        # barrier-to-barrier time
        tv = [0.0] * world
        tv[rank] = own
        per_rank_s = gdist.ctl_allreduce(tv, "sum")
        elapsed = gdist.ctl_allreduce([elapsed], "max")[0]
    gemm_ms, gemm_n = api.profile_read(L.STAGE_UTX_GEMM)
    post_ms, post_n = api.profile_read(L.STAGE_UTX_POST)
    assoc_ms, assoc_n = api.profile_read(L.STAGE_ASSOC)
    ing_ms, ing_n = api.profile_read(L.STAGE_INGEST)
    res = out.cpu().numpy()
    n_nan = int(np.isnan(res[:, 4]).sum())
    i8_path = os.environ.get("GEMMA_HIP_UTX_I8", "1") != "0"
    timed_kernel = api.last_utx_kernel()  # the matrix kernel the timed steps launched, as the library's launch site recorded it
    # the same blocks one at a time on one stream (gemma_hip_lmm_batch_d: what rounds 1-4 timed), outside the timed region: the
    # records kernel on ALL CUs, and a bit-for-bit comparison of the last block's records with the pipelined run's
    one_stream = None
    if piped and world == 1:
        out1 = torch.empty_like(out)
        lmm.batch(blocks[args.warmup], L.GENO_PLINK_2BIT, out=out1)
        torch.cuda.synchronize()
        for st in range(L.STAGE_UTX_POST + 1):
            api.profile_read(st, reset=True)
        k1 = min(args.steps, 6)
        t1 = time.perf_counter()
        for i in range(k1):
            lmm.batch(blocks[args.warmup + args.steps - k1 + i], L.GENO_PLINK_2BIT, out=out1)
        torch.cuda.synchronize()
        el1 = time.perf_counter() - t1
        g1_ms, g1_n = api.profile_read(L.STAGE_UTX_GEMM, reset=True)
        p1_ms, _ = api.profile_read(L.STAGE_UTX_POST, reset=True)
        a1_ms, _ = api.profile_read(L.STAGE_ASSOC, reset=True)
        i1_ms, _ = api.profile_read(L.STAGE_INGEST, reset=True)
        same = bool(np.array_equal(out1.cpu().numpy(), res, equal_nan=True))
        one_stream = {"steps": k1, "ms_per_step": round(el1 / k1 * 1e3, 3), "value": round(B * k1 / el1, 1), "unit": "SNPs/s",
                      "stage_ms_per_step": {"ingest": round(i1_ms / k1, 3), "utx_gemm": round(g1_ms / k1, 3), "utx_post": round(p1_ms / k1, 3),
                                            "assoc": round(a1_ms / k1, 3)},
                      "utx_gemm_avg_launch_ms": round(g1_ms / max(1, g1_n), 3),
                      "last_block_records_equal_the_pipelined_run_bit_for_bit": same}
        del out1

    # the fp64 MFMA GEMM path on the same blocks, outside the contract's timed region (single GPU only)
    fp64_path = None
    if args.fp64_steps < 0:
        args.fp64_steps = min(args.steps, 10)
    if world == 1 and i8_path and args.fp64_steps > 0:
        os.environ["GEMMA_HIP_UTX_I8"] = "0"
        api.reload_env()  # the library reads its switches once per setup, not per launch
        lmm.batch(blocks[0], L.GENO_PLINK_2BIT, out=out)
        torch.cuda.synchronize()
        api.profile_read(L.STAGE_UTX_GEMM, reset=True)
        t1 = time.perf_counter()
        for i in range(args.fp64_steps):
            lmm.batch(blocks[(args.warmup + i) % len(blocks)], L.GENO_PLINK_2BIT, out=out)
        torch.cuda.synchronize()
        el64 = time.perf_counter() - t1
        g64_ms, g64_n = api.profile_read(L.STAGE_UTX_GEMM)
        os.environ["GEMMA_HIP_UTX_I8"] = "1"
        api.reload_env()
        g64_s = g64_ms * 1e-3 / max(1, g64_n)
        tf = 2.0 * B * n * n / g64_s / 1e12
        fp64_path = {"value": round(B * args.fp64_steps / el64, 1), "unit": "SNPs/s", "steps": args.fp64_steps,
                     "ms_per_step": round(el64 / args.fp64_steps * 1e3, 3),
                     "roofline": {"kernel": "dgemm_mfma_glds_kernel (UtX = X*U, fp64)", "bound": "mfma",
                                  "achieved": round(tf, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(g64_s * 1e3, 3)}}

    # the same step at GEMMA's default missingness ceiling (-miss 0.05): ~2.4 over-full groups of four per row, every row goes
    # through i8_surplus_fix_kernel; outside the timed region, single GPU only
    miss_leg = None
    if world == 1 and i8_path and args.miss_leg > 0:
        mb = [synth_block(torch, n, B, gen, dev, miss=args.miss_leg) for _ in range(2)]
        lmm.batch(mb[0], L.GENO_PLINK_2BIT, out=out)
        torch.cuda.synchronize()
        for st in range(L.STAGE_UTX_POST + 1):
            api.profile_read(st, reset=True)
        t1 = time.perf_counter()
        for i in range(2):
            lmm.batch(mb[i], L.GENO_PLINK_2BIT, out=out)
        torch.cuda.synchronize()
        elm = time.perf_counter() - t1
        miss_leg = {"miss": args.miss_leg, "steps": 2, "ms_per_step": round(elm / 2 * 1e3, 3), "value": round(B * 2 / elm, 1), "unit": "SNPs/s",
                    "ratio_to_timed_step": round(elm / 2 / (elapsed / args.steps), 4),
                    "stage_ms_per_step": {"ingest": round(api.profile_read(L.STAGE_INGEST)[0] / 2, 3),
                                          "utx_gemm": round(api.profile_read(L.STAGE_UTX_GEMM)[0] / 2, 3),
                                          "utx_post (digit combine + fp64 fix-up of the dropped calls)": round(api.profile_read(L.STAGE_UTX_POST)[0] / 2, 3),
                                          "assoc": round(api.profile_read(L.STAGE_ASSOC)[0] / 2, 3)},
                    "nan_p_wald": int(np.isnan(out.cpu().numpy()[:, 4]).sum())}
        del mb

    # blocks without a missing call: the mask product is identically zero and the library drops it (device-side choice between the two
    # forms of the records kernel, same bits); outside the timed region, single GPU only.  NOT the headline: `value` keeps 1 % missing.
    complete_leg = None
    if world == 1 and i8_path and args.complete_steps > 0:
        ks = args.complete_steps
        cb = [synth_block(torch, n, B, gen, dev, miss=0.0) for _ in range(2)]
        lmm.batch(cb[0], L.GENO_PLINK_2BIT, out=out)
        flag = api.last_block_missing()
        torch.cuda.synchronize()
        for st in range(L.STAGE_UTX_POST + 1):
            api.profile_read(st, reset=True)
        t1 = time.perf_counter()
        for i in range(ks):
            lmm.batch(cb[i % 2], L.GENO_PLINK_2BIT, out=out)
        torch.cuda.synchronize()
        elc = time.perf_counter() - t1
        got_c = out.cpu().numpy().copy()
        complete_leg = {"input": "the same synthetic PLINK blocks with no missing call (miss = 0)", "steps": ks,
                        "any_missing_flag": flag, "ms_per_step": round(elc / ks * 1e3, 3), "value": round(B * ks / elc, 1), "unit": "SNPs/s",
                        "ratio_to_timed_step": round(elc / ks / (elapsed / args.steps), 4),
                        "stage_ms_per_step": {"ingest": round(api.profile_read(L.STAGE_INGEST)[0] / ks, 3),
                                              "utx_gemm (genotype product alone)": round(api.profile_read(L.STAGE_UTX_GEMM)[0] / ks, 3),
                                              "utx_post": round(api.profile_read(L.STAGE_UTX_POST)[0] / ks, 3),
                                              "assoc": round(api.profile_read(L.STAGE_ASSOC)[0] / ks, 3)}}
        gc_ms = complete_leg["stage_ms_per_step"]["utx_gemm (genotype product alone)"]
        if gc_ms > 0:
            dgc = ctypes_digits(L, n)  # digit products of the genotype operand alone, all on the dense instruction
            top = dgc * 2.0 * B * n * n / (gc_ms * 1e-3) / 1e12
            complete_leg["roofline"] = {"kernel": "i8gemm_sparse2_r16_g_kernel", "bound": "mfma", "achieved": round(top, 1), "peak": INT8_MFMA_PEAK_TOPS,
                                        "unit": "TOP/s", "frac": round(top / INT8_MFMA_PEAK_TOPS, 4), "avg_launch_ms": gc_ms,
                                        "dense_products": dgc}
        # the same block through both products (GEMMA_HIP_I8_COMPLETE=0): must be the same records, bit for bit
        os.environ["GEMMA_HIP_I8_COMPLETE"] = "0"
        api.reload_env()
        lmm.batch(cb[(ks - 1) % 2], L.GENO_PLINK_2BIT, out=out)
        torch.cuda.synchronize()
        complete_leg["records_bit_equal_to_both_products"] = bool(out.cpu().numpy().tobytes() == got_c.tobytes())
        del os.environ["GEMMA_HIP_I8_COMPLETE"]
        api.reload_env()
        del cb

    # fixed-point dosages (BIMBAM mean genotypes, doc/manual.tex:398-404) as fp64 input: int8-digit dosage planes, then the
    # fp64 MFMA GEMM on the same block as the check; outside the timed region, single GPU only
    dosage_path = None
    if world == 1 and i8_path and args.dosage_steps > 0:
        # k / 100 as a text parser produces it: the correctly rounded quotient.  (tensor / python scalar is a multiplication by
        # the reciprocal in torch -- k * 0.01 is not the double "0.37" parses to, and the library's dosage detection is exact.)
        hundred = torch.tensor([100.0], device=dev, dtype=torch.float64)
        Xd = [torch.randint(0, 201, (B, n), device=dev, generator=gen, dtype=torch.int32).to(torch.float64).div_(hundred)
              for _ in range(2)]
        probe = Xd[0][:4].cpu().numpy()
        assert np.array_equal(probe, np.rint(probe * 100.0) / 100.0), "synthetic dosages are not the parsed decimals"
        del probe
        outd = torch.empty((B, 8), dtype=torch.float64, device=dev)
        lmm.batch(Xd[0], L.GENO_F64_SNP_MAJOR, out=outd)
        torch.cuda.synchronize()
        took = api.last_utx_path()
        for st in (L.STAGE_UTX_GEMM, L.STAGE_UTX_POST, L.STAGE_INGEST):
            api.profile_read(st, reset=True)
        t1 = time.perf_counter()
        for i in range(args.dosage_steps):
            lmm.batch(Xd[i % 2], L.GENO_F64_SNP_MAJOR, out=outd)
        torch.cuda.synchronize()
        eld = time.perf_counter() - t1
        dos_kernel = api.last_utx_kernel()  # the library's launch site says which kernel the planes ran on
        gd_ms, gd_n = api.profile_read(L.STAGE_UTX_GEMM)
        pd_ms, _ = api.profile_read(L.STAGE_UTX_POST)
        id_ms, _ = api.profile_read(L.STAGE_INGEST)
        res_i8 = outd.cpu().numpy().copy()
        os.environ["GEMMA_HIP_UTX_DOSAGE_I8"] = "0"
        api.reload_env()
        lmm.batch(Xd[(args.dosage_steps - 1) % 2], L.GENO_F64_SNP_MAJOR, out=outd)
        torch.cuda.synchronize()
        os.environ.pop("GEMMA_HIP_UTX_DOSAGE_I8")
        api.reload_env()
        res_64 = outd.cpu().numpy()
        okm = np.isfinite(res_i8) & np.isfinite(res_64) & (res_64 != 0)
        cols_used = {1: [0, 1, 4, 7], 2: [5, 7], 3: [0, 1, 6], 4: [0, 1, 4, 5, 6, 7], 9: [0, 1, 5, 6, 7]}[args.a_mode]
        worst = max(float(np.max(np.abs(res_i8[:, c][okm[:, c]] - res_64[:, c][okm[:, c]]) / np.abs(res_64[:, c][okm[:, c]])))
                    for c in cols_used)
        dgd = ctypes_digits(L, n)
        gd_s = gd_ms * 1e-3 / max(1, args.dosage_steps)
        dosage_path = {"value": round(B * args.dosage_steps / eld, 1), "unit": "SNPs/s", "steps": args.dosage_steps,
                       "ms_per_step": round(eld / args.dosage_steps * 1e3, 3),
                       "input": "fp64 SNP-major rows, every value k/100 in [0, 2], no missing entry (BIMBAM mean genotypes)",
                       "utx_path": api.UTX_PATHS.get(took, str(took)),
                       "stage_ms_per_step": {"ingest_and_pack": round(id_ms / args.dosage_steps, 3), "utx_gemm": round(gd_ms / args.dosage_steps, 3),
                                             "utx_post": round(pd_ms / args.dosage_steps, 3)},
                       "roofline": {"kernel": "%s (one signed byte plane x %d digits of U, dense int8 MFMA)" % (dos_kernel["name"], dgd),
                                    "kernel_variant": {k: dos_kernel[k] for k in ("variant", "rows", "digits")},
                                    "bound": "mfma", "achieved": round(dgd * 2.0 * B * n * n / gd_s / 1e12, 1), "peak": INT8_MFMA_PEAK_TOPS,
                                    "unit": "TOP/s", "frac": round(dgd * 2.0 * B * n * n / gd_s / 1e12 / INT8_MFMA_PEAK_TOPS, 4),
                                    "launches_per_step": 1, "ms_per_step": round(gd_s * 1e3, 3)},
                       "vs_fp64_gemm_path_max_rel_diff": worst}
        del Xd, outd

    if rank == 0:
        total_snps = B * args.steps * world
        value = total_snps / elapsed
        # the product of a step may be launched in row chunks (gemma_hip.hip: lmm_batch_plink_chunked): price the step's total
        gemm_avg_s = gemm_ms * 1e-3 / max(1, args.steps)
        gemm_launches_per_step = gemm_n / max(1, args.steps)
        assoc_avg_s = assoc_ms * 1e-3 / max(1, args.steps)  # per step (the stage may run once per row chunk)
        if i8_path:
            # D digits of U x {genotype, missing mask}: 2 D int8 products of 2 n^2 ops per SNP (2 n^2 per SNP);
            # D = 7, or 6 from n = 16384 up (csrc/i8gemm.hip.h)
            import ctypes
            dg = ctypes.c_int(7)
            L.lib().gemma_hip_dbg_i8_digits(n, ctypes.byref(dg))
            ops_per_launch = 2.0 * dg.value * 2.0 * B * n * n
            logical = ops_per_launch / gemm_avg_s / 1e12
            # the kernel's name comes from the library's launch site (gemma_hip_dbg_last_utx_kernel), not from the environment
            kv = timed_kernel["variant"]
            sparse = kv in (L.UTX_KERNEL_SPARSE_BYTES, L.UTX_KERNEL_RECORDS_R32, L.UTX_KERNEL_RECORDS_R16)
            kfn = timed_kernel["name"] + {L.UTX_KERNEL_RECORDS_R16: " (records kernel on v_mfma_i32_16x16x64_i8 + v_smfmac_i32_16x16x128_i8)",
                                          L.UTX_KERNEL_RECORDS_R32: " (records kernel on the 32-row matrix instructions)"}.get(kv, "")
            # With the mask product on the 2:4 sparse MFMA (csrc/i8gemm_sparse.hip.h) a pair of K-steps issues 8 dense + 4 sparse
            # matrix instructions instead of 16 dense ones, and a sparse instruction holds the pipe as long as a dense one
            # (profiles/r02_smfmac_i8_rate.txt): the pipe does 12/16 of the dense work.  `achieved` / `frac` price what the pipe
            # executes against the DENSE int8 peak; `logical_top_s` is the rate of the 2 D products as written.
            achieved = logical * (0.75 if sparse else 1.0)
            kname = ("%s (%d int8-digit products = UtX: %d dense genotype products + %d mask products on the 2:4 sparse MFMA; "
                     "%d base-256 digits of U)" % (kfn, 2 * dg.value, dg.value, dg.value, dg.value)) if sparse else \
                    "i8gemm_packed_kernel (%d int8-digit products = UtX; %d base-256 digits of U)" % (2 * dg.value, dg.value)
            roof = {"kernel": kname, "bound": "mfma",
                    "achieved": round(achieved, 1), "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                    "frac": round(achieved / INT8_MFMA_PEAK_TOPS, 4), "logical_top_s": round(logical, 1),
                    "logical_frac_of_dense_peak": round(logical / INT8_MFMA_PEAK_TOPS, 4),
                    "note": "achieved = dense-equivalent work of the matrix pipe (a 2:4 sparse instruction counted as the dense one it "
                            "takes the time of); logical_top_s = the 2 D products as written; the kernel is power-limited: with an all-zero digit "
                            "operand the 32-row kernel takes 38.65 ms where full-range digits take 55.95, the 16-row kernel 40.8 / 49.8 "
                            "(profiles/r04_i8_operand_value_power.txt, r04_i8_g16s_prototype.txt, DESIGN 3.1c)" if sparse else "dense int8 MFMA",
                    "kernel_symbol": timed_kernel["name"], "kernel_variant": {k: timed_kernel[k] for k in ("variant", "rows", "digits", "fuse", "raster")},
                    "cus": (n_cu - int(os.environ.get("GEMMA_HIP_PIPE_CUS", "64") or 0)) if piped else n_cu,
                    "traffic": None, "launches": gemm_n, "launches_per_step": gemm_launches_per_step,
                    "avg_launch_ms": round(gemm_ms / max(1, gemm_n), 3), "ms_per_step": round(gemm_avg_s * 1e3, 3),
                    # the same launch priced as the fp64 product it replaces (2 n^2 flop per SNP)
                    "equiv_fp64_tflops": round(2.0 * B * n * n / gemm_avg_s / 1e12, 1)}
        else:
            flops_per_launch = 2.0 * B * n * n  # 2 n^2 flop per SNP
            achieved = flops_per_launch / gemm_avg_s / 1e12
            roof = {"kernel": "dgemm_mfma_glds_kernel (UtX = X*U)", "bound": "mfma", "achieved": round(achieved, 2),
                    "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
                    "traffic": None, "launches": gemm_n, "avg_launch_ms": round(gemm_avg_s * 1e3, 3)}
        dgd_ = ctypes_digits(L, n) if i8_path else 0
        line = {
            "metric": ("SNPs/s (-lmm 1 Wald) at n=20k on 1/2/4/8 MI355X; U^T x HBM GB/s vs roofline" if not preset else
                       "SNPs/s of the association stage, BASELINE config 4: n=50k, p=500k SNPs SNP-sharded over %d MI355X (-gk + -lmm 1)" % world),
            "value": round(value, 1), "unit": "SNPs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if preset else "weak", "vs_baseline": None,
            "dtype": ("f64 (U^T x as int8-digit MFMA products with exact int32 accumulation; every column of U scaled by its exact maximum and "
                      "rounded to %d base-256 digits = 2^-%d of that maximum%s)" % (dgd_, 8 * dgd_, ", below fp64's own entry rounding" if dgd_ >= 7 else
                                               " -- fp64 keeps 2^-53 per entry; value_strict / strict_leg (7g6m: U^T x at or below the fp64 GEMM's own error) "
                                               "and fp64_gemm_path carry the unrounded operand")) if i8_path else "f64",
            "data": "synthetic",
            "config": {"workload": "%ssynthetic n=%d, -lmm %d, %d SNPs per step per GPU (PLINK 2-bit, "
                                   "%g%% missing, Balding-Nichols Fst 0.05), c=1" % (
                                       ("configs[3] (--config 4: p = %d SNPs over %d rank(s), %d steps each; kinship from all of them, "
                                        "sharded the same way): " % (B * args.steps * world, world, args.steps)) if preset else
                                       "configs[2] (headline): " if (n == 20000 and args.a_mode == 1) else "", n, args.a_mode, B, 100.0 * args.miss),
                       "n": n, "snps_per_step": B, "kinship_snps": args.kin_snps, "parallelism": "snp-shard x%d" % world,
                       "device": name, "cus": n_cu, "utx_path": "int8-digit" if i8_path else "fp64-gemm",
                       "ranks_seen": sum(1 for x in per_rank_s if x > 0),
                       "per_rank": {"seconds": [round(x, 4) for x in per_rank_s],
                                    "value": [round(B * args.steps / x, 1) if x > 0 else None for x in per_rank_s], "unit": "SNPs/s"},
                       "comm": (dict(zip(("rank", "world", "transport"), api.comm_info()),
                                     transport_names={"0": "none", "1": "RCCL (librccl, ncclAllReduce / ncclBroadcast)",
                                                      "2": "shm test transport (GEMMA_HIP_COMM=shm)"},
                                     setup_mode=setup_mode,
                                     setup_transport={"native": "libgemma_hip.so's own communicator (see transport)",
                                                      "torch": "torch.distributed (%s): rank 0 ran the setup, one broadcast round of (U, eval, UtW, Uty, null)"
                                                               % os.environ.get("BENCH_DIST_BACKEND", "cpu:gloo,cuda:nccl"),
                                                      "replicated": "none: every rank ran the whole setup itself"}.get(setup_mode),
                                     control_plane=str(dist.get_backend()),
                                     error=comm_log.get("error"), staged_start=comm_log.get("tried"),
                                     collectives=(api.comm_stats() if setup_mode == "native" else None),
                                     timed_region="communication-free: every rank analyses its own blocks; barrier + clock exchange on the control plane")
                                if world > 1 else None),
                       "setup": setup_info, "nan_p_wald": n_nan},
            "roofline": roof,
            "roofline_assoc": {"kernel": "per-SNP stage: fixed-lambda table + bracket scan + interval series tables (fp64 MFMA) + series-driven "
                                         "Brent/Newton + one streaming pass per SNP (lmm_grid.hip.h, lmm_search.hip.h, lmm_assoc.hip.h)", "bound": "hbm",
                               "achieved": round(8.0 * n * B / assoc_avg_s / 1e9, 2) if assoc_avg_s else None,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(8.0 * n * B / assoc_avg_s / 1e9 / HBM_PEAK_GBS, 5) if assoc_avg_s else None,
                               "avg_launch_ms": round(assoc_avg_s * 1e3, 3)},
            "stage_ms_per_step": {"ingest": round(ing_ms / max(1, args.steps), 3), "utx_gemm": round(gemm_ms / max(1, args.steps), 3),
                                  "utx_post": round(post_ms / max(1, args.steps), 3),
                                  "assoc": round(assoc_ms / max(1, args.steps), 3),
                                  "overlap": ("two blocks in flight (gemma_hip_lmm_batch_pipe_d): ingest + records + utx_gemm of block i + 1 on %d CUs beside "
                                              "utx_post + assoc of block i on the other %s; the stage times overlap and do not add up to ms_per_step; "
                                              "the timed region ends with the flush (the last block's utx_post + assoc are inside it)"
                                              % (n_cu - int(os.environ.get("GEMMA_HIP_PIPE_CUS", "64") or 0), os.environ.get("GEMMA_HIP_PIPE_CUS", "64")))
                                             if piped else
                                             ("utx_post and assoc of row chunk c run on a side stream beside utx_gemm of chunk c + 1 "
                                              "(%d chunks per step): the stage times overlap and do not add up to ms_per_step"
                                              % int(round(gemm_launches_per_step))) if i8_path and gemm_launches_per_step > 1.5 else
                                             "none: one stream, the stage times add up to ms_per_step"},
        }
        # Amdahl statement for the BASELINE config of this n (SNP-sharded run over N GPUs: setup once, then p / N SNPs per
        # rank with no collective): every term measured in THIS run; the N > 1 rows are projections from the per-GPU rate,
        # the driver's SCALE run is the measurement
        p_cfg = {20000: 1000000, 50000: 500000, 5000: 100000, 10000: 500000}.get(n, B * args.steps)
        per_gpu = value / world
        setup_once = sum(float(setup_info.get(k) or 0.0) for k in ("kinship_s", "eigen_workspace_reserve_s", "eigen_workspace_release_s", "eigen_s", "broadcast_s"))
        est = setup_info.get("eigen_stages_s")
        eig_proj = None
        if est and world == 1:
            # the collective decomposition (gemma_hip_eigh_sharded_d): the back-transformations divide by N, the rest does not; the
            # slices of Z^T travel once (n^2 doubles in all, priced at one xGMI link: 7 links x ~153 GB/s per GPU)
            serial = est["reduction"] + est["bulge_chase"] + est["divide_conquer"] + est["sort_transpose"]
            bt = est["backtransform_q2"] + est["backtransform_q1"]
            other = max(0.0, float(setup_info.get("eigen_s") or 0.0) - serial - bt)  # allocation, copies
            xchg = 8.0 * n * n / 153e9
            eig_proj = {str(N): round(serial + other + bt / N + (xchg if N > 1 else 0.0), 3) for N in (1, 2, 4, 8)}
        eig_note = "eigendecomposition runs on one GPU (replicas only for that stage); ranks idle during it"
        if multi_real:
            setup_once = sum(float(setup_info.get(k) or 0.0) for k in ("kinship_s", "allreduce_s", "eigen_workspace_reserve_s", "eigen_workspace_release_s", "eigen_s", "broadcast_s"))
        if shard_eig:
            eig_note = ("eigendecomposition is a collective (gemma_hip_eigh_sharded_d): reduction + divide & conquer on every rank, the "
                        "back-transformations shared out by eigenvector; eigen_s above was measured with %d rank(s)" % world)
        line["amdahl"] = {
            "p_total": p_cfg, "setup_once_s": round(setup_once, 3),
            "setup_terms_s": {k: setup_info.get(k) for k in ("kinship_s", "allreduce_s", "eigen_workspace_reserve_s", "eigen_workspace_release_s", "eigen_s", "broadcast_s")},
            "kinship_note": "kinship_s covers %d SNPs here; over all p SNPs it shards with the SNPs (one ncclAllReduce of n^2 sums)" % args.kin_snps,
            "assoc_s_per_rank": {str(N): round(p_cfg / N / per_gpu, 3) for N in (1, 2, 4, 8)},
            "projected_total_s": {str(N): round(setup_once + p_cfg / N / per_gpu, 3) for N in (1, 2, 4, 8)},
            "serial_fraction_at_8": round(setup_once / (setup_once + p_cfg / 8 / per_gpu), 4) if setup_once else None,
            "eigen_s_collective_projection": eig_proj,
            "serial_fraction_at_8_collective_eigen": (round((setup_once - eig_proj["1"] + eig_proj["8"]) /
                                                            (setup_once - eig_proj["1"] + eig_proj["8"] + p_cfg / 8 / per_gpu), 4)
                                                      if (eig_proj and setup_once) else None),
            "note": eig_note}
        if one_stream:
            # the kernel's own roofline on the whole chip, beside the pipelined step's (192 CUs, the other 64 busy behind it)
            ops1 = 2.0 * ctypes_digits(L, n) * 2.0 * B * n * n
            one_stream["roofline_frac_all_cus"] = round(ops1 * 0.75 / (one_stream["utx_gemm_avg_launch_ms"] * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS, 4)
            line["one_stream_leg"] = one_stream
            line["roofline"]["note_partition"] = ("the timed steps launch this kernel on %d of %d CUs (gemma_hip_lmm_batch_pipe_d); achieved / frac "
                                                  "price it against the WHOLE chip's peak; one_stream_leg.roofline_frac_all_cus is the same kernel on "
                                                  "all CUs, one block at a time" % (line["roofline"]["cus"], n_cu))
        if fp64_path:
            line["fp64_gemm_path"] = fp64_path
        if dosage_path:
            line["dosage_path"] = dosage_path
        if miss_leg:
            line["miss_leg"] = miss_leg
        if complete_leg:
            line["complete_leg"] = complete_leg
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                if pj.get("n") == n and pj.get("batch") == B:
                    # NOT measured in this run: PMC counters need their own rocprofv3 passes (MI355X_MICROARCH.md), so the
                    # per-launch HBM bytes are read from the committed counter summary of the same kernels and shapes
                    line["roofline"]["traffic_source"] = line["roofline_assoc"]["traffic_source"] = \
                        "static profile: profiles/pmc_traffic.json (rocprofv3 --pmc passes, %s)" % pj.get("collected", "round 1")
                    line["roofline_assoc"]["traffic"] = pj.get("assoc_hbm_bytes_per_launch")
                    # counters of ANOTHER kernel are not this kernel's traffic: the file names the kernel its passes instrumented
                    # and the library names the one the timed steps launched -- they must agree
                    counted = str(pj.get("i8gemm_kernel", "")).split("::")[-1].split("(")[0].strip()
                    if not i8_path:
                        line["roofline"]["traffic"] = pj.get("utx_gemm_hbm_bytes_per_launch")
                    elif counted == timed_kernel["name"]:
                        line["roofline"]["traffic"] = pj.get("i8gemm_hbm_bytes_per_launch")
                        line["roofline"]["traffic_kernel"] = pj.get("i8gemm_kernel")
                        if pj.get("i8gemm_note"):
                            line["roofline"]["traffic_note"] = pj["i8gemm_note"]
                    else:
                        line["roofline"]["traffic"] = None
                        line["roofline"]["traffic_source"] = ("refused: profiles/pmc_traffic.json holds the counters of %r, the timed steps "
                                                              "launched %r" % (pj.get("i8gemm_kernel"), timed_kernel["name"]))
                    if fp64_path:
                        fp64_path["roofline"]["traffic"] = pj.get("utx_gemm_hbm_bytes_per_launch")
            except Exception:
                pass
        if cpu_setup is not None:
            setup_cpu = finish_cpu_setup(args, cpu_setup, t_bench0, setup_info)
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args, np, torch, blocks[args.warmup + args.steps - 1], U, ev, UtW, Uty, res, n, B,
                                                null=(float(null[0]), float(null[1])))
        if cpu_setup is not None and "cpu_baseline" in line:
            line["cpu_baseline"]["setup"] = setup_cpu
        if args.setup_parity:
            # one object for the at-size checks of the setup stages: (U, eval) of this run (residual, orthogonality, eigenvalues vs
            # rocSOLVER -- measured in the setup above) and K / eval against the REFERENCE's outputs (the ref_setup child)
            sp = dict(setup_info.pop("setup_parity", {}) or {})
            if cpu_setup is not None and isinstance(setup_cpu, dict):
                sp.update(setup_cpu.pop("setup_parity", {}) or {})
            line["setup_parity"] = sp
        lmm.finish()
        dg_timed = ctypes_digits(L, n) if i8_path else None
        n_strict = args.steps if args.digits7_steps < 0 else args.digits7_steps
        if world == 1 and i8_path and n_strict > 0 and dg_timed != 7:
            # The strict-precision form at equal standing: seven digits for the genotype product, the mask product
            # on the upper six ("7g6m": 13 int8 products = 10 dense equivalents instead of 12 = 9) -- U^T x at or below the fp64 GEMM's own
            # rounding error, rms and maximum.  The SAME blocks as the timed region, the same number of steps, the same barrier-free
            # single-rank clock; a fresh setup (the digits are cut once per setup).
            os.environ["GEMMA_HIP_I8_FORM"] = "7g6m"
            try:
                lmm7 = api.LMM(a_mode=args.a_mode, l_mle_null=float(null[0]), logl_mle_H0=float(null[1]))
                lmm7.setup(U, ev, UtW, Uty, plink=True)
                out7 = torch.empty((B, 8), dtype=torch.float64, device=dev)
                for i in range(max(1, min(args.warmup, 2))):
                    lmm7.batch(blocks[i % len(blocks)], L.GENO_PLINK_2BIT, out=out7)
                torch.cuda.synchronize()
                for st in range(L.STAGE_UTX_POST + 1):
                    api.profile_read(st, reset=True)
                t1 = time.perf_counter()
                for i in range(n_strict):
                    lmm7.batch(blocks[(args.warmup + args.steps - n_strict + i) % len(blocks)], L.GENO_PLINK_2BIT, out=out7)
                torch.cuda.synchronize()
                el7 = time.perf_counter() - t1
                g7_ms, g7_n = api.profile_read(L.STAGE_UTX_GEMM)
                k7 = api.last_utx_kernel()
                r7 = out7.cpu().numpy()  # the last block of the leg is the last timed block: res holds its 6-digit results
                cols_used = {1: [0, 1, 4, 7], 2: [5, 7], 3: [0, 1, 6], 4: [0, 1, 4, 5, 6, 7], 9: [0, 1, 5, 6, 7]}[args.a_mode]
                okm = np.isfinite(r7) & np.isfinite(res) & (r7 != 0)
                g7_s = g7_ms * 1e-3 / max(1, n_strict)
                line["value_strict"] = round(B * n_strict / el7, 1)
                line["strict_leg"] = {
                    "form": "7g6m", "steps": n_strict, "ms_per_step": round(el7 / n_strict * 1e3, 3),
                    "value": round(B * n_strict / el7, 1), "unit": "SNPs/s",
                    "utx_gemm_ms_per_step": round(g7_ms / n_strict, 3), "launches_per_step": g7_n / max(1, n_strict),
                    "roofline_frac": round(10.0 * 2.0 * B * n * n / g7_s / 1e12 / INT8_MFMA_PEAK_TOPS, 4),
                    "kernel": k7["name"] + " (planes {2,1} {4,3} {6,5}) + i8gemm_sparse2_r16_g_kernel (digit 0, genotype product alone)",
                    "what": "GEMMA_HIP_I8_FORM=7g6m: U cut to 7 balanced base-256 digits against each column's exact maximum (rounded at 2^-56 of it); "
                            "genotype product on all seven, mask product on the upper six (its term is sqrt(n / missing calls) smaller): 13 int8 "
                            "products = 10 dense equivalents where the timed region runs 12 = 9.  U^T x then carries no more error than the "
                            "reference's cblas_dgemm on fp64 operands (src/fastblas.cpp:202): at or below the fp64 MFMA GEMM's rms and maximum "
                            "error against long-double products (tests/test_gpu_at_size.py::test_six_digit_rounding_of_U_is_what_the_model_says)",
                    "timed_region_vs_strict_max_rel_diff": max(float(np.max(np.abs(res[:, c][okm[:, c]] - r7[:, c][okm[:, c]]) / np.abs(r7[:, c][okm[:, c]])))
                                                               for c in cols_used),
                    "lambda_max_rel_diff": float(np.nanmax(np.abs(res[:, 2] - r7[:, 2]) / np.maximum(np.abs(r7[:, 2]), 1e-300))) if args.a_mode in (1, 4) else None}
                lmm7.finish()
                del out7
            except Exception as e:
                line["strict_leg"] = {"error": repr(e)[:300]}
            os.environ.pop("GEMMA_HIP_I8_FORM", None)
            api.reload_env()
        if world == 1 and i8_path and args.lowh2_leg > 0 and args.a_mode in (1, 4):
            # The same trait on a kinship measured in other units: eigenvalues times S moves every lambda-hat to lambda-hat / S
            # (only lambda * delta enters the likelihood), so with S = lambda_null / lambda0 the brackets of the whole block sit
            # in the decades below 1e-3 -- where a trait with next to no heritability puts them on a kinship with large eigenvalues.
            l0 = float(setup_info.get("null", {}).get("l_remle_null", 1.0)) if not loaded else 1.0
            S = max(1.0, l0 / args.lowh2_leg)
            # (One random covariate instead of the intercept: U^T 1 lies in the null space of the centred kinship, and with the
            # eigenvalues times 1e4 the REFERENCE's own REML Newton iteration fails on that structure -- NaN in the oracle too.)
            ev2 = (ev * S).contiguous()
            Uty2 = Uty.clone()
            Uty2[torch.argmin(ev)] = 0.0  # the phenotype's mean (its component along the kinship's null vector): no intercept here to absorb it
            UtW2 = torch.randn((n, 1), dtype=torch.float64, device=dev, generator=gen)
            nm2 = api.CalcLambdaNull(ev2.cpu().numpy(), UtW2.cpu().numpy(), Uty2.cpu().numpy(), trace_G=float(ev2.mean()))
            lmm2 = api.LMM(a_mode=args.a_mode, l_mle_null=nm2["l_mle_null"], logl_mle_H0=nm2["logl_mle_H0"])
            lmm2.setup(U, ev2, UtW2, Uty2, plink=True)
            out2 = torch.empty((B, 8), dtype=torch.float64, device=dev)
            legs = {}
            for low in ("1", "0"):  # tables below 1e-3 in Q form (default) / streaming below 1e-3 (round 2)
                os.environ["GEMMA_HIP_CHEB_LOWLAMBDA"] = low
                if low == "0":
                    lmm2.finish()
                    lmm2.setup(U, ev2, UtW2, Uty2, plink=True)
                lmm2.batch(blocks[0], L.GENO_PLINK_2BIT, out=out2)
                torch.cuda.synchronize()
                api.profile_read(L.STAGE_ASSOC, reset=True)
                t1 = time.perf_counter()
                for i in range(2):
                    lmm2.batch(blocks[i % len(blocks)], L.GENO_PLINK_2BIT, out=out2)
                torch.cuda.synchronize()
                el2 = time.perf_counter() - t1
                legs[low] = {"ms_per_step": round(el2 / 2 * 1e3, 3), "assoc_ms_per_step": round(api.profile_read(L.STAGE_ASSOC)[0] / 2, 3),
                             "lambda_remle_median": float(np.nanmedian(out2.cpu().numpy()[:, 2])) if np.isfinite(out2.cpu().numpy()[:, 2]).any() else None,
                             "nan_p_wald": int(np.isnan(out2.cpu().numpy()[:, 4]).sum())}
            os.environ.pop("GEMMA_HIP_CHEB_LOWLAMBDA", None)
            lmm2.finish()
            line["lowh2_leg"] = {"lambda0": args.lowh2_leg, "eigenvalue_scale": S, "l_remle_null": nm2["l_remle_null"], "steps": 2,
                                 "tables_in_Q_form_below_1e-3": legs["1"], "streaming_below_1e-3 (round 2)": legs["0"],
                                 "assoc_ratio_to_timed_region": round(legs["1"]["assoc_ms_per_step"] / max(1e-9, assoc_ms / max(1, args.steps)), 3)}
            del out2
        if world == 1 and args.e2e_snps > 0:
            del blocks, out, U
            torch.cuda.empty_cache()
            line["e2e"] = e2e_files(args, n)
        if world == 1 and args.c4_leg and n == 20000 and args.cpu_sample > 0:
            line["c4_leg"] = c4_leg(args, t_bench0)
        print(json.dumps(line), flush=True)
    else:
        lmm.finish()
    if world > 1:
        hung = gdist._native_poisoned or any((t.get("error") or "").startswith("no answer within") for t in comm_log.get("tried", []))
        gdist.ctl_barrier()  # rank 0's line is out before anybody tears the group down
        if hung or setup_mode == "replicated":
            # a helper thread may still sit inside a collective that never returns (or torch's device backend is in an error state):
            # leave without the tear-down that would wait for it
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        dist.destroy_process_group()


def start_cpu_setup(args, np, torch, Kc, n, B, dev):
    """Lay out the inputs of `bench.py --child ref_setup` while the kinship is at hand: one synthetic .bed batch of B SNPs for the
    reference's PlinkKin (src/gemma_io.cpp:1599-1738) and the centred kinship of this run (its leading --cpu-setup-n block) for
    its EigenDecomp_Zeroed (src/lapack.cpp:260-291, dsyevr).  The child itself runs AFTER every GPU leg (finish_cpu_setup): 64
    spinning BLAS threads beside the timed region cost it 4 % in an early version of this leg."""
    import subprocess
    from oracle import oracle as O
    if O.ref_lib() is None:
        return None
    try:
        d = shm_dir()
        ne = n if args.cpu_setup_n <= 0 else min(n, args.cpu_setup_n)
        np.save(os.path.join(d, "G.npy"), Kc[:ne, :ne].contiguous().cpu().numpy())
        g2 = torch.Generator(device=dev).manual_seed(args.seed + 777)
        st = torch.random.get_rng_state()
        dst = torch.cuda.get_rng_state(dev)
        blk = synth_block(torch, n, B, g2, dev)
        torch.random.set_rng_state(st)  # synth_block draws its Beta variates from the global generators: leave them as they were
        torch.cuda.set_rng_state(dst, dev)
        bed = os.path.join(d, "batch.bed")
        with open(bed, "wb") as f:
            f.write(bytes([0x6C, 0x1B, 0x01]))
            f.write(blk.cpu().numpy().tobytes())
        spec = {"n": n, "G": os.path.join(d, "G.npy"), "bed": bed, "bed_snps": B, "out": os.path.join(d, "setup.json")}
        gpu = {}
        if args.setup_parity:
            # what the child compares its outputs with: this library's kinship of the SAME .bed batch (every entry) ...
            from gemma_amd import api
            from gemma_amd import _lib as L
            Kb = torch.empty((n, n), dtype=torch.float64, device=dev)
            api.kin_begin(n, 1)
            api.kin_add(blk, L.GENO_PLINK_2BIT)
            api.kin_end(Kb)
            torch.cuda.synchronize()
            try:
                np.save(os.path.join(d, "K_gpu.npy"), Kb.cpu().numpy())
                spec["K_gpu"] = os.path.join(d, "K_gpu.npy")
            except OSError as e:  # a small /dev/shm: the comparison is dropped, the timing leg stays
                gpu["kin_error"] = repr(e)[:200]
            del Kb
            # ... and this library's eigenvalues of the SAME leading block, through both reductions
            Gb = Kc[:ne, :ne].contiguous()
            Ub = torch.empty_like(Gb)
            prev = os.environ.get("GEMMA_HIP_EIGH_STAGES")
            for stages in ("1", "2"):
                if stages == "2" and (ne < 384 or ne % 2):
                    continue
                os.environ["GEMMA_HIP_EIGH_STAGES"] = stages
                evb = torch.empty(ne, dtype=torch.float64, device=dev)
                api.EigenDecomp_Zeroed(Gb.clone(), Ub, evb)
                gpu["eval_stages" + stages] = evb.cpu().numpy()
            if prev is None:
                os.environ.pop("GEMMA_HIP_EIGH_STAGES", None)
            else:
                os.environ["GEMMA_HIP_EIGH_STAGES"] = prev
            del Gb, Ub
        del blk
        json.dump(spec, open(os.path.join(d, "spec.json"), "w"))
        env = dict(os.environ)
        env.pop("RANK", None); env.pop("WORLD_SIZE", None)
        return {"dir": d, "out": spec["out"], "spec": os.path.join(d, "spec.json"), "env": env, "n_eig": ne, "gpu": gpu}
    except Exception as e:  # a reported baseline must never take the bench down
        return {"error": repr(e)[:300]}


def finish_cpu_setup(args, cs, t_bench0, setup_info):
    """Wait for the child (bounded), read what it measured, and put the GPU's figures of the same stages beside it."""
    import shutil
    import subprocess
    if "error" in cs:
        return cs
    cs["t0"] = time.time()
    pr = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", "ref_setup", "--child-spec", cs["spec"]],
                          env=cs["env"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    try:
        pr.wait(timeout=args.cpu_setup_budget)
        finished = True
    except Exception:
        pr.kill()
        finished = False
    res = {}
    try:
        res = json.load(open(cs["out"]))
    except Exception:
        pass
    out = {"kind": "reference", "where": "child process after the GPU legs, host cores of this box",
           "threads": res.get("threads"), "never_in_value": True}
    gpu = cs.get("gpu", {})
    parity = {}
    if "plink_kin" in res:
        pk = res["plink_kin"]
        if "kin_max_abs_diff_over_max" in pk:
            parity["kin_max_rel"] = pk["kin_max_abs_diff_over_max"]
            parity["kin_max_rel_entrywise"] = pk["kin_max_rel_entries_above_1e-3_max"]
            parity["kin_what"] = ("this library's kinship (kin_begin / kin_add / kin_end, -gk 1) of one %d-SNP .bed batch at n = %d against the "
                                  "reference's PlinkKin (src/gemma_io.cpp:1599-1738) on the same file, all %d entries: max |dK| / max |K|; "
                                  "entrywise relative error over the entries above 1e-3 max |K|" % (pk["snps"], res.get("eigen", {}).get("n", 0) or args.n, pk["kin_entries_compared"]))
        elif gpu.get("kin_error"):
            parity["kin_error"] = gpu["kin_error"]
        out["plink_kin"] = {"seconds_per_batch": pk["seconds"], "snps": pk["snps"],
                            "what": "the reference's PlinkKin (src/gemma_io.cpp:1599-1738) on one .bed batch: per-SNP decode / impute / "
                                    "centre + cblas_dgemm(Xlarge Xlarge^T)",
                            "gpu_seconds_per_batch": round(setup_info.get("kinship_s", 0.0) / max(1, (args.kin_snps + pk["snps"] - 1) // pk["snps"]), 4)}
    if "eigen" in res:
        ne = cs["n_eig"]
        try:
            import numpy as np
            ev_ref = np.sort(np.load(cs["out"] + ".eval.npy"))
            knorm = float(np.abs(ev_ref).max())
            for k in ("eval_stages1", "eval_stages2"):
                if k in gpu:
                    parity["eval_max_rel" + ("" if k.endswith("1") else "_two_stage")] = float(np.abs(np.sort(gpu[k]) - ev_ref).max() / knorm)
            parity["eval_what"] = ("eigenvalues of the leading %d x %d block of this run's centred kinship: gemma_hip_eigh (one-stage; '_two_stage': "
                                   "GEMMA_HIP_EIGH_STAGES=2) against the reference's EigenDecomp_Zeroed (src/lapack.cpp:260-291, dsyevr): "
                                   "max |d eval| / ||K||_2" % (ne, ne))
        except Exception as e:
            parity["eval_error"] = repr(e)[:200]
        out["eigen"] = {"seconds": res["eigen"]["seconds"], "n": ne,
                        "what": "the reference's EigenDecomp_Zeroed (src/lapack.cpp:260-291 -> dsyevr_) on the leading %d x %d block of this "
                                "run's centred kinship" % (ne, ne),
                        "gpu_seconds_at_bench_n": setup_info.get("eigen_s"), "bench_n": res["eigen"]["n"], "eval_sum": res["eigen"]["eval_sum"]}
        full = os.path.join(ROOT, "profiles", "r03_cpu_setup_baseline.json")
        if ne != res["eigen"]["n"] and os.path.exists(full):
            try:
                fj = json.load(open(full))
                if fj.get("eigen", {}).get("n") == res["eigen"]["n"]:
                    out["eigen"]["measured_once_at_bench_n"] = dict(fj["eigen"], source="profiles/r03_cpu_setup_baseline.json")
            except Exception:
                pass
    elif not finished:
        out["eigen"] = {"unfinished_after_s": round(time.time() - cs["t0"], 1), "n": None,
                        "what": "the reference's EigenDecomp_Zeroed (dsyevr_) did not finish inside --cpu-setup-budget"}
    else:
        out["error"] = (pr.stderr.read().decode(errors="replace")[-300:] if pr.stderr else "no result")
    shutil.rmtree(cs["dir"], ignore_errors=True)
    out["setup_parity"] = parity
    return out


def c4_leg(args, t_bench0):
    """BASELINE config 4 (n = 50 000, SNP-sharded over 8 GPUs) is one rank's piece per GPU: this bench itself at n = 50 000 in a
    child process (the parent's tensors are gone by now), every optional leg off, a 64-SNP sample against the oracle and the
    reference.  Driver-run evidence of the shape the GPU suite only covers at n = 33 000."""
    import subprocess
    if time.time() - t_bench0 > 400.0:
        return {"skipped": "the bench had already run for %.0f s" % (time.time() - t_bench0)}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--individuals", "50000", "--batch", "20000", "--steps", "2",
           "--warmup", "1", "--kin-snps", "40000", "--cpu-sample", "64", "--ref-procs", "1", "--cpu-setup", "0", "--setup-parity", "1",
           "--fp64-steps", "0", "--dosage-steps", "0", "--miss-leg", "0", "--lowh2-leg", "0", "--digits7-steps", "0", "--e2e-snps", "0",
           "--c4-leg", "0", "--seed", str(args.seed + 4)]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": (r.stderr or r.stdout)[-300:], "seconds": round(time.time() - t0, 1)}
        d = json.loads(lines[-1])
        cb = d.get("cpu_baseline", {})
        return {"workload": d["config"]["workload"], "value": d["value"], "unit": "SNPs/s per GPU", "ms_per_step": d["ms_per_step"],
                "steps": d["steps"], "stage_ms_per_step": d["stage_ms_per_step"], "roofline_frac": d["roofline"]["frac"],
                "utx_digits_note": d["dtype"], "setup": {k: d["config"]["setup"].get(k) for k in ("kinship_s", "eigen_workspace_reserve_s", "eigen_workspace_release_s", "eigen_s", "eigen", "eigen_stages_s")},
                "setup_parity": d.get("setup_parity"),
                "parity": {k: cb.get(k) for k in ("kind", "sample", "gpu_vs_reference_max_rel_err", "gpu_vs_reference_lambda",
                                                   "gpu_vs_oracle_max_rel_err", "gpu_vs_oracle_lambda") if k in cb},
                "projected_8gpu_snps_per_s": round(8 * d["value"], 1), "seconds": round(time.time() - t0, 1)}
    except Exception as e:  # a reported leg must never take the bench line down
        return {"error": repr(e)[:300], "seconds": round(time.time() - t0, 1)}


def e2e_files(args, n):
    """End to end from files (SURVEY 8d "end-to-end"): PLINK .bed/.bim/.fam on disk -> first pass -> kinship ->
    eigendecomposition -> -lmm -> .assoc.txt through the C++ host layer (include/gemma_host.hpp over the host-pointer
    entry points of the C ABI; tests/cpp/gemma_file_driver.cpp -inproc), in a child process; the synthetic set is written by
    tests/cpp/io_host_check plinkgen.  Never part of `value`."""
    import shutil
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="gemma_e2e_")
    try:
        inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "gemma_amd")
        gen, drv = os.path.join(tmp, "io_host_check"), os.path.join(tmp, "gemma_file_driver")
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + inc, os.path.join(ROOT, "tests", "cpp", "io_host_check.cpp"),
                               "-lz", "-pthread", "-o", gen])
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-I" + inc, os.path.join(ROOT, "tests", "cpp", "gemma_file_driver.cpp"),
                               "-L" + libdir, "-lgemma_hip", "-Wl,-rpath," + libdir, "-lz", "-pthread", "-o", drv])
        prefix = os.path.join(tmp, "S")
        tg = time.perf_counter()
        subprocess.check_call([gen, "plinkgen", prefix, str(n), str(args.e2e_snps), str(min(64, os.cpu_count() or 8))],
                              stdout=subprocess.DEVNULL)
        gen_s = time.perf_counter() - tg
        t0 = time.perf_counter()
        r = subprocess.run([drv, "-bfile", prefix, "-inproc", "1", "-lmm", str(args.a_mode), "-outdir", tmp, "-o", "e2e"],
                           capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-300:]}
        kv = dict(t.split("=", 1) for t in r.stdout.split() if "=" in t)
        f = lambda k: float(kv[k]) if k in kv else None
        return {"workload": "PLINK files n=%d p=%d -> .assoc.txt, -lmm %d, one process, files in the page cache" % (n, args.e2e_snps, args.a_mode),
                "wall_s": round(wall, 2), "synthetic_set_written_in_s": round(gen_s, 2), "snps": int(kv.get("snps", 0)), "ni_test": int(kv.get("ni_test", 0)),
                "ns_test": int(kv.get("ns_test", 0)),
                "stage_end_s": {k: f(k) for k in ("t_first_pass", "t_kinship", "t_eigen", "t_null", "t_assoc", "t_written")},
                "assoc_snps_per_s": f("assoc_snps_per_s"), "whole_run_snps_per_s": round(int(kv.get("snps", 0)) / wall, 1)}
    except Exception as e:  # opt-in diagnostics must not take the bench line down
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(args, np, torch, block, U, ev, UtW, Uty, gpu_res, n, B, null=(0.0, 0.0)):
    """The CPU path on a bounded sample of the last timed block, on this box's host cores.
    kind "reference": the reference's own LMM::Analyze (src/lmm.cpp:1474-1658: Xlarge batching, mean imputation,
    fast_dgemm(U^T X) on OpenBLAS, the serial per-SNP loop), called in-process from oracle/_ref/libgemma_ref.so -- the
    reference's sources compiled unchanged (oracle/Makefile `ref`) -- when that library travelled with the repo;
    kind "port": the oracle (numpy/OpenBLAS dgemm + the serial C restatement of the same loop) otherwise.  The oracle leg
    always runs: it is also the parity check of the timed GPU block."""
    from oracle import oracle as O
    S = min(args.cpu_sample, B)
    raw = block[:S].cpu().numpy()
    X = O.bed_decode(raw, n)
    Uh, evh = U.cpu().numpy(), ev.cpu().numpy()
    UtWh, Utyh = UtW.cpu().numpy(), Uty.cpu().numpy()
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    Xi = O.impute_mean(X)
    UtX = np.ascontiguousarray(Xi @ Uh)
    t1 = time.perf_counter()
    ref = O.lmm_batch_UtX(args.a_mode, evh, UtWh, Utyh, UtX, l_mle_null=null[0], logl_mle_H0=null[1], plink_nan_rule=1)
    t2 = time.perf_counter()

    cols = {"beta": 0, "se": 1, "lambda_remle": 2, "lambda_mle": 3, "p_wald": 4, "p_lrt": 5, "p_score": 6, "logl_H1": 7}
    used = {1: ["beta", "se", "logl_H1", "p_wald"], 2: ["logl_H1", "p_lrt"], 3: ["beta", "se", "p_score"],
            4: ["beta", "se", "logl_H1", "p_wald", "p_lrt", "p_score"], 9: ["beta", "se", "logl_H1", "p_lrt", "p_score"]}[args.a_mode]
    lams = {1: ["lambda_remle"], 2: ["lambda_mle"], 3: [], 4: ["lambda_remle", "lambda_mle"], 9: ["lambda_mle"]}[args.a_mode]

    def rel_err(r, k):
        g = gpu_res[:len(r[k]), cols[k]]
        ok = np.isfinite(r[k]) & np.isfinite(g)
        return np.abs(g[ok] - r[k][ok]) / np.maximum(np.abs(r[k][ok]), 1e-300)

    def worst_err(r):
        """beta / se / logl / p-values: the worst relative error over the sample (bar: 1e-6)."""
        return max(float(rel_err(r, k).max()) for k in used)

    def lambda_err(r):
        """lambda-hat: the reference reports the Newton iterate before the one that met its stopping rule
        (src/lmm.cpp:2071-2073,2096), so a flipped trip count moves it by the size of the penultimate step: reported as
        the fraction of the sample within 1e-6 and the worst case (bar: >= 98 % within 1e-6, all within 1e-3)."""
        out = {}
        for k in lams:
            e = rel_err(r, k)
            out[k] = {"frac_within_1e-6": round(float(np.mean(e <= 1e-6)), 5), "max_rel_err": float(e.max()), "n": int(e.size)}
            # second tier (tests/test_gpu_parity.py _classify_lambda): every value outside 1e-6 must be a flipped Brent / Newton trip
            # count -- the reference's own stopping rule holds at the GPU's value, or the likelihood there equals the reference's
            g = gpu_res[:len(r[k]), cols[k]]
            okm = np.isfinite(r[k]) & np.isfinite(g)
            with np.errstate(divide="ignore", invalid="ignore"):
                outl = np.flatnonzero(okm & (np.abs(g - r[k]) > 1e-6 * np.abs(r[k])))
            if len(outl):
                fn = "R" if k == "lambda_remle" else "L"
                sg, lg = O.newton_step_rel(fn, evh, UtWh, Utyh, UtX[outl], g[outl])
                _, lr = O.newton_step_rel(fn, evh, UtWh, Utyh, UtX[outl], r[k][outl])
                flipped = (sg < 1e-5) | (np.abs(lg - lr) <= 2e-12 * np.abs(lr))
                out[k].update({"outside_1e-6": int(len(outl)), "flipped_trip_counts": int(flipped.sum()), "wrong": int((~flipped).sum())})
            else:
                out[k].update({"outside_1e-6": 0, "flipped_trip_counts": 0, "wrong": 0})
        g_nan = ~np.isfinite(gpu_res[:len(r["logl_H1"]), 7])
        out["failed_search_flips"] = int((g_nan != ~np.isfinite(r["logl_H1"])).sum()) if args.a_mode != 3 else 0
        return out

    # the reference amortises the GEMM over 20000-SNP batches: report the GEMM leg's flop rate beside it
    gemm_rate = 2.0 * n * n * S / (t1 - t0)
    port = {"value": round(S / (t2 - t0), 2), "unit": "SNPs/s", "cores": cores, "kind": "port",
            "sample": "%d SNPs of the last timed block: numpy/OpenBLAS dgemm on %d threads (%.1f GFLOP/s) %.2f s + "
                      "serial per-SNP loop on 1 thread %.2f s" % (S, cores, gemm_rate / 1e9, t1 - t0, t2 - t1),
            "host_threads": cores, "gpu_vs_oracle_max_rel_err": worst_err(ref), "gpu_vs_oracle_lambda": lambda_err(ref)}
    if O.ref_lib() is None:
        return port
    # the reference's loop costs ~45-60 ms per SNP at n = 20 000 (13x the oracle's: heap allocations and strided Uab columns in
    # every likelihood evaluation), so ONE call is cut to ~15-20 s; --ref-procs independent processes, each the reference's own
    # LMM::Analyze on its own slice, run side by side (the host has the cores) so that the parity sample is >= 2000 SNPs
    S_port = S
    S1 = min(S, max(64, int(320 * (20000.0 / n) ** 2)))
    P = max(1, min(args.ref_procs, S // S1))
    threads = O.ref_blas_threads()
    try:
        if P == 1:
            t3 = time.perf_counter()
            rr = O.ref_lmm_analyze(args.a_mode, Uh, evh, UtWh, Utyh, X[:S1], l_mle_null=null[0], logl_mle_H0=null[1])
            wall = time.perf_counter() - t3
            per_proc = [wall]
        else:
            import shutil
            import subprocess
            d = shm_dir()
            try:
                for k, v in (("U", Uh), ("ev", evh), ("UtW", UtWh), ("Uty", Utyh)):
                    np.save(os.path.join(d, k + ".npy"), v)
                procs = []
                env = dict(os.environ)
                # few BLAS threads per process: the loop is serial, its per-SNP vector calls wake the whole pool each time, and 8 x 32
                # spinning threads on a 256-thread host made one process 20 x slower than alone (277 s for 320 SNPs)
                env["OPENBLAS_NUM_THREADS"] = str(max(1, min(threads, 8, cores // (2 * P))))
                t3 = time.perf_counter()
                for i in range(P):
                    np.save(os.path.join(d, "X%d.npy" % i), X[i * S1:(i + 1) * S1])
                    spec = {"U": os.path.join(d, "U.npy"), "ev": os.path.join(d, "ev.npy"), "UtW": os.path.join(d, "UtW.npy"),
                            "Uty": os.path.join(d, "Uty.npy"), "X": os.path.join(d, "X%d.npy" % i), "a_mode": args.a_mode,
                            "l_mle_null": null[0], "logl_mle_H0": null[1], "out": os.path.join(d, "out%d.npy" % i)}
                    json.dump(spec, open(os.path.join(d, "spec%d.json" % i), "w"))
                    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", "ref_lmm", "--child-spec",
                                                   os.path.join(d, "spec%d.json" % i)], env=env, stdout=subprocess.DEVNULL,
                                                  stderr=subprocess.PIPE))
                try:
                    for pr in procs:
                        if pr.wait(timeout=max(5.0, 150.0 - (time.perf_counter() - t3))) != 0:
                            raise RuntimeError("reference child failed: " + pr.stderr.read().decode(errors="replace")[-200:])
                finally:
                    for pr in procs:
                        if pr.poll() is None:
                            pr.kill()
                wall = time.perf_counter() - t3
                outs = [np.load(os.path.join(d, "out%d.npy" % i)) for i in range(P)]
                per_proc = [json.load(open(os.path.join(d, "out%d.npy.json" % i)))["seconds"] for i in range(P)]
                threads = int(env["OPENBLAS_NUM_THREADS"])
            finally:
                shutil.rmtree(d, ignore_errors=True)
            rr = np.zeros(P * S1, dtype=[(k, "<f8") for k in ("beta", "se", "lambda_remle", "lambda_mle", "p_wald", "p_lrt", "p_score", "logl_H1")])
            allo = np.concatenate(outs, axis=0)
            for j, k in enumerate(rr.dtype.names):
                rr[k] = allo[:, j]
    except Exception as e:  # the checker must never take the bench down
        port["reference_error"] = repr(e)[:300]
        return port
    one = float(np.median(per_proc))
    # value = the WHOLE-HOST rate of this leg (all processes side by side: what a reader expects under `value`); the rate
    # of one process beside it
    return {"value": round(P * S1 / wall, 2), "unit": "SNPs/s", "cores": threads * P, "kind": "reference",
            "value_one_process": round(S1 / one, 2), "threads_per_process": threads,
            "sample": "%d SNPs of the last timed block through the reference's own LMM::Analyze (oracle/_ref/libgemma_ref.so = "
                      "/root/reference/src compiled unchanged, GSL API from oracle/gslshim): %d independent processes x %d SNPs side by "
                      "side, %.2f s wall, %.2f s median per process; value = all %d processes together (each: OpenBLAS dgemm on %d threads + "
                      "its serial per-SNP loop, incl. its 2 x n x 20000 batch buffers); value_one_process = one of them" % (P * S1, P, S1, wall, one, P, threads),
            "processes": P, "aggregate_snps_per_s": round(P * S1 / wall, 2), "host_threads": cores,
            "threads_note": "the reference's OpenBLAS build caps its pool at %d threads; the box has %d" % (O.ref_blas_threads(), cores),
            "gpu_vs_reference_max_rel_err": worst_err(rr), "gpu_vs_reference_lambda": lambda_err(rr),
            "port": port, "port_sample_snps": S_port}


if __name__ == "__main__":
    main()
