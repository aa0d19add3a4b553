"""Two ranks of the HIP library (not the oracle) on ONE device: the N > 1 protocol of gemma_amd.dist with the real
per-shard compute -- rank 0 holds (U, eval, UtW, Uty), one broadcast round, every rank runs libgemma_hip.so on its
contiguous SNP range (device-pointer entry points), SUMSTAT gathered in rank order -- must equal the single-rank run
BIT FOR BIT, including AnalyzePlink's carry of the previous SNP's beta / se over a failed lambda search at a shard
boundary (src/lmm.cpp:1725,1870-1884).  gloo carries the collectives here because RCCL refuses two ranks on one
device; on the multi-GPU node the same code runs over nccl = RCCL (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case_path, outdir, batch):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from gemma_amd import api
    from gemma_amd import _lib as L
    from gemma_amd import dist as gdist

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api.init(0, verbose=0)
    d = np.load(case_path)
    n = d["U"].shape[0]
    if rank == 0:
        U, ev = torch.from_numpy(d["U"]).to(dev), torch.from_numpy(d["ev"]).to(dev)
        UtW, Uty = torch.from_numpy(d["UtW"]).to(dev), torch.from_numpy(d["Uty"]).to(dev)
    else:
        U, ev = torch.zeros((n, n), dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev)
        UtW = torch.zeros((n, 1), dtype=torch.float64, device=dev)
        Uty = torch.zeros(n, dtype=torch.float64, device=dev)
    gdist.broadcast_state([U, ev, UtW, Uty], small_limit=4 * n)  # U on its own, the three vectors coalesced
    assert torch.equal(U.cpu(), torch.from_numpy(d["U"]))
    raw = torch.from_numpy(d["raw"]).to(dev)
    p_total = raw.shape[0]
    lo, hi = gdist.shard_range(p_total, rank, world)
    lmm = api.LMM(a_mode=1)
    lmm.setup(U, ev, UtW, Uty, plink=True)
    lmm.set_indicator(d["ind"])
    gdist.seed_plink_carry(lambda j: float(lmm.batch(raw[j:j + 1], L.GENO_PLINK_2BIT)[0, 7]), lo)
    outs = [lmm.batch(raw[s0:min(hi, s0 + batch)], L.GENO_PLINK_2BIT) for s0 in range(lo, hi, batch)]
    local = torch.cat(outs) if outs else torch.zeros((0, 8), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    lmm.finish()
    full = gdist.gather_sumstat(local, p_total)
    if rank == 0:
        np.save(os.path.join(outdir, "gathered.npy"), full.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [97, 4096])
def test_two_hip_ranks_on_one_device_equal_single_rank(gpu_api, oracle, tmp_path, batch):
    import torch.multiprocessing as mp
    from gemma_amd import _lib as L
    from test_gpu_parity import _plink_case
    rng = np.random.default_rng(2024)
    ni_total, p_total = 700, 601
    ind, raw = _plink_case(oracle, rng, ni_total, p_total)
    n = int(ind.sum())
    # SNPs whose lambda search fails (every call missing: the imputed mean is 0/0, the likelihoods are NaN,
    # src/lmm.cpp:1819-1827): the first SNP of the file, the first two of rank 1's shard, the one just before it and
    # one in the middle of a shard
    lo1 = (p_total + 1) // 2
    dead = [0, lo1 - 1, lo1, lo1 + 1, 450]
    for s in dead:
        raw[s, :] = 0x55  # code 01 = missing for every individual
    Kg = np.delete(oracle.bed_decode(raw, ni_total)[:, ind == 1], dead, axis=0)
    U, ev, _ = oracle.eigen_decomp_zeroed(oracle.center_matrix(oracle.calc_kin(Kg, 1)))
    y = rng.standard_normal(n)
    UtW, Uty = U.T @ np.ones((n, 1)), U.T @ y
    case = tmp_path / "case.npz"
    np.savez(case, U=U, ev=ev, UtW=UtW, Uty=Uty, raw=raw, ind=ind)
    # single rank, same batching
    lmm = gpu_api.LMM(a_mode=1)
    lmm.setup(U, ev, UtW, Uty, plink=True)
    lmm.set_indicator(ind)
    single = np.concatenate([lmm.batch(raw[s0:s0 + batch], L.GENO_PLINK_2BIT) for s0 in range(0, p_total, batch)])
    lmm.finish()
    single = single.view(np.float64).reshape(-1, 8)
    failed = np.isnan(single[:, 7])
    assert failed[dead].all() and failed.sum() < 12
    assert single[0, 0] == 0.0 and single[lo1, 0] == single[lo1 - 2, 0] != 0.0  # the carry is what is being tested
    mp.spawn(_worker, args=(2, _free_port(), str(case), str(tmp_path), batch), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    assert got.shape == single.shape
    assert np.array_equal(got, single, equal_nan=True)


def _eigh_worker(rank, world, port, case_path, outdir, env):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["GEMMA_HIP_COMM"] = "shm"  # RCCL refuses two ranks on one device: the library's shm test transport
    os.environ.update(env)
    import torch
    import torch.distributed as dist
    from gemma_amd import api
    from gemma_amd import dist as gdist

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api.init(0, verbose=0)
    assert gdist.native_comm_init(), "the library's communicator (shm transport) did not come up"
    # round 6: the staged start -- ONE KiB through the all-reduce and the broadcast (from rank 0 and from the last rank), every value
    # checked on every rank, before anything n^2 goes through the communicator
    assert gdist.native_comm_selftest(timeout=60.0), gdist.native_comm_error()
    st0 = api.comm_stats()
    assert st0["allreduce_calls"] == 1 and st0["bcast_calls"] == 2 and st0["allreduce_bytes"] == 1024.0 and st0["bcast_bytes"] == 2048.0
    K = torch.from_numpy(np.load(case_path)).to(dev)
    n = K.shape[0]
    U = torch.empty_like(K)
    ev = torch.empty(n, dtype=torch.float64, device=dev)
    tr = api.EigenDecomp_Zeroed_sharded(K.clone(), U, ev)
    torch.cuda.synchronize()
    st1 = api.comm_stats()  # the collective solve's agreements, and -- when the back-transformations were shared out -- one slice per rank
    if os.environ.get("GEMMA_HIP_EIGH_SHARD", "1") != "0":  # (=0: replicas, nothing is exchanged)
        assert st1["allreduce_calls"] + st1["bcast_calls"] > st0["allreduce_calls"] + st0["bcast_calls"]
    assert st1["bcast_pieces"] >= st1["bcast_calls"] and st1["allreduce_pieces"] >= st1["allreduce_calls"]
    np.savez(os.path.join(outdir, "eig_rank%d.npz" % rank), U=U.cpu().numpy(), ev=ev.cpu().numpy(), tr=tr)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world,env", [(1538, 2, {}), (1090, 3, {}), (1538, 2, {"GEMMA_HIP_EIGH_SHARD_FORCE_DIFFER": "1"}),
                                         (1538, 2, {"GEMMA_HIP_EIGH_SHARD": "0"})])
def test_sharded_backtransformation_equals_single_rank(gpu_api, tmp_path, monkeypatch, n, world, env):
    """gemma_hip_eigh_sharded_d -- every rank reduces and runs the divide & conquer on its own copy, the two
    back-transformations are shared out by eigenvector (rows of Z^T, whole 64-row blocks), the slices are exchanged once.  Two
    / three ranks on ONE device over the library's shm test transport (the two-stage path forced at this small n): every
    rank's (U, eval) must equal the single-rank result BIT FOR BIT -- also when the ranks' agreement check is made to fail
    (rank 0 then finishes alone and broadcasts) and with the sharing switched off (replicas)."""
    import torch.multiprocessing as mp
    from test_gpu_eigh import _sym
    monkeypatch.setenv("GEMMA_HIP_EIGH_STAGES", "2")
    A = _sym(n, 77 + n, "kinship")
    case = tmp_path / "K.npy"
    np.save(case, A)
    U1, w1 = np.zeros((n, n)), np.zeros(n)
    tr1 = gpu_api.EigenDecomp_Zeroed(A.copy(), U1, w1)
    mp.spawn(_eigh_worker, args=(world, _free_port(), str(case), str(tmp_path), dict(env, GEMMA_HIP_EIGH_STAGES="2")),
             nprocs=world, join=True)
    for r in range(world):
        d = np.load(tmp_path / ("eig_rank%d.npz" % r))
        assert np.array_equal(d["ev"], w1), "rank %d eigenvalues" % r
        assert np.array_equal(d["U"], U1), "rank %d eigenvectors differ from the single-rank solve" % r
        assert float(d["tr"]) == tr1
    resid = np.linalg.norm(A @ U1 - U1 * w1[None, :]) / (np.linalg.norm(A, 2) * n * np.finfo(float).eps)
    assert resid < 30


def _eigh_fail_worker(rank, world, port, case_path, outdir, env):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["GEMMA_HIP_COMM"] = "shm"
    os.environ.update(env)
    import torch
    import torch.distributed as dist
    from gemma_amd import api
    from gemma_amd import dist as gdist
    from gemma_amd._lib import GemmaHipError

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api.init(0, verbose=0)
    assert gdist.native_comm_init()
    K = torch.from_numpy(np.load(case_path)).to(dev)
    n = K.shape[0]
    U = torch.empty_like(K)
    ev = torch.empty(n, dtype=torch.float64, device=dev)
    verdict = "returned a result"
    try:
        api.EigenDecomp_Zeroed_sharded(K.clone(), U, ev)
    except GemmaHipError as e:
        verdict = "error: %s" % e
    with open(os.path.join(outdir, "fail_rank%d.txt" % rank), "w") as f:
        f.write(verdict)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("env", [{"GEMMA_HIP_EIGH_FAIL_RANK": "1"},
                                 {"GEMMA_HIP_EIGH_SHARD_FORCE_DIFFER": "1", "GEMMA_HIP_EIGH_FAIL_FALLBACK": "1"}])
def test_collective_eigensolver_ranks_agree_on_a_failure_outside_the_core(gpu_api, tmp_path, monkeypatch, env):
    """(1) a rank that fails BEFORE the collective solver (its own allocations) and (2) a rank 0 that fails inside the
    root-only fall-back used to leave the other ranks waiting in a collective for ever.  Now every rank takes part in the solver's
    agreement and ALL of them return an error -- the spawn below would time out otherwise."""
    import torch.multiprocessing as mp
    from test_gpu_eigh import _sym
    n = 1538
    A = _sym(n, 77 + n, "kinship")
    case = tmp_path / "K.npy"
    np.save(case, A)
    ctx = mp.spawn(_eigh_fail_worker, args=(2, _free_port(), str(case), str(tmp_path), dict(env, GEMMA_HIP_EIGH_STAGES="2")),
                   nprocs=2, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > 150:
            for p in ctx.processes:
                p.kill()
            pytest.fail("the ranks of a collective solve with one failing rank did not return: they wait for each other")
    for r in range(2):
        v = (tmp_path / ("fail_rank%d.txt" % r)).read_text()
        assert v.startswith("error:"), "rank %d %s" % (r, v)
