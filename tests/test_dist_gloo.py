"""The N > 1 protocol on CPU: world_size 2, gloo.  rank 0 owns (U, eval, UtW, Uty); ONE broadcast round;
each rank analyses its contiguous SNP range; SUMSTAT blocks come back in SNP order.  The product has
no CPU compute path, so the oracle stands in for the per-shard compute here -- what is under test is
gemma_amd.dist (shard_range / broadcast_state / gather_sumstat) and the claim that sharding does not
change any SNP's result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _analyze(O, U, ev, UtW, Uty, X, l_mle_null, logl_mle_H0):
    """Per-shard compute stand-in.  UtX is formed one SNP at a time (dgemv) so that the stand-in, like
    the HIP GEMM, gives a SNP the same bits whatever block it arrives in (OpenBLAS dgemm does not)."""
    Xi = O.impute_mean(X)
    UtX = np.stack([x @ U for x in Xi]) if len(Xi) else np.zeros((0, U.shape[0]))
    return O.lmm_batch_UtX(4, ev, UtW, Uty, UtX, l_mle_null=l_mle_null, logl_mle_H0=logl_mle_H0)


def _worker(rank, world, port, p_total, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gemma_amd import dist as gdist
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(os.path.join(ROOT, "tests", "golden", "bxd.npz"))
    n = d["U"].shape[0]
    if rank == 0:
        U, ev = torch.from_numpy(d["U"].copy()), torch.from_numpy(d["eval"].copy())
        UtW, Uty = torch.from_numpy(d["UtW"].copy()), torch.from_numpy(d["Uty"].copy())
        null = torch.from_numpy(d["null"][:2].copy())
    else:
        U, ev = torch.zeros((n, n), dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
        UtW, Uty = torch.zeros((n, 3), dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
        null = torch.zeros(2, dtype=torch.float64)
    gdist.broadcast_state([U, ev, UtW, Uty, null])
    assert torch.equal(U, torch.from_numpy(d["U"]))  # every rank now holds rank 0's state bit for bit
    lo, hi = gdist.shard_range(p_total, rank, world)
    X = d["X"].astype(np.float64)[:p_total][lo:hi]
    st = _analyze(O, U.numpy(), ev.numpy(), UtW.numpy(), Uty.numpy(), X, float(null[0]), float(null[1]))
    local = torch.from_numpy(st.view(np.float64).reshape(-1, 8).copy())
    full = gdist.gather_sumstat(local, p_total)
    if rank == 0:
        np.save(os.path.join(outdir, "gathered.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("p_total", [301, 64])
def test_two_rank_sharding_equals_single_rank(tmp_path, p_total, oracle, bxd):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, p_total, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    null = bxd["null"]
    ref = _analyze(oracle, bxd["U"], bxd["eval"], bxd["UtW"], bxd["Uty"], bxd["X"].astype(np.float64)[:p_total],
                   null[0], null[1])
    ref = ref.view(np.float64).reshape(-1, 8)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref, equal_nan=True)  # bit-identical per SNP, in SNP order


def _kin_worker(rank, world, port, p_total, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gemma_amd import dist as gdist
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(os.path.join(ROOT, "tests", "golden", "bxd.npz"))
    X = d["X"].astype(np.float64)[:p_total]
    lo, hi = gdist.shard_range(p_total, rank, world)
    K_r = torch.from_numpy(O.calc_kin(X[lo:hi], 1))  # stand-in for kin_begin / kin_add / kin_end on this rank's SNPs
    K, ns = gdist.allreduce_kinship(K_r, hi - lo)
    assert ns == p_total
    np.save(os.path.join(outdir, "K_rank%d.npy" % rank), K.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_kinship_allreduce(tmp_path, oracle, bxd):
    """SNP-sharded kinship: partial X_r X_r^T on each rank, ONE all-reduce, every rank ends with X X^T / ns."""
    import torch.multiprocessing as mp
    p_total = 1001  # uneven shards: 501 + 500
    port = _free_port()
    mp.spawn(_kin_worker, args=(2, port, p_total, str(tmp_path)), nprocs=2, join=True)
    ref = oracle.calc_kin(bxd["X"].astype(np.float64)[:p_total], 1)
    K0, K1 = np.load(tmp_path / "K_rank0.npy"), np.load(tmp_path / "K_rank1.npy")
    assert np.array_equal(K0, K1)
    np.testing.assert_allclose(K0, ref, rtol=1e-12, atol=1e-14)


def test_seed_plink_carry_reproduces_the_unsharded_chain():
    """AnalyzePlink's beta / se carry over failed lambda searches (src/lmm.cpp:1725,1870-1884) at a shard boundary:
    gemma_amd.dist.seed_plink_carry walks back from the shard's first SNP until a search succeeds; a model of the
    library's carry state (a success overwrites it, a failure copies it) must then give the shard the rows of the
    unsharded run for every failure pattern, including 'everything before the shard failed'."""
    from gemma_amd import dist as gdist
    rng = np.random.default_rng(5)
    for trial in range(200):
        p = int(rng.integers(2, 40))
        failed = rng.random(p) < (0.6 if trial % 3 == 0 else 0.15)
        beta = rng.standard_normal(p)

        def run(lo, hi, carry):
            rows = []
            for s in range(lo, hi):
                if not failed[s]:
                    carry = beta[s]
                rows.append(carry)
            return rows, carry

        full, _ = run(0, p, 0.0)
        for world in (2, 3, 5):
            got = []
            for r in range(world):
                lo, hi = gdist.shard_range(p, r, world)
                state = {"carry": 0.0}

                def analyse_one(j):
                    _, state["carry"] = run(j, j + 1, state["carry"])
                    return float("nan") if failed[j] else 1.0

                steps = gdist.seed_plink_carry(analyse_one, lo)
                assert steps <= lo
                rows, _ = run(lo, hi, state["carry"])
                got += rows
            assert got == full


class _FakeCommLib:
    """Stands in for libgemma_hip.so's communicator entry points (no GPU here): comm_init fails on `bad_rank`."""

    def __init__(self, rank, bad_rank):
        self.rank, self.bad_rank = rank, bad_rank
        self.inits = self.finalizes = 0

    def gemma_hip_comm_unique_id(self, buf):
        return 0

    def gemma_hip_comm_init(self, ident, rank, world):
        self.inits += 1
        return 4 if rank == self.bad_rank else 0

    def gemma_hip_comm_finalize(self):
        self.finalizes += 1
        return 0


def _native_init_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import time

    import torch
    import torch.distributed as dist
    from gemma_amd import _lib as L
    from gemma_amd import dist as gdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fake = _FakeCommLib(rank, bad_rank=1)
    L.lib = lambda: fake
    L.check = lambda rc, what="": None
    torch.cuda.current_device = lambda: 0
    torch.cuda.set_device = lambda d: None
    first = gdist.native_comm_init(timeout=20.0)
    # the second call must return at once on EVERY rank, without a collective (rank 1 alone calls it a third time:
    # a collective inside would hang it, the join below would time out)
    t0 = time.time()
    second = gdist.native_comm_init(timeout=20.0)
    third = gdist.native_comm_init(timeout=20.0) if rank == 1 else False
    dt = time.time() - t0
    with open(os.path.join(outdir, "r%d.txt" % rank), "w") as f:
        f.write("%d %d %d %d %d %.3f" % (first, second, third, fake.inits, fake.finalizes, dt))
    dist.barrier()
    dist.destroy_process_group()


def test_native_comm_init_failure_is_agreed_and_remembered(tmp_path):
    """one rank's communicator bootstrap fails -> every rank returns False, the rank that could drops its
    communicator, and later calls return False on all ranks with no collective (gemma_amd/dist.py)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_native_init_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = open(tmp_path / "r0.txt").read().split()
    r1 = open(tmp_path / "r1.txt").read().split()
    assert r0[:3] == ["0", "0", "0"] and r1[:3] == ["0", "0", "0"]
    assert r0[3] == "1" and r1[3] == "1"          # one attempt each, never a second
    assert r0[4] == "1" and r1[4] == "0"          # the rank that succeeded finalises its communicator
    assert float(r0[5]) < 1.0 and float(r1[5]) < 1.0


class _FakeSelftestLib:
    """The communicator's first contact (gemma_hip_comm_selftest) on a CPU box: `mode` decides what rank 1 does with it."""

    def __init__(self, rank, mode):
        self.rank, self.mode = rank, mode
        self.finalizes = 0

    def gemma_hip_comm_selftest(self, stream):
        import time
        if self.rank == 1 and self.mode == "fail":
            return 4
        if self.rank == 1 and self.mode == "hang":
            time.sleep(30.0)
        return 0

    def gemma_hip_last_error(self):
        return b"comm selftest: the all-reduce returned a wrong sum"

    def gemma_hip_comm_finalize(self):
        self.finalizes += 1
        return 0


def _selftest_worker(rank, world, port, outdir, mode):
    sys.path.insert(0, ROOT)
    import time
    import types

    import torch
    import torch.distributed as dist
    from gemma_amd import _lib as L
    from gemma_amd import dist as gdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fake = _FakeSelftestLib(rank, mode)
    L.lib = lambda: fake
    torch.cuda.current_stream = lambda: types.SimpleNamespace(cuda_stream=0)
    torch.cuda.is_available = lambda: False
    gdist._native_ready = True  # as after a successful native_comm_init
    t0 = time.time()
    ok = gdist.native_comm_selftest(timeout=2.0)
    dt = time.time() - t0
    # the control plane the bench uses around its timed region: agreement, barrier, clock exchange -- CPU tensors on gloo
    assert gdist.ctl_device().type == "cpu"
    assert gdist.agree(True) and not gdist.agree(rank == 0)
    gdist.ctl_barrier()
    tv = [0.0, 0.0]
    tv[rank] = 1.5 + rank
    assert gdist.ctl_allreduce(tv, "sum") == [1.5, 2.5]
    assert gdist.ctl_allreduce([float(rank)], "max") == [1.0]
    fin, res = gdist._with_deadline(lambda: 7, 1.0)
    assert fin and res == 7
    fin, res = gdist._with_deadline(lambda: 1 / 0, 1.0)
    assert fin and "ZeroDivisionError" in res
    with open(os.path.join(outdir, "s%d.txt" % rank), "w") as f:
        f.write("%d %d %d %d %.3f %s" % (ok, gdist._native_ready, gdist._native_poisoned, fake.finalizes, dt,
                                         gdist.native_comm_error().replace(" ", "_")))
    gdist.ctl_barrier()
    if mode == "hang":
        os._exit(0)  # rank 1's helper thread is still asleep inside the fake library
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["pass", "fail", "hang"])
def test_comm_selftest_is_an_agreement_under_a_deadline(tmp_path, mode):
    """The staged start (round 6): the communicator's 1 KiB self-test runs under a wall-clock deadline and its outcome is AGREED --
    one rank that fails (or never answers) makes every rank drop the native transport; a rank whose helper is stuck inside the library
    never touches the comm API again (no finalize)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_selftest_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    r0 = open(tmp_path / "s0.txt").read().split()
    r1 = open(tmp_path / "s1.txt").read().split()
    if mode == "pass":
        assert r0[:4] == ["1", "1", "0", "0"] and r1[:4] == ["1", "1", "0", "0"]
    elif mode == "fail":
        assert r0[:4] == ["0", "0", "0", "1"] and r1[:4] == ["0", "0", "0", "1"]
        assert "wrong_sum" in r1[5] and "another_rank" in r0[5]
    else:
        assert r0[:4] == ["0", "0", "0", "1"]      # the healthy rank finalises
        assert r1[:4] == ["0", "0", "1", "0"]      # the stuck one is poisoned and leaves the API alone
        assert 1.9 < float(r1[4]) < 10.0 and "no_answer" in r1[5]
