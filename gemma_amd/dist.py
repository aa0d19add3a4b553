"""SNP-block sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

SNPs are independent ("the computations are independent per SNP", GEMMA src/lmm.cpp:1514); the only
shared state is read-only: (U, eval, UtW, Uty) and the null-model scalars.  So the whole multi-GPU
protocol is: rank 0 computes the eigendecomposition, ONE broadcast round (U on its own, everything
else coalesced into one flat buffer) ships them,
every rank analyses a contiguous range of the analysed-SNP sequence, and the 64 B/SNP SUMSTAT
records are gathered in rank order (= SNP order, what LMM::WriteFiles needs, src/lmm.cpp:204-219).
No collective runs in steady state.
"""


def shard_range(p, rank, world):
    """Contiguous block [lo, hi) of p analysed SNPs for `rank`: ceil(p/world) per rank."""
    per = (p + world - 1) // world
    lo = min(p, rank * per)
    hi = min(p, lo + per)
    return lo, hi


def _staged(t, group):
    """gloo (CPU tests, and the same-device 2-rank GPU test) moves host memory: device tensors are staged through a
    host copy there; with nccl (= RCCL, the real multi-GPU runs) tensors are used in place."""
    import torch.distributed as dist
    return t.is_cuda and "nccl" not in str(dist.get_backend(group))  # "nccl", or the mixed "cpu:gloo,cuda:nccl" of bench.py


def broadcast_state(tensors, src=0, group=None, small_limit=1 << 22):
    """The single broadcast round of (U, eval, UtW, Uty[, scalars]) -- in place on every rank.  Tensors of up to
    `small_limit` elements (eval, UtW, Uty, the null-model scalars: 8 n (c + 2) bytes) travel coalesced in ONE flat
    buffer; larger ones (U, n^2) are sent on their own so that no second 8 n^2-byte staging copy is made.  At n = 20 000
    that is two collectives in total: 3.2 GB + 0.5 MB."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tensors
    small = [t for t in tensors if t.numel() <= small_limit]
    for t in tensors:
        if t.numel() > small_limit:
            if _staged(t, group):
                h = t.cpu()
                dist.broadcast(h, src=src, group=group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src, group=group)
    # one flat buffer per (dtype, device) class -- in practice one: everything here is float64 on this rank's GPU
    classes = {}
    for t in small:
        classes.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), ts in classes.items():
        flat = torch.cat([t.reshape(-1) for t in ts])
        if _staged(flat, group):
            h = flat.cpu()
            dist.broadcast(h, src=src, group=group)
            flat = h.to(device)
        else:
            dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            t.copy_(flat[off:off + t.numel()].reshape(t.shape))
            off += t.numel()
    return tensors


_native_ready = False
_native_poisoned = False  # a bootstrap thread timed out inside the library: the comm API is off limits for this process
_native_failed = False    # the AGREED outcome of an earlier attempt was "no": every rank knows it, no rank tries again


def native_comm_init(group=None, timeout=180.0):
    """Bootstrap the LIBRARY's own RCCL communicator (csrc/comm.hip.h: ncclGetUniqueId on rank 0, the 128-byte id shipped
    through torch.distributed's store, ncclCommInitRank on every rank's device).  Every rank learns whether ALL ranks
    succeeded (one MIN all-reduce), so that they take the same branch afterwards.  Returns True when the native
    communicator is usable.  The function is a collective: a failed attempt is remembered as the result all ranks AGREED on
    (`_native_failed`, set on every rank by the same all-reduce), so a later call returns False everywhere without any
    collective -- a per-rank early return (one rank's helper thread timed out, the others' did not) would leave the others
    waiting in a broadcast the first never joins."""
    global _native_ready, _native_poisoned, _native_failed
    import ctypes as C
    import torch
    import torch.distributed as dist
    from . import _lib as L
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return False
    if _native_ready:
        return True
    if _native_failed:
        return False
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ok = 1
    try:
        if _native_poisoned:  # cannot happen after an agreed failure; kept so that a poisoned rank still reaches the agreement
            raise RuntimeError("comm API poisoned")
        ident = C.create_string_buffer(L.COMM_ID_BYTES)
        box = [None]
        if rank == 0:
            L.check(L.lib().gemma_hip_comm_unique_id(ident), "comm_unique_id")
            box[0] = ident.raw
        # on the control plane explicitly: a mixed-backend group would otherwise be free to serialise the object onto the device and
        # make torch's own (lazy) RCCL communicator the first thing to come up -- outside any deadline
        dist.broadcast_object_list(box, src=0, group=group, device=ctl_device(group))
        ident = C.create_string_buffer(box[0], L.COMM_ID_BYTES)
        # ncclCommInitRank is itself a collective: run it on a helper thread with a deadline, so that a bootstrap that
        # never completes costs a delay and the agreed fallback below, not the run
        import threading
        res = {}
        dev_index = torch.cuda.current_device()

        def _init():
            try:
                torch.cuda.set_device(dev_index)
                res["rc"] = L.lib().gemma_hip_comm_init(ident, rank, world)
            except Exception as e:  # noqa: BLE001
                res["rc"] = repr(e)

        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(timeout)
        if th.is_alive():
            # the helper is still inside gemma_hip_comm_init and may complete (or mutate the library's communicator) at any
            # later time: never touch the comm API again from this process -- no finalize, no second attempt
            _native_poisoned = True
            ok = 0
        elif res.get("rc") != 0:
            ok = 0
    except Exception:  # the other ranks must still reach the agreement below
        ok = 0
    _native_ready = agree(ok == 1, group)
    _native_failed = not _native_ready
    if not _native_ready and ok == 1 and not _native_poisoned:
        L.lib().gemma_hip_comm_finalize()  # this rank could, another could not: drop the communicator, all take the fallback
    if not _native_ready:
        try:
            _native_error[0] = ("bootstrap timed out after %.0f s" % timeout) if _native_poisoned else \
                (L.lib().gemma_hip_last_error().decode() or "another rank could not create the communicator")
        except Exception:  # noqa: BLE001
            pass
    return _native_ready


_native_error = [""]


def native_comm_error():
    """why the last native_comm_init / native_comm_selftest said no on THIS rank ('' when it did not)"""
    return _native_error[0]


def ctl_device(group=None):
    """Where the small control-plane tensors live: the CPU whenever the process group has a CPU backend (gloo; bench.py's default
    group is "cpu:gloo,cuda:nccl"), so that agreements, barriers and clock exchanges never depend on the device transport whose
    health they are about; the current device for a pure-nccl group."""
    import torch
    import torch.distributed as dist
    b = str(dist.get_backend(group))
    return torch.device("cpu") if ("gloo" in b or b == "undefined") else torch.device("cuda", torch.cuda.current_device())


def agree(ok, group=None):
    """True on every rank iff `ok` is true on every rank (one MIN all-reduce of one int on the control plane)."""
    import torch
    import torch.distributed as dist
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ctl_device(group))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag[0]) == 1)


def ctl_barrier(group=None):
    """A barrier on the control plane (an all-reduce of one int; dist.barrier() of a mixed-backend group may pick the device backend)."""
    agree(True, group)


def ctl_allreduce(values, op="sum", group=None):
    """All-reduce a short list of floats on the control plane; returns a list."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=ctl_device(group))
    dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op], group=group)
    return [float(x) for x in t.cpu()]


def _with_deadline(fn, timeout):
    """Run fn() on a helper thread; (finished, result or exception repr).  A call that never returns costs `timeout`, not the run."""
    import threading
    import torch
    box = {}
    dev_index = torch.cuda.current_device() if torch.cuda.is_available() else None

    def _run():
        try:
            if dev_index is not None:
                torch.cuda.set_device(dev_index)
            box["v"] = fn()
        except Exception as e:  # noqa: BLE001
            box["e"] = repr(e)

    th = threading.Thread(target=_run, daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        return False, "no answer within %.0f s" % timeout
    if "e" in box:
        return True, box["e"]
    return True, box.get("v")


def native_comm_selftest(group=None, timeout=60.0):
    """The staged start of the library's communicator: gemma_hip_comm_selftest -- ONE KiB through ncclAllReduce
    and ncclBroadcast, every value checked -- under a wall-clock deadline, BEFORE anything n^2 goes through it.  A collective like
    native_comm_init: every rank learns whether all ranks passed; after a "no" the communicator is finalised where that is safe and
    native_comm_init answers False from then on."""
    global _native_ready, _native_poisoned, _native_failed
    import ctypes as C
    import torch
    from . import _lib as L
    if not _native_ready:
        return False
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    finished, res = _with_deadline(lambda: L.lib().gemma_hip_comm_selftest(stream), timeout)
    ok = bool(finished and res == 0)
    if not ok:
        if not finished:
            _native_poisoned = True
            _native_error[0] = "comm self-test: " + str(res)
        else:
            try:
                _native_error[0] = L.lib().gemma_hip_last_error().decode() if isinstance(res, int) else str(res)
            except Exception:  # noqa: BLE001
                _native_error[0] = str(res)
    all_ok = agree(ok, group)
    if not all_ok:
        _native_ready = False
        _native_failed = True
        if not _native_poisoned:
            L.lib().gemma_hip_comm_finalize()
        if not _native_error[0]:
            _native_error[0] = "another rank failed the communicator's self-test"
    return all_ok


def torch_device_selftest(group=None, timeout=60.0):
    """The same first contact for torch.distributed's DEVICE backend (nccl = RCCL; lazily initialised in a mixed-backend group): a
    tiny all-reduce and broadcast of device tensors under a deadline.  Returns (ok on every rank, this rank's error text)."""
    import torch
    import torch.distributed as dist

    def _probe():
        dev = torch.device("cuda", torch.cuda.current_device())
        t = torch.full((128,), float(dist.get_rank(group) + 1), dtype=torch.float64, device=dev)
        dist.all_reduce(t, group=group)
        w = dist.get_world_size(group)
        b = torch.arange(128, dtype=torch.float64, device=dev) + (1000.0 if dist.get_rank(group) == 0 else -1.0)
        dist.broadcast(b, src=0, group=group)
        torch.cuda.synchronize()
        good = bool((t == 0.5 * w * (w + 1)).all()) and bool((b == torch.arange(128, dtype=torch.float64, device=dev) + 1000.0).all())
        return 0 if good else "torch.distributed device self-test: wrong values"

    finished, res = _with_deadline(_probe, timeout)
    ok = bool(finished and res == 0)
    err = "" if ok else str(res)
    return agree(ok, group), err


def broadcast_state_native(tensors, src=0, small_limit=1 << 22):
    """broadcast_state over the library's communicator: ncclBroadcast issued by libgemma_hip.so itself on torch's current
    stream (device tensors, in place): U on its own, everything else coalesced into one flat buffer."""
    import ctypes as C
    import torch
    from . import _lib as L
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    small = [t for t in tensors if t.numel() <= small_limit]
    for t in tensors:
        if t.numel() > small_limit:
            assert t.is_contiguous()
            L.check(L.lib().gemma_hip_comm_bcast_d(C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), src, stream), "comm_bcast")
    if small:
        flat = torch.cat([t.reshape(-1).to(torch.float64) for t in small])
        L.check(L.lib().gemma_hip_comm_bcast_d(C.c_void_p(flat.data_ptr()), flat.numel() * 8, src, stream), "comm_bcast")
        off = 0
        for t in small:
            t.copy_(flat[off:off + t.numel()].reshape(t.shape))
            off += t.numel()
    torch.cuda.current_stream().synchronize()
    return tensors


def seed_plink_carry(analyse_one, lo):
    """AnalyzePlink prints the PREVIOUS SNP's beta / se for a SNP whose lambda search failed (function-scope variables,
    src/lmm.cpp:1725,1870-1884), so a shard that starts at SNP `lo` > 0 must start with the carry the unsharded run has
    there.  A successful SNP overwrites the carry with its own values and a failed one leaves it alone, so it is enough
    to analyse -- before the shard, results discarded -- the SNPs lo-1, lo-2, ... until one succeeds (almost always the
    first).  analyse_one(j) runs SNP j alone through LMM.batch and returns its logl_H1."""
    j = lo - 1
    while j >= 0:
        logl = analyse_one(j)
        if logl == logl:  # not NaN: the search succeeded, the carry now holds this SNP's beta / se
            break
        j -= 1
    return lo - 1 - j if j >= 0 else lo


def gather_sumstat(local, p_total, group=None):
    """All ranks -> rank-ordered concatenation of per-rank SUMSTAT blocks (torch tensors [l_r, 8]).
    Ranks may hold different l_r (last shard shorter): pad to ceil(p/world) and trim."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    per = (p_total + world - 1) // world
    buf = torch.zeros((per, local.shape[1]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    if _staged(buf, group):
        buf = buf.cpu()
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    parts = [p.to(local.device) for p in parts]
    out = []
    for r, part in enumerate(parts):
        lo, hi = shard_range(p_total, r, world)
        out.append(part[: hi - lo])
    return torch.cat(out, dim=0)


def allreduce_kinship(K_local, ns_local, group=None):
    """SNP-sharded kinship: every rank holds K_r = X_r X_r^T / ns_r over ITS SNPs (what kin_end returns)
    and ns_r; ONE all-reduce of the n^2 unscaled sums (plus the SNP counts) gives K = sum_r ns_r K_r / sum_r ns_r =
    X X^T / ns on every rank -- the matrix PARAM::CalcKin (src/param.cpp:1300-1321) produces, up to summation order.
    In place on K_local (torch tensor); returns (K, ns_total)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return K_local, int(ns_local)
    K_local.mul_(float(ns_local))
    ns = torch.tensor([float(ns_local)], dtype=torch.float64, device=K_local.device)
    dist.all_reduce(K_local, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(ns, op=dist.ReduceOp.SUM, group=group)
    K_local.div_(float(ns[0]))
    return K_local, int(ns[0])
