"""SNP-block sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

SNPs are independent ("the computations are independent per SNP", GEMMA src/lmm.cpp:1514); the only
shared state is read-only: (U, eval, UtW, Uty) and the null-model scalars.  So the whole multi-GPU
protocol is: rank 0 computes the eigendecomposition, ONE broadcast round ships the four tensors,
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


def broadcast_state(tensors, src=0, group=None):
    """The single broadcast of (U, eval, UtW, Uty[, scalars]) -- in place on every rank.
    Coalesced into one flat buffer when the tensors share dtype/device and are small; U (n^2) is
    sent on its own so no 3.2 GB staging copy is made."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tensors
    for t in tensors:
        dist.broadcast(t, src=src, group=group)
    return tensors


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
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = []
    for r, part in enumerate(parts):
        lo, hi = shard_range(p_total, r, world)
        out.append(part[: hi - lo])
    return torch.cat(out, dim=0)
