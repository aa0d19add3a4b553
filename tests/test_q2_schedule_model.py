"""The dynamic schedule of the stage-2 back-transformation (gemma_amd/csrc/eigh2.hip.h: eig2_apply_q2 / q2_apply_kernel, round 3)
modelled on the CPU: the chase steps k are cut into segments of about equal numbers of groups, tasks (segment, row block) are
drawn from a counter in segment-major order by a fixed set of workers, and a task waits for the previous segment of its row
block.  The model checks what the kernel relies on: every task runs once, a row block's segments run in order, nobody waits for
a task that has not been claimed (no deadlock with any number of workers), and the cut is balanced.  The device side is
tests/test_gpu_eigh.py::test_stage2_backtransform_dynamic_schedule (bit-identical eigenvectors)."""
import heapq

import numpy as np
import pytest

E2_B, E2_NB, Q2_MAXSEG = 128, 32, 64


def segments(n, workers, nseg_override=None):
    """kseg as eig2_apply_q2 builds it"""
    kmaxall = (n - 1 + E2_B - 1) // E2_B
    nJ = (n - 2 + E2_NB - 1) // E2_NB
    nrb = (n + 63) // 64
    groups = []
    for k in range(kmaxall):
        lim = n - 2 - k * E2_B
        groups.append(0 if lim < 0 else min(lim // E2_NB, nJ - 1) + 1)
    total = sum(groups)
    nseg = min(Q2_MAXSEG, kmaxall, (24 * workers + nrb - 1) // nrb)
    if nseg_override is not None:
        nseg = max(1, min(nseg_override, Q2_MAXSEG, kmaxall))
    kseg, acc = [0], 0
    for k in range(kmaxall):
        if len(kseg) >= nseg:
            break
        acc += groups[k]
        if acc * nseg >= total * len(kseg) and k + 1 < kmaxall:
            kseg.append(k + 1)
    kseg.append(kmaxall)
    return kseg, groups, nrb


@pytest.mark.parametrize("n,workers", [(20000, 256), (24576, 512), (50000, 512), (1538, 7), (2050, 512), (1000, 3)])
def test_cut_covers_every_chase_step_once_and_is_balanced(n, workers):
    kseg, groups, nrb = segments(n, workers)
    assert kseg[0] == 0 and kseg[-1] == len(groups) and all(a < b for a, b in zip(kseg, kseg[1:]))
    per = [sum(groups[a:b]) for a, b in zip(kseg, kseg[1:])]
    assert sum(per) == sum(groups)
    if len(per) > 2:
        ideal = sum(groups) / len(per)
        assert max(per) <= ideal + max(groups) + 1  # a segment overshoots its share by at most one chase step


@pytest.mark.parametrize("n,workers,nseg", [(20000, 256, None), (50000, 512, None), (1538, 7, 5), (2050, 512, 64), (1000, 3, 1), (4096, 1, 9)])
def test_segment_major_claims_never_deadlock_and_keep_row_blocks_in_order(n, workers, nseg):
    kseg, groups, nrb = segments(n, workers, nseg)
    nseg = len(kseg) - 1
    cost = [sum(groups[a:b]) for a, b in zip(kseg, kseg[1:])]
    rng = np.random.default_rng(n + workers)
    ntasks = nseg * nrb
    done_at = {}              # (seg, rb) -> finish time
    order = {rb: [] for rb in range(nrb)}
    free = [(0.0, w) for w in range(min(workers, ntasks))]
    heapq.heapify(free)
    nxt = 0
    while nxt < ntasks:
        t, w = heapq.heappop(free)            # the worker that becomes free first claims the next task (the atomic counter)
        seg, rb = divmod(nxt, nrb)
        nxt += 1
        start = t
        if seg > 0:
            # the predecessor has a SMALLER task index: it was claimed earlier, by a worker that is running or done -- the
            # model would raise KeyError here if a claim could ever precede its predecessor's
            start = max(start, done_at[(seg - 1, rb)])
        fin = start + cost[seg] * rng.uniform(0.8, 1.25)
        done_at[(seg, rb)] = fin
        order[rb].append(seg)
        heapq.heappush(free, (fin, w))
    assert len(done_at) == ntasks
    assert all(order[rb] == list(range(nseg)) for rb in range(nrb))
    for rb in range(nrb):
        assert all(done_at[(sg, rb)] <= done_at[(sg + 1, rb)] for sg in range(nseg - 1))
    # balance: with at least as many row blocks as workers the makespan is within a task of the ideal
    if nrb >= workers and nseg >= 8:
        ideal = sum(cost) * nrb / min(workers, ntasks)
        assert max(done_at.values()) <= 1.25 * ideal * 1.15 + 2 * max(cost)
