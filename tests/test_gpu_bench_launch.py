"""`python bench.py --gpus N` with no torch.distributed environment must launch its N ranks itself: two
ranks on the ONE device of the test box -- gloo carries torch.distributed's part, the library's communicator runs over its
shared-memory test transport behind the same entry points as RCCL (GEMMA_HIP_COMM=shm; RCCL refuses two ranks on one
device) -- and the JSON line must say n_gpus = 2 and that the broadcast went through the library's communicator."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_bench_gpus_n_launches_its_own_ranks(gpu_api, world):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"BENCH_FORCE_DEVICE": "0", "BENCH_DIST_BACKEND": "gloo", "GEMMA_HIP_COMM": "shm"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                        "--individuals", "3000", "--batch", "3000", "--kin-snps", "6000", "--cpu-sample", "0",
                        "--fp64-steps", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "weak"
    assert d["config"]["setup"]["broadcast"].startswith("native:"), d["config"]["setup"]["broadcast"]
    assert d["config"]["parallelism"] == "snp-shard x%d" % world
    assert d["value"] > 0 and d["config"]["nan_p_wald"] == 0
    am = d["amdahl"]
    assert set(am["projected_total_s"]) == {"1", "2", "4", "8"} and am["projected_total_s"]["8"] < am["projected_total_s"]["1"]


def _run_bench(env_extra, world=2, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"BENCH_FORCE_DEVICE": "0", "BENCH_COMM_DEADLINE": "60"})
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                        "--individuals", "3000", "--batch", "3000", "--kin-snps", "6000", "--cpu-sample", "0",
                        "--fp64-steps", "0"], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("fail,mode", [("init", "torch"), ("selftest", "torch"), ("allreduce", "torch"), ("bcast", "torch"),
                                       ("allreduce_large", "replicated")])
def test_a_failing_transport_still_yields_a_complete_line(gpu_api, fail, mode):
    """the first N > 1 contact must not come back empty.  GEMMA_HIP_COMM_FAIL makes the library's communicator fail at
    one point of the staged start -- its creation, its 1 KiB self-test (directly, or through a failing all-reduce / broadcast), or the
    first LARGE collective of the real setup (the all-reduce of the n^2 kinship sums, after a passed self-test; at this n nothing large
    is broadcast: the solve is one-stage, replicated) -- and the run must still end with rc 0, ONE line, a positive
    whole-job value, every rank seen, and the reason under config.comm.error; the setup then went over torch.distributed (gloo here,
    labelled) or, when a collective failed inside the native setup, was replicated on every rank."""
    d = _run_bench({"BENCH_DIST_BACKEND": "gloo", "GEMMA_HIP_COMM": "shm", "GEMMA_HIP_COMM_FAIL": fail})
    cm = d["config"]["comm"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["nan_p_wald"] == 0
    assert d["config"]["ranks_seen"] == 2 and all(v and v > 0 for v in d["config"]["per_rank"]["value"])
    assert cm["setup_mode"] == mode, cm
    assert cm["error"] and "injected" in cm["error"], cm
    assert any(not t["ok"] for t in cm["staged_start"])
    if mode == "torch":
        assert d["config"]["setup"]["broadcast"].startswith("torch.distributed broadcast"), d["config"]["setup"]["broadcast"]
    else:
        assert d["config"]["setup"]["broadcast"].startswith("none:"), d["config"]["setup"]["broadcast"]


def test_two_ranks_on_one_device_without_any_test_hook_still_yield_a_line(gpu_api):
    """The driver's own command shape (default backends: control plane gloo, device collectives RCCL) on a box where RCCL cannot work --
    two ranks on ONE device, which RCCL refuses: neither the library's communicator nor torch's comes up, every rank runs the setup itself,
    and the line says so."""
    d = _run_bench({})
    cm = d["config"]["comm"]
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert cm["setup_mode"] == "replicated" and cm["error"], cm
    assert d["config"]["ranks_seen"] == 2


@pytest.mark.parametrize("backend", ["gloo", None])
def test_native_flow_reports_its_collectives(gpu_api, backend):
    """the healthy path: staged start passed, the line carries bytes and 1-GiB pieces per collective of the library's communicator.
    backend None = the process group the driver's run gets (control plane gloo, device collectives RCCL -- which is never touched when the
    library's own communicator carries the setup: torch's lazy RCCL communicator is not even created, so this also runs on one device)."""
    env = {"GEMMA_HIP_COMM": "shm", "GEMMA_HIP_COMM_TIMING": "1"}
    if backend:
        env["BENCH_DIST_BACKEND"] = backend
    d = _run_bench(env)
    cm = d["config"]["comm"]
    assert cm["control_plane"] == ("gloo" if backend else "cpu:gloo,cuda:nccl") or "gloo" in cm["control_plane"]
    assert cm["setup_mode"] == "native" and cm["error"] is None
    assert all(t["ok"] for t in cm["staged_start"]) and len(cm["staged_start"]) == 2
    st = cm["collectives"]
    assert st["allreduce_calls"] >= 2 and st["allreduce_bytes"] >= 8.0 * 3000 * 3000 and st["bcast_calls"] >= 3
    assert st["allreduce_s"] > 0 and st["bcast_s"] > 0
