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
