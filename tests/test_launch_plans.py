"""Host-side launch plans of two round-4 kernels, on the CPU: the cross-XCD tile raster of the records kernel
(s2_build_raster) and the K-slice plan of the fp64 GEMM (gemm_kslice_plan).  tests/cpp/raster_slices_check.hip is compiled
with hipcc (host code only runs) -- the functions under test are the ones the library calls, not restatements."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = next((c for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")) if c and os.path.exists(c)), None)


@pytest.mark.skipif(HIPCC is None, reason="needs hipcc")
def test_raster_covers_every_tile_and_slices_add_up(tmp_path):
    exe = str(tmp_path / "raster_slices_check")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "gemma_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "raster_slices_check.hip"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad = 0" in r.stdout
