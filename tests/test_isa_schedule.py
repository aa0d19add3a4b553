"""Build-quality guard for the dominant kernel (CPU; needs only the built library and llvm-objdump): the steady-state K loop of
i8gemm_sparse2_kernel must be what gemma_amd/csrc/i8gemm_sparse2.hip.h writes down -- 16 dense + 8 sparse matrix instructions,
4 LDS-DMA pieces, 18 ds_read_b128 (wavefronts 8 x 1), ONE counted vmcnt wait and no compiler-inserted `s_waitcnt vmcnt(0)` (round 2's kernel lost
its two-tiles-ahead prefetch to exactly that), no two consecutive matrix instructions on one accumulator, no scratch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gemma_amd", "libgemma_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _kernel_text(tmp_path, symbol):
    work = tmp_path / "o"
    work.mkdir()
    shutil.copy(LIB, work / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        dis = subprocess.run([OBJDUMP, "-d", f], cwd=work, check=True, capture_output=True, text=True).stdout
        m = re.search(r"^[0-9a-f]+ <(_ZN9gemma_hip\d+%s[^>]*)>:\n(.*?)(?=^\S|\Z)" % symbol, dis, re.S | re.M)
        if m:
            return m.group(2).split("\n")
    return None


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_sparse2_kernel_steady_state_loop_is_as_written(tmp_path):
    lines = _kernel_text(tmp_path, "i8gemm_sparse2_kernel")
    assert lines, "i8gemm_sparse2_kernel not found in the gfx950 code object"
    ops = [ln.split("//")[0].strip() for ln in lines if ln.strip()]
    # NO scratch access anywhere in the kernel, not only in the loop: a spill is a vector-memory operation, and the counted
    # s_waitcnt vmcnt of the LDS-DMA pipeline would then wait for the wrong transfers (round 4: a second copy of the K loop made
    # the allocator spill and the product came out wrong on a fraction of the tiles)
    assert not any(o.startswith("scratch_") for o in ops), [o for o in ops if o.startswith("scratch_")][:4]
    # the steady-state loop: the backward branch whose body holds the most matrix instructions
    addr = []
    for ln in lines:
        m = re.search(r"//\s*([0-9A-F]{8,16}):", ln)
        addr.append(int(m.group(1), 16) if m else None)
    best = None
    loops = []
    for i, op in enumerate(ops):
        m = re.match(r"s_cbranch_scc1\s+(\d+)", op)
        if not m or int(m.group(1)) < 32768:  # backward: the 16-bit offset is negative
            continue
        target = addr[i] + 4 + 4 * (int(m.group(1)) - 65536)
        j = next(k for k in range(i) if addr[k] == target)
        body = ops[j:i + 1]
        nm = sum(("v_mfma" in o) or ("v_smfmac" in o) for o in body)
        loops.append(body)
        if best is None or nm > best[0]:
            best = (nm, body)
    assert best is not None
    body = best[1]
    cnt = lambda pat: sum(bool(re.match(pat, o)) for o in body)
    assert cnt(r"v_mfma_i32_32x32x32_i8") == 16 and cnt(r"v_smfmac_i32_32x32x64_i8") == 8
    assert cnt(r"global_load_lds_dwordx4") == 4 and cnt(r"ds_read_b128") == 18
    assert cnt(r"s_barrier") == 1
    assert cnt(r"s_waitcnt vmcnt\(8\)") == 1 and cnt(r"s_waitcnt vmcnt\(0\)") == 0, [o for o in body if "vmcnt" in o]
    assert not any(o.startswith(("scratch_", "buffer_store", "buffer_load_dword ")) for o in body)
    mats = [re.match(r"v_s?mfmac?\w*\s+(v\[\d+:\d+\])", o).group(1) for o in body if re.match(r"v_s?mfma", o)]
    assert len(mats) == 24 and all(a != b for a, b in zip(mats, mats[1:] + mats[:1])), mats
    # every accumulator block is visited the same number of times: 4 genotype + 4 mask accumulators
    assert sorted(mats.count(a) for a in set(mats)) == [2] * 4 + [4] * 4


def _steady_loop(lines):
    """ops of the backward-branch loop that holds the most matrix instructions"""
    ops = [ln.split("//")[0].strip() for ln in lines if ln.strip()]
    addr = []
    for ln in lines:
        if not ln.strip():
            continue
        m = re.search(r"//\s*([0-9A-F]{8,16}):", ln)
        addr.append(int(m.group(1), 16) if m else None)
    best = None
    for i, op in enumerate(ops):
        m = re.match(r"s_cbranch_scc1\s+(\d+)", op)
        if not m or int(m.group(1)) < 32768:
            continue
        target = addr[i] + 4 + 4 * (int(m.group(1)) - 65536)
        j = next(k for k in range(i) if addr[k] == target)
        body = ops[j:i + 1]
        nm = sum(("v_mfma" in o) or ("v_smfmac" in o) for o in body)
        if best is None or nm > best[0]:
            best = (nm, body)
    return ops, (best[1] if best else None)


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_sixteen_row_records_kernel_steady_state_loop_is_as_written(tmp_path):
    """i8gemm_sparse2_r16_kernel (the shipped form of the records kernel since the end of round 4: v_mfma_i32_16x16x64_i8 +
    v_smfmac_i32_16x16x128_i8): per K-tile and wavefront 32 dense + 16 sparse instructions, 4 LDS-DMA pieces, 20 ds_read_b128 (4
    records, 16 digit fragments), one barrier, ONE counted s_waitcnt vmcnt(8) and no vmcnt(0), no lane exchange, no register copy,
    and no scratch access anywhere in the kernel (a spill is a vector-memory operation the counted wait does not know about)."""
    lines = _kernel_text(tmp_path, "i8gemm_sparse2_r16_kernel")
    assert lines, "i8gemm_sparse2_r16_kernel not found in the gfx950 code object"
    ops, body = _steady_loop(lines)
    assert not any(o.startswith("scratch_") for o in ops), [o for o in ops if o.startswith("scratch_")][:4]
    assert body is not None
    cnt = lambda pat: sum(bool(re.match(pat, o)) for o in body)
    assert cnt(r"v_mfma_i32_16x16x64_i8") == 32 and cnt(r"v_smfmac_i32_16x16x128_i8") == 16
    assert cnt(r"global_load_lds_dwordx4") == 4 and cnt(r"ds_read_b128") == 20 and cnt(r"ds_read") == 20
    assert cnt(r"s_barrier") == 1
    assert cnt(r"s_waitcnt vmcnt\(8\)") == 1 and cnt(r"s_waitcnt vmcnt\(0\)") == 0, [o for o in body if "vmcnt" in o]
    assert cnt(r"v_permlane") == 0 and cnt(r"v_mov_b32") == 0
    assert not any(o.startswith(("scratch_", "buffer_store", "buffer_load_dword ")) for o in body)
    mats = [re.match(r"v_s?mfmac?\w*\s+(v\[\d+:\d+\])", o).group(1) for o in body if re.match(r"v_s?mfma", o)]
    # 16 genotype accumulators visited twice (two pairs of K-steps), 16 mask accumulators once; never the same one twice in a row
    assert len(mats) == 48 and all(a != b for a, b in zip(mats, mats[1:] + mats[:1])), mats
    assert sorted(mats.count(a) for a in set(mats)) == [1] * 16 + [2] * 16


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_sixteen_row_genotype_only_instance_of_the_records_kernel(tmp_path):
    """i8gemm_sparse2_r16_g_kernel (round 6: the lowest digit of the 7g6m form -- genotype product alone, the SAME source as the shipped
    kernel with the sparse instructions, the mask accumulators and the M rows compiled out): 32 dense matrix instructions and no sparse
    one per K-tile and wavefront, the same 4 LDS-DMA pieces and 20 ds_read_b128, the same single counted wait, no scratch."""
    lines = _kernel_text(tmp_path, "i8gemm_sparse2_r16_g_kernel")
    assert lines, "i8gemm_sparse2_r16_g_kernel not found in the gfx950 code object"
    ops, body = _steady_loop(lines)
    assert not any(o.startswith("scratch_") for o in ops), [o for o in ops if o.startswith("scratch_")][:4]
    assert not any("v_smfmac" in o for o in ops)
    assert body is not None
    cnt = lambda pat: sum(bool(re.match(pat, o)) for o in body)
    assert cnt(r"v_mfma_i32_16x16x64_i8") == 32
    assert cnt(r"global_load_lds_dwordx4") == 4 and cnt(r"ds_read_b128") == 20 and cnt(r"ds_read") == 20
    assert cnt(r"s_barrier") == 1
    assert cnt(r"s_waitcnt vmcnt\(8\)") == 1 and cnt(r"s_waitcnt vmcnt\(0\)") == 0, [o for o in body if "vmcnt" in o]
    mats = [re.match(r"v_mfma\w*\s+(v\[\d+:\d+\])", o).group(1) for o in body if re.match(r"v_mfma", o)]
    assert len(mats) == 32 and sorted(mats.count(a) for a in set(mats)) == [2] * 16
    assert all(a != b for a, b in zip(mats, mats[1:] + mats[:1])), mats


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_run_time_mvlmm_kernel_private_memory_budget(tmp_path):
    """The run-time multivariate kernel (mvlmm_kernels_rt.hip) keeps its small matrices in private memory; its large pieces are
    CALLED (MV_OUTLINE), not inlined: 60 KB per lane.  Fully inlined it was 120 KB per lane and eight minutes of compile time, close
    to the 128 KB a wavefront's scratch can address: a change that inlines them again must not pass silently.  The fixed kernels
    stay register-resident (no more than a few hundred bytes of spill space for the largest shapes)."""
    readelf = os.path.join(os.path.dirname(OBJDUMP), "llvm-readelf")
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not found")
    work = tmp_path / "o"
    work.mkdir()
    shutil.copy(LIB, work / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
    found = {}
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", f], cwd=work, check=True, capture_output=True, text=True).stdout
        name = None
        for ln in notes.splitlines():
            m = re.search(r"\.name:\s+(\S+)", ln)
            if m:
                name = m.group(1)
            m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", ln)
            if m and name:
                found[name] = int(m.group(1))
    rt = {k: v for k, v in found.items() if "mvlmm_rt_kernel" in k or "mvlmm_null_rt_kernel" in k}
    assert len(rt) == 2, sorted(found)[:5]
    for k, v in rt.items():
        assert 0 < v <= 72 * 1024, (k, v)
    fixed = {k: v for k, v in found.items() if "12mvlmm_kernelILi" in k}
    assert len(fixed) >= 24
    assert max(fixed.values()) <= 4096, max(fixed.items(), key=lambda kv: kv[1])
    # round 5: six / seven phenotypes have fixed kernels too (two / one wavefront per workgroup: mvlmm_kernels_d6.hip, _d7.hip) -- a few KB of
    # spill space per lane where the run-time kernel they replace for those shapes needs 60 KB
    wide = {k: v for k, v in found.items() if "14mvlmm_kernel_wILi" in k}
    assert len(wide) == 7, sorted(wide)  # d = 6, 7 with c = 2, 3, 4; d = 8 with c = 2
    assert max(wide.values()) <= 6 * 1024, max(wide.items(), key=lambda kv: kv[1])


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_dense_sixteen_row_kernel_steady_state_loop(tmp_path):
    """i8gemm_dense16_kernel_t<true> (round 5: the dosage byte planes on v_mfma_i32_16x16x64_i8): per K-tile and wavefront 32 matrix
    instructions on 16 accumulators (each visited twice: the two pairs of K-steps), 16 ds_read_b128 (one per block and pair), 6 LDS-DMA
    pieces, one barrier, ONE counted s_waitcnt vmcnt(6) and no vmcnt(0), no scratch anywhere in the kernel."""
    lines = _kernel_text(tmp_path, r"i8gemm_dense16_kernel_tILb1")
    assert lines, "i8gemm_dense16_kernel_t<true> not found in the gfx950 code object"
    ops, body = _steady_loop(lines)
    assert not any(o.startswith("scratch_") for o in ops), [o for o in ops if o.startswith("scratch_")][:4]
    assert body is not None
    cnt = lambda pat: sum(bool(re.match(pat, o)) for o in body)
    assert cnt(r"v_mfma_i32_16x16x64_i8") == 32 and cnt(r"v_mfma_i32_32x32x32_i8") == 0
    assert cnt(r"global_load_lds_dwordx4") == 6 and cnt(r"ds_read_b128") == 16 and cnt(r"ds_read") == 16
    assert cnt(r"s_barrier") == 1
    assert cnt(r"s_waitcnt vmcnt\(6\)") == 1 and cnt(r"s_waitcnt vmcnt\(0\)") == 0, [o for o in body if "vmcnt" in o]
    mats = [re.match(r"v_mfma\w*\s+(v\[\d+:\d+\])", o).group(1) for o in body if re.match(r"v_mfma", o)]
    assert len(set(mats)) == 16 and all(mats.count(a) == 2 for a in set(mats)), mats
    assert all(a != b for a, b in zip(mats, mats[1:])), mats
