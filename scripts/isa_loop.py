"""Steady-state K loop of a kernel inside a host binary / shared library, as tests/test_isa_schedule.py extracts it -- for kernel
variants built into scripts/abl_bin/ (development aid).  usage: python scripts/isa_loop.py <binary> <kernel symbol fragment> [--dump]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_isa_schedule as T  # noqa: E402

binary, sym = sys.argv[1], sys.argv[2]
work = tempfile.mkdtemp()
shutil.copy(binary, os.path.join(work, "lib.so"))
subprocess.run([T.OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True)
for f in sorted(os.listdir(work)):
    if "gfx950" not in f:
        continue
    dis = subprocess.run([T.OBJDUMP, "-d", f], cwd=work, check=True, capture_output=True, text=True).stdout
    m = re.search(r"^[0-9a-f]+ <(_ZN9gemma_hip\d+%s[^>]*)>:\n(.*?)(?=^\S|\Z)" % sym, dis, re.S | re.M)
    if not m:
        continue
    ops, body = T._steady_loop(m.group(2).split("\n"))
    print("kernel", m.group(1), "instructions", len(ops), "scratch ops", sum(o.startswith("scratch_") for o in ops))
    cls = lambda o: ("MAT" if re.match(r"v_s?mfma", o) else "LDS" if o.startswith("ds_") else "DMA" if o.startswith("global_load_lds") else
                     "VMEM" if o.startswith(("global_", "buffer_", "scratch_")) else "WAIT" if o.startswith("s_waitcnt") else
                     "BAR" if o.startswith("s_barrier") else "NOP" if o.startswith("s_nop") else "SALU" if o.startswith("s_") else "VALU")
    import collections
    print("loop:", len(body), dict(collections.Counter(cls(o) for o in body)))
    # run lengths of non-matrix instructions between consecutive matrix instructions
    gaps, g = [], 0
    for o in body:
        if cls(o) == "MAT":
            gaps.append(g); g = 0
        elif cls(o) in ("VALU", "LDS", "DMA", "SALU"):
            g += 1
    print("issue-slot instructions between matrix instructions:", gaps)
    print("waits:", [o for o in body if o.startswith("s_waitcnt")])
    if "--dump" in sys.argv:
        print("\n".join(body))
