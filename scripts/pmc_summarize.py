"""scripts/pmc_bench.sh's summary.csv (+ the kernel traces of its passes) -> the per-launch figures bench.py reads from
profiles/pmc_traffic.json.  HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 counts the 128-byte requests of 16-byte-per-lane
loads at 64 B: MI355X_MICROARCH.md, HBM section).  The matrix kernel is whichever i8gemm_* kernel the passes saw -- its NAME goes into
the file, and bench.py refuses the figures when the library reports another kernel."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
tab = collections.defaultdict(dict)
for ln in open(out + "/summary.csv").read().splitlines()[1:]:
    kern, counter, launches, mean = ln.rsplit(",", 3)  # kernel names hold commas (template arguments)
    tab[kern][counter] = (float(mean), int(launches))


def hbm(k):
    c = tab[k]
    return 2.0 * 1024.0 * c.get("FETCH_SIZE", (0, 0))[0] + 1024.0 * c.get("WRITE_SIZE", (0, 0))[0]


mm = [k for k in tab if "i8gemm" in k]
mk = max(mm, key=lambda k: tab[k].get("GRBM_GUI_ACTIVE", (0, 0))[0]) if mm else None
assoc = [k for k in tab if any(t in k for t in ("cheb_scan", "cheb_search", "lmm_assoc1", "table_reduce", "table_v2"))]
post = [k for k in tab if any(t in k for t in ("i8_combine", "i8_surplus"))]
# launch durations of the matrix kernel under the counters (any pass's kernel trace)
dur = []
for f in glob.glob(out + "/pass*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if mk and r["Kernel_Name"].split("(")[0] == mk:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
res = {"n": 20000, "batch": 20000,
       "correction": "gfx950 FETCH_SIZE counts the 128-B requests of 16-B/lane loads at 64 B: x2 (MI355X_MICROARCH.md, HBM section); "
                     "WRITE_SIZE as reported; hbm bytes = 2 x FETCH + WRITE (KB -> bytes)"}
if mk:
    c = tab[mk]
    hit, miss = c.get("TCC_HIT_sum", (0, 0))[0], c.get("TCC_MISS_sum", (0, 0))[0]
    act, busy = c.get("GRBM_GUI_ACTIVE", (0, 0))[0], c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0]
    ms = sum(dur) / len(dur) if dur else None
    res.update({"i8gemm_kernel": mk, "i8gemm_hbm_bytes_per_launch": round(hbm(mk)),
                "i8gemm_fetch_bytes": round(2048.0 * c.get("FETCH_SIZE", (0, 0))[0]), "i8gemm_write_bytes": round(1024.0 * c.get("WRITE_SIZE", (0, 0))[0]),
                "i8gemm_tcc_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None,
                "i8gemm_launch_ms_under_counters": round(ms, 3) if ms else None,
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (rounds 2-4 used the same
                # conventions: round 4's 7.048e8 / 8 / 56.3 ms = 1.57 GHz, 7.224e10 / (7.048e8 x 128) = 0.80)
                "i8gemm_clock_GHz": round(act / 8.0 / (ms * 1e6), 3) if ms and act else None,
                "i8gemm_mfma_util": round(busy / (act * 128.0), 4) if act else None,
                "i8gemm_wave_cycles": {k: c[k][0] for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in c},
                "i8gemm_ea_rdreq": {k: c[k][0] for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum") if k in c},
                "i8gemm_launches_counted": c.get("FETCH_SIZE", (0, 0))[1]})
res["assoc_hbm_bytes_per_launch"] = round(sum(hbm(k) for k in assoc))
res["assoc_stage_kernels"] = sorted(assoc)
res["utx_post_hbm_bytes_per_launch"] = round(sum(hbm(k) for k in post))
print(json.dumps(res, indent=1))
