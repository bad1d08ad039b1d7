#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one directory per pass) for one kernel-name substring."""
import csv, glob, os, sys
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "gemm")
tot = {}
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    rows = list(csv.DictReader(open(f)))
    per = {}
    for r in rows:
        if pat not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        v = v[1:] if len(v) > 1 else v            # drop the first (cold) dispatch
        tot[k] = sum(v) / len(v)
dur = []
for f in glob.glob(os.path.join(root, "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"# {root}  kernel~'{pat}'  avg duration under profiling {sum(dur) / max(1, len(dur)) / 1e3:.1f} us over {len(dur)} dispatches")
for k in sorted(tot):
    print(f"{k:32s} {tot[k]:18.1f}")
g = tot.get
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    print(f"wait_any/wave_cycles      {g('SQ_WAIT_ANY', 0) / wc:.3f}   (s_waitcnt / barrier)")
    print(f"wait_inst_any/wave_cycles {g('SQ_WAIT_INST_ANY', 0) / wc:.3f}   (issue stalls)")
    print(f"active_inst_any/wave_cyc  {g('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}")
if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CYCLES"):
    print(f"mfma_busy/(busy_cycles)   {g('SQ_VALU_MFMA_BUSY_CYCLES') / g('SQ_BUSY_CYCLES'):.3f}  (units differ per SE/XCD: see MI355X_MICROARCH.md)")
if g("SQ_LDS_IDX_ACTIVE"):
    print(f"lds_bank_conflict/idx_act {g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.3f}")
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
    print(f"L2 hit rate               {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}")
# FETCH_SIZE = TCC_EA0_RDREQ x 64 B (KB).  Calibration (tools/probes/gather_probe.hip under --pmc, profiles/r4_a_gather_probe*.txt): a wide
# COALESCED stream books exactly half of its bytes (128-byte requests tallied at 64: the guide's x2); a SCATTERED gather of 4- / 8- / 16-byte
# rows books one request = 64 B per row whatever the row width (TCC_REQ = rows, 97 % misses, TCC_EA0_RDREQ_32B = 0).  Whether such a
# request moves 64 or 128 bytes cannot be told from the counters (the scattered-row rate is the same from a 120 MB and from a 2 GB table:
# the memory side is not what limits it), so the gather kernels get both: x1 = lower bound, x2 = upper bound.
GATHER = ("zip_encode", "grid_encode", "zip_bin_emit", "zip_bin_write", "gather<", "g3_", "grid_fwd_kernel", "grid_bwd_kernel")
if g("FETCH_SIZE") is not None:
    if any(k in pat for k in GATHER):
        print(f"FETCH_SIZE, gather class (KB->bytes): {g('FETCH_SIZE') * 1024 / 1e9:.3f} GB per launch at 64 B per request (lower bound), "
              f"{2 * g('FETCH_SIZE') * 1024 / 1e9:.3f} GB at 128 B (upper bound)")
    else:
        print(f"FETCH_SIZE x2 (gfx950 correction for coalesced streams, KB->bytes): {2 * g('FETCH_SIZE') * 1024 / 1e9:.3f} GB per launch")
if g("WRITE_SIZE") is not None:
    print(f"WRITE_SIZE (KB->bytes): {g('WRITE_SIZE') * 1024 / 1e9:.3f} GB per launch")
