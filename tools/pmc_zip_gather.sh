#!/bin/bash
# What bounds path C's featurisation forward (zip_encode_fwd_all_kernel, C = 1 proposal levels and the C = 4 NeRF level): address unit (TA),
# L1 (TCP), translation (UTCL1), L2 (TCC) and issue counters (the TA_* passes return nothing on this pool and are not run) of the train step, separate rocprofv3 --pmc passes (no trace domains beside
# --kernel-trace).  Output: gpurun_out/pmc_zip_gather/{pass}/ + summary.txt (one block per kernel, averages per launch; the derived lines
# at the end relate them to the launch time and to the scattered-row rate of tools/probes/gather_probe.hip, profiles/r4_a_gather_probe.txt).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_zip_gather
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -o p -- python $ROOT/tools/bench_zip.py --rays 65536 --steps 2 --train-only > /dev/null 2>&1 < /dev/null; }
run tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
run utc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_BUSY_avr
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM
run sq2 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_WAVES
run sq3 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
{
for k in "zip_encode_fwd_all_kernelIfDF16_Li1ELb1" "zip_encode_fwd_all_kernelI6__half"; do
  python $ROOT/tools/pmc_summary.py $OUT "$k" | grep -v "^wait_\|^active_inst\|^L2 hit"
  python - "$OUT" "$k" <<'PY'
import csv, glob, os, sys
root, pat = sys.argv[1], sys.argv[2]
tot, dur = {}, []
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    per = {}
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        v = v[1:] if len(v) > 1 else v
        tot[k] = sum(v) / len(v)
for f in glob.glob(os.path.join(root, "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
us = sum(dur) / max(1, len(dur)) / 1e3
g = lambda k: tot.get(k, float("nan"))
CUS, GHZ, SCATTER = 256, 2.1, 54.6e9            # scattered 4- / 8- / 16-byte rows per second on this part (gather_probe, every row one EA request)
cyc = us * 1e-6 * GHZ * 1e9                      # shader-clock cycles of one launch (2.1 GHz nominal; GRBM_GUI_ACTIVE below is the measured figure)
print(f"derived: launch {us:.0f} us = {cyc / 1e6:.2f} M cycles at {GHZ} GHz (GRBM_GUI_ACTIVE {g('GRBM_GUI_ACTIVE') / 1e6:.2f} M)")
print(f"derived: fabric (EA) read requests {g('TCC_EA0_RDREQ_sum') / 1e6:.1f} M -> {g('TCC_EA0_RDREQ_sum') / SCATTER * 1e3:.2f} ms at the scattered-row rate "
      f"({g('TCC_EA0_RDREQ_sum') / SCATTER * 1e6 / us:.2f} of the launch)")
print(f"derived: L2 requests {g('TCC_REQ_sum') / 1e6:.1f} M = {g('TCC_REQ_sum') / (us * 1e-6) / 1e9:.1f} G/s; hit rate {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}; "
      f"tag stall cycles / request {g('TCC_TAG_STALL_sum') / g('TCC_REQ_sum'):.3f}")
print(f"derived: L1 line accesses (TCP_TOTAL_CACHE_ACCESSES) {g('TCP_TOTAL_CACHE_ACCESSES_sum') / 1e6:.1f} M; L1 -> L2 read requests {g('TCP_TCC_READ_REQ_sum') / 1e6:.1f} M "
      f"(L1 hit rate {1 - g('TCP_TCC_READ_REQ_sum') / g('TCP_TOTAL_CACHE_ACCESSES_sum'):.3f})")
print(f"derived: L1 busy (TCP_GATE_EN2 / TCP_GATE_EN1) {g('TCP_GATE_EN2_sum') / g('TCP_GATE_EN1_sum'):.3f}; L1 line accesses per CU-cycle {g('TCP_TOTAL_CACHE_ACCESSES_sum') / (CUS * cyc):.3f} "
      f"(the tag pipe takes one line per cycle); pending-request stall {g('TCP_PENDING_STALL_CYCLES_sum') / (CUS * cyc):.3f}, TCR->TCP stall {g('TCP_TCR_TCP_STALL_CYCLES_sum') / (CUS * cyc):.3f}, "
      f"read tag-conflict stall {g('TCP_READ_TAGCONFLICT_STALL_CYCLES_sum') / (CUS * cyc):.3f} of the CU-cycles")
print(f"derived: mean L1->L2 read latency {g('TCP_TCC_READ_REQ_LATENCY_sum') / g('TCP_TCC_READ_REQ_sum'):.0f} cycles; "
      f"UTCL1 translation misses {g('TCP_UTCL1_TRANSLATION_MISS_sum') / 1e6:.2f} M of {g('TCP_UTCL1_REQUEST_sum') / 1e6:.1f} M requests")
print(f"derived: VMEM read instructions {g('SQ_INSTS_VMEM_RD') / 1e6:.2f} M, VALU instructions {g('SQ_INSTS_VALU') / 1e6:.1f} M, waves {g('SQ_WAVES') / 1e3:.0f} k; "
      f"issue-stalled wave cycles / wave cycles {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f}, waiting (s_waitcnt) {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}")
PY
done
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
