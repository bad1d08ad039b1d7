#!/bin/bash
# HBM counters of the stand-alone GridEncoder's kernels (bench.py's grid_encoder leg: 14.7 M ray-ordered points, L = 10, C = 4, half table):
# separate rocprofv3 --pmc passes -> gpurun_out/pmc_grid/summary.txt + roofline_traffic_grid.json (merged into profiles/roofline_traffic_paths.json)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/pmc_grid; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $spec; tag=$1; shift
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/g/$tag -o p -- python $ROOT/tools/bench_grid.py --once > /dev/null 2>&1 < /dev/null
done
cd $ROOT
# (per launch: the backward runs its writer / accumulate once per (level, chunk) -- 30 launches at the default 1 GB workspace -- the count once per level,
# the transposition once per group of levels: pmc_summary.py prints the AVERAGE launch and the number of dispatches it saw)
{ for k in "g3_fwd_kernelIDF16_Li4ELi0" "g3_write_staged_kernelIDF16_Li4ELb1" "g3_accumulate_kernel<4, true, float>" "g3_count_kernel" "g3_transpose_kernel" "grid_fwd_kernel<__half, 3, 4>" "grid_bwd_kernel<__half, 3, 4>"; do python tools/pmc_summary.py $O/g "$k"; done; } > $O/summary.txt 2>&1
python - <<'PY'
import json, os, re
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
txt = open(os.path.join(root, "gpurun_out/pmc_grid/summary.txt")).read()
def block(pat):
    m = re.search(r"kernel~'" + re.escape(pat) + r"'.*?(?=\n# |\Z)", txt, re.S)
    b = m.group(0) if m else ""
    f = re.search(r"FETCH_SIZE\s+([0-9.]+)", b); w = re.search(r"WRITE_SIZE\s+([0-9.]+)", b)
    return (float(f.group(1)) * 1024 if f else None), (float(w.group(1)) * 1024 if w else None)
out = {}
f, w = block("g3_fwd_kernelIDF16_Li4ELi0")
if f is not None and w is not None:
    out["grid_encoder_fwd"] = {"bytes": f + w, "bytes_upper": 2 * f + w, "source": os.environ.get("PMC_SOURCE", "profiles/r6_x_grid_encoder_pmc.txt"),
        "how": "FETCH_SIZE (one 64-B request per gathered row: lower bound; x2 if the requests are 128 B) + WRITE_SIZE, rocprofv3 --pmc, 14 680 064 points"}
json.dump(out, open(os.path.join(root, "gpurun_out/pmc_grid/roofline_traffic_grid.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/g
