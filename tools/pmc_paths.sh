#!/bin/bash
# HBM counters of the dominant kernels of the path-C and path-B train steps (separate rocprofv3 --pmc passes, on the GPU box) ->
# gpurun_out/pmc_paths/summary.txt + profiles/roofline_traffic_paths.json (what bench.py's path_c / path_b roofline blocks cite)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/pmc_paths; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { dir=$1; tag=$2; cmd=$3; shift 3; timeout -k 5 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$dir/$tag -o p -- $cmd > /dev/null 2>&1 < /dev/null; }
ZC="python $ROOT/tools/bench_zip.py --rays 65536 --steps 2 --train-only"
BC="python $ROOT/tools/bench_classic.py --rays 32768 --steps 2 --train-only"
for spec in "fetch FETCH_SIZE" "write WRITE_SIZE" "req TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum"; do
  set -- $spec; tag=$1; shift
  run zip $tag "$ZC" "$@"
  run classic $tag "$BC" "$@"
done
cd $ROOT
# (kernel-name patterns of the fp16 compute mode: NeRF-level / proposal-level training forward, staged writer C = 4, all-levels writer C = 1, accumulate)
{ for k in "zip_encode_fwd_all_kernelI6__half" "zip_encode_fwd_all_kernelIfDF16_Li1ELb1" "zip_bin_write_staged_kernelIDF16_Li4" "zip_bin_emit_all_kernelIDF16_Li1" "zip_bin_accumulate_kernel<4, true," "zip_bin_accumulate_kernel<1, true,"; do python tools/pmc_summary.py $O/zip "$k"; done
  for k in "fmlp_kernel<0, false, true>" "fchain_bwd_kernel<0>"; do python tools/pmc_summary.py $O/classic "$k"; done; } > $O/summary.txt 2>&1
python - <<'PY'
import json, os, re
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
txt = open(os.path.join(root, "gpurun_out/pmc_paths/summary.txt")).read()
def block(pat):
    m = re.search(r"kernel~'" + re.escape(pat) + r"'.*?(?=\n# |\Z)", txt, re.S)
    b = m.group(0) if m else ""
    f = re.search(r"FETCH_SIZE\s+([0-9.]+)", b); w = re.search(r"WRITE_SIZE\s+([0-9.]+)", b)
    return (float(f.group(1)) * 1024 if f else None), (float(w.group(1)) * 1024 if w else None)
out = {}
f, w = block("zip_encode_fwd_all_kernelI6__half")
if f is not None and w is not None:
    out["zip_encode_fwd_all_nerf_train"] = {"bytes": f + w, "bytes_upper": 2 * f + w, "source": os.environ.get("PMC_SOURCE", "profiles/r4_x_pathC_pathB_pmc.txt"),
        "how": "FETCH_SIZE (one 64-B request per gathered row: lower bound; x2 if the requests are 128 B) + WRITE_SIZE, rocprofv3 --pmc, 65 536 rays"}
f, w = block("fmlp_kernel<0, false, true>")
if f is not None and w is not None:
    out["fmlp_kernel_train_fwd"] = {"bytes": 2 * f + w, "source": os.environ.get("PMC_SOURCE", "profiles/r4_x_pathC_pathB_pmc.txt"),
        "how": "FETCH_SIZE x 2 (coalesced stream correction) + WRITE_SIZE per launch (average of the coarse and the fine pass), rocprofv3 --pmc, 32 768 rays"}
json.dump(out, open(os.path.join(root, "gpurun_out/pmc_paths/roofline_traffic_paths.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/zip $O/classic
