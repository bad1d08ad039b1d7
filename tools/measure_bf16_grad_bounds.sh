#!/bin/bash
# Records the per-parameter bf16 gradient errors of tests/test_paths.py::*backward_vs_autograd (maximum over 3 runs per backend) into
# $1 (default gpurun_out/bf16_grad_bounds.json); tests/bf16_grad_bounds.json is that file, committed.  Run on the GPU box for the
# "hip" entries (-m gpu) and anywhere for the "emulated" ones.
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/bf16_grad_bounds.json}
mkdir -p "$(dirname "$OUT")"
for i in 1 2 3; do
  SNERF_DUMP_BF16_BOUNDS=$OUT timeout -k 5 300 python -m pytest tests/test_paths.py -q -k "backward_vs_autograd" > /dev/null 2>&1 < /dev/null
done
python - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for t, x in sorted(d.items()):
    for b, v in x.items():
        worst = max(v, key=v.get)
        print(f"{t} [{b}]: {len(v)} parameters, worst {worst} {v[worst]:.3e}")
PY
