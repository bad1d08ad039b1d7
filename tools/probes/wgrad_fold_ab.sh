cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="--no-frame --no-cpu --no-eager --no-f32 --no-dropin --no-paths --no-ert-scene --steps 30 --warmup 5"
for i in 1 2 3; do
for mm in 1 0; do
SNERF_WGRAD_FOLD=$mm python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fold $mm:', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['kernel_ms_per_step'])"
done; done
