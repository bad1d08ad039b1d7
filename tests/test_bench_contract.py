"""The one-line JSON contract of bench.py, checked on the last line the GPU box produced (profiles/r4_x_bench_default.json.log) and on
bench.py's own source (the fields are literal keys there): metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload + roofline{bound, achieved, peak, unit, frac, traffic} +
cpu_baseline{value, unit, cores, kind, sample}."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_LOG = "r4_x_bench_default.json.log"          # the default line as the GPU box printed it this round


def _last_line():
    with open(os.path.join(REPO, "profiles", BENCH_LOG)) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def test_bench_line_has_the_contract_fields():
    d = _last_line()
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "bf16"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["vs_baseline"] is None                                   # BASELINE.md holds no published number for this metric on this hardware
    assert abs(d["value"] - d["config"]["global_rays_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["peak"] == 2500.0 and r["traffic"] > 2.0e9
    c = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("port", "reference") and c["unit"] == d["unit"]
    # round-2 fields of the default line: eager PyTorch-ROCm baselines on the same GPU, the fp32-parity mode's own throughput / roofline
    e, f = d["eager_baseline"], d["f32_mode"]
    assert e["fp32"] > 0 and e["bf16_autocast"] > 0 and abs(e["speedup_vs_bf16_autocast"] - d["value"] / e["bf16_autocast"]) < 0.02
    assert f["roofline"]["peak"] == 157.3 and abs(f["roofline"]["frac"] - f["roofline"]["achieved"] / 157.3) < 1e-3
    assert d["ms_per_step_median"] > 0 and d["frame"]["rays"] == 1440000
    # kernel time within the step, whole-step rate below the peak
    assert r["kernel_ms_per_step"] < d["ms_per_step"] and r["whole_step_tflops"] < r["peak"]
    # the roofline's FLOPs are counted from the launches the instrumented step recorded (real weights per launch): never more than the
    # padded work the same launches execute (the closed form over-credited the kernel by 2.5 %)
    assert 0 < r["algorithmic_flops_per_step"] <= r["padded_flops_per_step"]
    assert abs(r["achieved"] - r["algorithmic_flops_per_step"] / (r["kernel_ms_per_step"] * 1e-3) / 1e12) < 0.02 * r["achieved"]
    # the other two renderers of the reference have a driver-timed number in the same line (BASELINE configs 4 - 5, path B)
    pc, pb = d["path_c"], d["path_b"]
    assert pc["rays_per_step"] == 65536 and pc["train_ms_per_step"] > 0 and pc["frame_1920x1280_ms"] > 0 and pc["frame_ok"]
    assert pc["roofline"]["bound"] == "hbm" and abs(pc["roofline"]["frac"] - pc["roofline"]["achieved"] / 8000.0) < 1e-3
    assert pb["rays_per_step"] == 32768 and pb["roofline"]["bound"] == "mfma" and 0 < pb["roofline"]["frac"] < 1
    # early ray termination on the fitted scene: a labelled extra with both frame times, the kept fraction and the errors
    e = d["ert_scene"]
    assert e["eps_t"] == 1e-4 and e["ms_per_frame_full"] > 0 and e["ms_per_frame_ert"] > 0 and 0 < e["fine_samples_evaluated"] <= 1
    # front-to-back termination: the skipped samples' weights sum to <= eps_t, so acc AND rgb (colours in [-0.001, 1.001]) are inside it.  (Until the
    # K = 128 race of the persistent GEMM was fixed in round 4, two renders of one model could differ by more in a few 32-row blocks.)
    assert e["max_abs_err_acc"] <= 1.01e-4 and e["max_abs_err_rgb"] <= 1.1e-4
    if isinstance(base, dict) and "metric" in base:
        assert str(base["metric"]).split()[0].lower() in d["metric"].lower() or "ray" in d["metric"].lower()


def test_roofline_traffic_measurement_matches_the_kernel_source():
    """roofline.traffic is the committed PMC measurement of profiles/roofline_traffic.json, valid only for the kernel source it was taken
    on: an edit of gemm_nt8p_kernel (tile walk, staging, epilogue) without a new measurement fails HERE instead of leaving a stale
    number in the driver's line."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from kernel_hash import kernel_hash
    d = json.load(open(os.path.join(REPO, "profiles", "roofline_traffic.json")))
    assert d["kernel_sha256"] == kernel_hash(d["kernel"]), "gemm_nt8p_kernel changed: rerun tools/pmc_gemm_traffic.sh, then tools/kernel_hash.py --update"
    assert os.path.exists(os.path.join(REPO, d["source"])) and d["fetch_bytes"] > 0 and d["write_bytes"] > 0
    sys.path.insert(0, REPO)
    import bench
    t, src = bench.roofline_traffic("bf16", 8)
    assert t == d["fetch_bytes"] + d["write_bytes"] and d["source"] in src
    assert bench.roofline_traffic("f32", 8)[0] is None


def test_bench_source_keeps_the_driver_flags_and_defaults():
    src = open(os.path.join(REPO, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src, flag
    for key in ('"roofline"', '"cpu_baseline"', '"vs_baseline"', '"higher_is_better"', '"scaling"', '"workload"'):
        assert key in src, key


def test_bench_relaunches_itself_for_multi_gpu_without_a_launcher(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment must start N ranks under torch.distributed.run (127.0.0.1
    rendezvous) with its own arguments instead of refusing to run."""
    import importlib
    import subprocess
    import sys
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu_without_a_launcher():
    """The N > 1 control flow end to end on the 1-GPU box: self-launch, gloo rendezvous, both ranks on cuda:0, sharded frame render,
    strong scaling split -- rank 0 prints one JSON line with the contract fields."""
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "2", "--warmup", "1",
           "--rays", "512", "--scaling", "strong", "--no-cpu", "--no-eager", "--no-f32", "--frame-chunk", "16384"]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["rays_per_gpu_per_step"] == 256 and d["config"]["global_rays_per_step"] == 512
    assert d["comm"]["ranks"] == 2 and d["comm"]["bytes"] > 3.5e7 and d["ms_per_frame"] > 0 and "roofline" in d


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_weak_scaling_line_carries_the_strong_scaling_step():
    """`bench.py --gpus N` as the driver launches it (default --scaling weak): the line also carries `strong_scaling` -- the reference's
    --rays batch split over the ranks, launched per kernel and as a hipGraph -- beside the weak-scaling headline."""
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "2", "--warmup", "1",
           "--rays", "512", "--no-cpu", "--no-eager", "--no-f32", "--no-frame", "--no-dropin", "--no-paths", "--no-ert-scene"]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["rays_per_gpu_per_step"] == 512 and d["config"]["global_rays_per_step"] == 1024
    ss = d["strong_scaling"]
    assert ss["global_rays_per_step"] == 512 and ss["rays_per_gpu_per_step"] == 256
    assert ss["ms_per_step_launches"] > 0 and ss["ms_per_step_hipgraph"] > 0 and ss["rays_per_s_hipgraph"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_eight_ranks_strong_scaling_on_one_gpu():
    """The reference's 4096-ray batch split across EIGHT ranks (512 rays each: BASELINE config 3's `ray batches DDP over 8 x MI355X`,
    s-nerf/train.py:284-296) -- functional run of the 8-rank control flow on the 1-GPU box: gloo rendezvous, every rank on cuda:0, the
    hipGraph-captured small-batch step with the gradient exchange outside the graph; one JSON line, 8 x 512 rays per step."""
    import subprocess
    import sys
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--backend", "gloo", "--same-device", "--steps", "2", "--warmup", "1",
           "--rays", "4096", "--scaling", "strong", "--graph", "--no-cpu", "--no-eager", "--no-f32", "--no-frame", "--no-dropin", "--no-paths",
           "--no-ert-scene"]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["rays_per_gpu_per_step"] == 512 and d["config"]["global_rays_per_step"] == 4096
    assert d["comm"]["ranks"] == 8 and d["comm"]["bytes"] > 3.5e7 and d["value"] > 0 and "roofline" in d
    assert d["roofline"]["algorithmic_flops_per_step"] <= d["roofline"]["padded_flops_per_step"]
