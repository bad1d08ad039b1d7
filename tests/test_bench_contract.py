"""The one-line JSON contract of bench.py, checked on the last line the GPU box produced (profiles/r1_z_bench_default.json.log) and on
bench.py's own source (the fields are literal keys there): metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload + roofline{bound, achieved, peak, unit, frac, traffic} +
cpu_baseline{value, unit, cores, kind, sample}."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_line():
    with open(os.path.join(REPO, "profiles", "r1_z_bench_default.json.log")) as f:
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
    if isinstance(base, dict) and "metric" in base:
        assert str(base["metric"]).split()[0].lower() in d["metric"].lower() or "ray" in d["metric"].lower()


def test_bench_source_keeps_the_driver_flags_and_defaults():
    src = open(os.path.join(REPO, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in src, flag
    for key in ('"roofline"', '"cpu_baseline"', '"vs_baseline"', '"higher_is_better"', '"scaling"', '"workload"'):
        assert key in src, key
