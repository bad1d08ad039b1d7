#!/usr/bin/env python3
"""Socket power and shader clock while ONE kernel runs back to back (evidence for which kernels are power-limited: DESIGN.md section 3.1).
A child process launches the kernel in a loop for a few seconds; the parent samples `rocm-smi -c -P --json` meanwhile.

    python tools/power_trace.py gemm      # gemm_nt8p, M = 524288, N = K = 1024, uniform random operands
    python tools/power_trace.py fmlp      # fused classic NeRF 8 x 256, 6.3 M rows
    python tools/power_trace.py idle
"""
import json, os, subprocess, sys, time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from snerf_amd import classic, ops
kind, secs = sys.argv[1], float(sys.argv[2])
if kind == "gemm":
    M, N, K = 524288, 1024, 1024
    A = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16(); W = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).bfloat16()
    b = torch.rand(N, device="cuda"); Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    fn = lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=8)
    flop = 2.0 * M * N * K
elif kind == "fmlp":
    M = 32768 * 192
    net = classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="bf16"); net.net._fused_ready()
    pts = torch.randn(M, 3, device="cuda"); vd = torch.nn.functional.normalize(torch.randn(M // 192, 3, device="cuda"), dim=-1)
    out = torch.empty(M, 4, device="cuda")
    fn = lambda: ops.fmlp_classic_pts_fwd(pts, vd, 192, net.net.fstream, net.net.fbias, out)
    flop = 2.0 * 593408 * M
else:
    fn = None
if fn is None:
    time.sleep(secs); print("idle"); sys.exit(0)
for _ in range(5): fn()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
print(f"{kind}: {n} launches in {dt:.2f} s = {n * flop / dt / 1e12:.1f} TFLOP/s sustained")
''' % REPO


def sample():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-c", "-P", "-u", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = next(v for k, v in d.items() if k.startswith("card"))
        pick = lambda sub: next((v for k, v in card.items() if sub in k.lower()), None)
        return {"sclk": pick("sclk clock speed"), "mclk": pick("mclk clock speed"), "power_w": pick("power"), "busy": pick("gpu use")}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    child = subprocess.Popen([sys.executable, "-c", CHILD, kind, str(secs)], stdout=subprocess.PIPE, text=True)
    time.sleep(4.0 if kind != "idle" else 0.0)                     # torch import + warm-up
    rows = []
    while child.poll() is None:
        rows.append(sample())
        time.sleep(0.25)
    print(json.dumps({"kernel": kind, "child": child.stdout.read().strip(), "samples": rows}))


if __name__ == "__main__":
    main()
