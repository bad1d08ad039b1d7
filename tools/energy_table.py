#!/usr/bin/env python3
"""Energy per TFLOP and sustained shader clock of the MFMA kernels on the headline layer shape: every variant runs
back to back for a few seconds in a child process while the parent samples `rocm-smi -c -P` (socket power, sclk); the first 1.5 s of
samples (ramp) are dropped.  J / TFLOP = mean watts / sustained TFLOP/s.

    python tools/energy_table.py [seconds per variant] > profiles/rN_x_energy_per_tflop.txt

Variants (M = 524288, N = K = 1024 bf16 unless noted): the shipped persistent 8-phase NT kernel (forward + ReLU epilogue, and the
data-gradient flavour with mask + column sums is the same kernel), its non-persistent form, the 256 x 256 block-issue kernel, the
128 x 128 kernel, the 8-phase weight-gradient kernel on dense and on ReLU-sparse operands, the vendor library (torch.mm = hipBLASLt) on
the same shape, and the fused register-resident 8 x 256 network (csrc/fmlp.hip)."""
import json, os, subprocess, sys, time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from snerf_amd import classic, ops
kind, secs = sys.argv[1], float(sys.argv[2])
M, N, K = 524288, 1024, 1024
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1)
if kind.startswith("nt"):
    variant = {"nt8p": 8, "nt8": 4, "nt256": 1, "nt128": 0}[kind]
    A = rnd(M, K).bfloat16(); W = (rnd(N, K) / K ** 0.5).bfloat16(); b = rnd(N); Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    fn = lambda: ops.linear_fwd(A, W, b, Y, K, N, ops.ACT_RELU, ops.BF16, variant=variant)
    flop = 2.0 * M * N * K
elif kind.startswith("tn8"):
    dZ = rnd(M, N); X = rnd(M, K)
    if kind == "tn8_sparse":                      # what the real step feeds it: a ReLU output and a masked gradient (half zeros each)
        dZ = dZ * (rnd(M, N) > 0); X = torch.relu(X)
    dZ, X = dZ.bfloat16(), X.bfloat16(); dW = torch.zeros(N, K, device="cuda")
    fn = lambda: ops.linear_wgrad(dZ, X, dW, N, K, ops.BF16, variant=3)
    flop = 2.0 * M * N * K
elif kind == "vendor":
    A = rnd(M, K).bfloat16(); W = (rnd(K, N) / K ** 0.5).bfloat16(); Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    fn = lambda: torch.mm(A, W, out=Y)
    flop = 2.0 * M * N * K
elif kind == "fmlp":
    M = 32768 * 192
    net = classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="bf16"); net.net._fused_ready()
    pts = torch.randn(M, 3, device="cuda"); vd = torch.nn.functional.normalize(torch.randn(M // 192, 3, device="cuda"), dim=-1)
    out = torch.empty(M, 4, device="cuda")
    fn = lambda: ops.fmlp_classic_pts_fwd(pts, vd, 192, net.net.fstream, net.net.fbias, out)
    flop = 2.0 * 593408 * M
for _ in range(5): fn()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
print(n * flop / dt / 1e12)
''' % REPO


def sample():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=20).stdout
        card = next(v for k, v in json.loads(out).items() if k.startswith("card"))
        pick = lambda sub: next((v for k, v in card.items() if sub in k.lower()), None)
        num = lambda v: float("".join(ch for ch in str(v) if ch.isdigit() or ch == ".") or "nan")
        return num(pick("power")), num(pick("sclk clock speed"))
    except Exception:  # noqa: BLE001
        return None


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    print(f"{'variant':12s} {'TFLOP/s':>9s} {'watts':>8s} {'sclk MHz':>9s} {'J/TFLOP':>8s}   samples")
    for kind in ("nt8p", "nt8", "nt256", "nt128", "tn8_dense", "tn8_sparse", "vendor", "fmlp"):
        child = subprocess.Popen([sys.executable, "-c", CHILD, kind, str(secs)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        t0 = time.time()
        rows = []
        while child.poll() is None:
            s = sample()
            if s is not None:
                rows.append((time.time() - t0, s))
            time.sleep(0.2)
        try:
            tf = float(child.stdout.read().strip().splitlines()[-1])
        except (ValueError, IndexError):
            print(f"{kind:12s} failed"); continue
        end = rows[-1][0] if rows else 0.0
        steady = [s for t, s in rows if t > end - secs + 1.5 and t < end - 0.3] or [s for _, s in rows]
        w = sum(a for a, _ in steady) / max(len(steady), 1)
        clk = sum(b for _, b in steady) / max(len(steady), 1)
        print(f"{kind:12s} {tf:9.1f} {w:8.0f} {clk:9.0f} {w / tf:8.3f}   {len(steady)}")


if __name__ == "__main__":
    main()
