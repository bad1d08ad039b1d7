#!/usr/bin/env python3
"""Long screen for LDS ordering races (round 4): every path's inference pass and deterministic training pass repeated many times with the
LDS scribble kernel (snerf_debug_lds_scribble, a new seed each time) in front of EVERY library call; every repeat must equal the first bit
for bit.  A DMA that is not ordered before its reader (the K = 128 race of gemm_nt8p_kernel fired about once in 300 launches) shows up here
as a count > 0; tests/test_stale_lds.py is the short form.

    python tools/stress_stale_lds.py [repeats]        # default 300 inference / 100 training repeats per path
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd import _lib, ops

REAL = _lib.call
STATE = {"n": 0}


def call(name, *args):
    if name != "snerf_debug_lds_scribble":
        STATE["n"] += 1
        REAL("snerf_debug_lds_scribble", (STATE["n"] * 2654435761) & 0x7fffffff, ops._stream())
    return REAL(name, *args)


_lib.call = call


def screen(name, fn, reps):
    ref = [t.detach().clone() for t in fn()]
    bad, t0 = [], time.time()
    for r in range(reps):
        out = fn()
        for k, (a, b) in enumerate(zip(out, ref)):
            if not torch.equal(a, b):
                bad.append((r, k, int((a != b).sum())))
    print(f"{name}: {len(bad)} differing outputs in {reps} repeats ({time.time() - t0:.1f} s) {bad[:5]}", flush=True)
    return len(bad)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from snerf_amd import classic, mipnerf, zipnerf
    from oracle import common
    total = 0
    torch.manual_seed(0)
    m = mipnerf.MipNerfModel(n_samples=64, N_fine=129, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True, rgb_layer=3,
                             hidden_layer=1024, density_noise=0., max_deg_point=16, proposal_hidden_layer=256, proposal_loss=True, compute="bf16")
    m.set_deterministic(True)
    n = 4096
    rays = mipnerf.Rays(**{k: v.cuda() for k, v in common.synthetic_rays(n, seed=3).items()})
    tgt = torch.rand(n, 3, device="cuda")

    def a_infer():
        with torch.no_grad():
            ret = m(rays, False, False, 0.)
        return [ret[1][0], ret[1][1], ret[1][2], ret[0][1]]

    def a_train():
        for p in m.parameters():
            p.grad = None
        ret = m(rays, False, False, 0.)
        (((ret[1][0] - tgt) ** 2).mean() + 0.05 * (1 / ret[0][1]).mean()).backward()
        return [ret[1][0]] + [p.grad for p in m.parameters() if p.grad is not None]
    total += screen("path A inference (4096 rays)", a_infer, reps)
    total += screen("path A deterministic train pass", a_train, max(reps // 3, 1))
    del m
    net = classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="bf16", device="cuda")
    net.set_deterministic(True)
    M, S = 4096 * 64, 64
    pts = torch.rand(M, 3, device="cuda") * 4 - 2
    vd = torch.nn.functional.normalize(torch.randn(M // S, 3, device="cuda"), dim=-1)
    nn_ = net.net

    def b_infer():
        with torch.no_grad():
            return [nn_.forward(pts, vd, S, False)[0]]

    def b_train():
        raw, saved = nn_.forward(pts, vd, S, True)
        nn_.a.grad.zero_()
        nn_.backward(torch.ones_like(raw) * 1e-3, saved)
        return [raw, nn_.a.grad]
    total += screen("path B inference (262 144 samples)", b_infer, reps)
    total += screen("path B deterministic train pass", b_train, max(reps // 3, 1))
    del net
    R = 8192
    g = torch.Generator().manual_seed(4)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.tensor([0.0, 1.0, 0.0]).expand(R, 3), dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    batch = {k: v.cuda() for k, v in dict(origins=torch.randn(R, 3, generator=g) * 0.05, directions=d, viewdirs=d, radii=torch.full((R, 1), 5e-4),
                                          near=torch.full((R, 1), 0.1), far=torch.full((R, 1), 10.0), base_x=bx, base_y=by).items()}
    tgt = torch.rand(R, 3, generator=g).cuda()
    zm = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute="fp16", table_dtype="ref",
                       grid_log2_hashmap_size=19, init_std=0.1)
    for net in zm.nets:
        net.deterministic = True
    draws = zm._draws(R, False, zm.arena.flat.device, 7)

    def c_infer():
        with torch.no_grad():
            ren, hist = zm(False, batch, 1.0, False, draws=draws)
        return [ren[2]["rgb"], ren[2]["depth"], hist[0]["weights"], hist[1]["weights"]]

    def c_train():
        for p in zm.parameters():
            p.grad = None
        ren, hist = zm(False, batch, 1.0, False, draws=draws)
        (((ren[2]["rgb"] - tgt) ** 2).mean() + 0.05 * sum((h["weights"] ** 2).sum() for h in hist[:2]) / R).backward()
        return [ren[2]["rgb"]] + [p.grad for p in zm.parameters() if p.grad is not None]
    total += screen("path C inference (8192 rays)", c_infer, reps)
    total += screen("path C deterministic train pass", c_train, max(reps // 3, 1))
    # the fused training forward + gradient chain of path C's NeRF MLP (not used in the deterministic mode above): everything they STORE
    from snerf_amd.mlp import ParamArena, ZipNerfNet
    for compute in ("fp16", "bf16"):
        torch.manual_seed(3)
        dev = torch.device("cuda")
        shapes = [("n." + k, sh) for k, sh in ZipNerfNet.param_shapes(40)]
        arena = ParamArena(shapes, dev)
        arena.load({k: (torch.randn(sh) * (1.2 / sh[-1] ** 0.5) if len(sh) == 2 else torch.randn(sh) * 0.1) for k, sh in shapes})
        net2 = ZipNerfNet(arena, "n.", ops.F16 if compute == "fp16" else ops.BF16, 40)
        M2 = 256 * 1100 + 77
        Fb0 = torch.zeros(M2, 64, device=dev); Fb0[:, :40] = torch.randn(M2, 40, device=dev) * 0.5
        Dn = torch.zeros(M2, 16, device=dev); Dn[:, :9] = torch.randn(M2, 9, device=dev)
        d_rgb = torch.randn(M2, 3, device=dev) * 1e-2
        d_den = torch.randn(M2, 20, device=dev) * 1e-2

        def fz_step():
            Fb, SB = net2.alloc(M2)
            Fb.copy_(Fb0.to(net2.tdt)); SB[:, 512:] = 0; SB[:, 512:528] = Dn.to(net2.tdt)
            raw_rgb, raw_d, saved = net2.forward(Fb, SB, True)
            arena.grad.zero_()
            dF = net2.backward(d_rgb, d_den, saved)
            return [raw_rgb, raw_d, saved[1], saved[2], saved[3], dF] + list(net2._zip_bits[0])
        total += screen(f"path C fused MLP forward + gradient chain, {compute} (281 677 rows)", fz_step, reps)
    print("TOTAL differing outputs:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
