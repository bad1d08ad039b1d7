#!/usr/bin/env python3
"""Path B (classic render_rays: stratified 64 -> PE-63/27 -> NeRF 8x256 -> raw2outputs -> sample_pdf 128 -> sort -> fine NeRF)
throughput on one MI355X: forward+backward (autograd through the drop-in operators) and forward-only rays/s.
Algorithmic MLP work: 2 * 593 408 * (64 + 192) = 303.8 MFLOP/ray forward."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=32768)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--compute", default="bf16")
    ap.add_argument("--train-only", action="store_true", help="just the fused training step (rocprofv3 target)")
    args = ap.parse_args()
    from snerf_amd import classic
    torch.manual_seed(0)
    dev = "cuda"
    mk = lambda: classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute=args.compute, device=dev)
    coarse, fine = mk(), mk()
    embed_fn, _ = classic.get_embedder(10, 0)
    embeddirs_fn, _ = classic.get_embedder(4, 0)
    q = classic.make_network_query_fn(embed_fn, embeddirs_fn, netchunk=1 << 30)
    N = args.rays
    g = torch.Generator().manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    o = torch.randn(N, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 4.0])
    rays = torch.cat([o, -d * 1.0, torch.full((N, 1), 2.0), torch.full((N, 1), 6.0), -d], -1).to(dev)
    tgt = torch.rand(N, 3, generator=g).to(dev)
    opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)

    def fwd():
        return classic.render_rays(rays, coarse, q, 64, perturb=1.0, N_importance=128, network_fine=fine, white_bkgd=False, raw_noise_std=0.0)

    def train():
        opt.zero_grad(set_to_none=False)
        r = fwd()
        loss = ((r["rgb_map"] - tgt) ** 2).mean() + ((r["rgb0"] - tgt) ** 2).mean()
        loss.backward()
        opt.step()
        coarse.arena.bump(); fine.arena.bump()

    def timeit(fn):
        fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    dt_train = timeit(train)
    if args.train_only:
        print(json.dumps({"train_ms": round(dt_train * 1e3, 3)}))
        return
    fine.net.fused = coarse.net.fused = False         # A/B: the training forward as one GEMM launch per layer
    dt_train_layered = timeit(train)
    fine.net.fused = coarse.net.fused = True
    fine.net.fused_chain = coarse.net.fused_chain = False     # A/B: the data gradients as one GEMM launch per layer
    dt_train_dgrad_layered = timeit(train)
    fine.net.fused_chain = coarse.net.fused_chain = True
    flops = 2.0 * 593408 * (64 + 192)
    with torch.no_grad():
        dt_fwd = timeit(fwd)
        # A/B of the inference MLP: the fused register-resident kernel (csrc/fmlp.hip) vs one GEMM launch per layer, the network alone
        M = N * 192
        pts = torch.randn(N, 192, 3, device=dev)
        vd = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
        mlp_only = {}
        for name, flag in (("fused", True), ("per_layer", False)):
            fine.net.fused = coarse.net.fused = flag
            dt = timeit(lambda: classic.run_network(pts, vd, fine, embed_fn, embeddirs_fn))
            mlp_only[name] = {"ms": round(dt * 1e3, 3), "TFLOPs": round(M * 2.0 * 593408 / dt / 1e12, 1)}
            if not flag:
                dt_fwd_layered = timeit(fwd)
        fine.net.fused = coarse.net.fused = True
    print(json.dumps({"path": "B (classic render_rays, 64 coarse + 192 fine evals/ray, NeRF 8x256 x2)", "rays": N, "compute": args.compute,
                      "train_ms": round(dt_train * 1e3, 3), "train_rays_per_s": round(N / dt_train, 1), "fwd_ms": round(dt_fwd * 1e3, 3),
                      "fwd_rays_per_s": round(N / dt_fwd, 1), "fwd_mlp_TFLOPs": round(N * flops / dt_fwd / 1e12, 1),
                      "train_mlp_TFLOPs": round(3 * N * flops / dt_train / 1e12, 1),
                      "train_per_layer_fwd_ms": round(dt_train_layered * 1e3, 3),
                      "train_per_layer_dgrad_ms": round(dt_train_dgrad_layered * 1e3, 3),
                      "frame_1600x900_s": round(1440000 / (N / dt_fwd), 3),
                      "fwd_per_layer_ms": round(dt_fwd_layered * 1e3, 3), "fwd_per_layer_mlp_TFLOPs": round(N * flops / dt_fwd_layered / 1e12, 1),
                      "run_network_only": {"rows": M, **mlp_only,
                                           "note": "embedding kernel + NeRF 8x256 on N x 192 samples; TFLOPs = algorithmic 2 x 593 408 FLOP per sample"}}))


if __name__ == "__main__":
    main()
