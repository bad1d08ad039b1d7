import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snerf_amd import classic
dev = "cuda"; N = 32768
mk = lambda: classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="bf16", device=dev)
coarse, fine = mk(), mk()
fl = os.environ.get("SNERF_CLASSIC_MODE", "fused") == "fused"
coarse.net.fused = fine.net.fused = fl
e, _ = classic.get_embedder(10, 0); ev, _ = classic.get_embedder(4, 0)
q = classic.make_network_query_fn(e, ev, netchunk=1 << 30)
g = torch.Generator().manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
o = torch.randn(N, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 4.0])
rays = torch.cat([o, -d, torch.full((N, 1), 2.0), torch.full((N, 1), 6.0), -d], -1).to(dev)
tgt = torch.rand(N, 3, generator=g).to(dev)
opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
for _ in range(5):
    opt.zero_grad(set_to_none=False)
    r = classic.render_rays(rays, coarse, q, 64, perturb=1.0, N_importance=128, network_fine=fine, white_bkgd=False, raw_noise_std=0.0)
    (((r["rgb_map"] - tgt) ** 2).mean() + ((r["rgb0"] - tgt) ** 2).mean()).backward()
    opt.step(); coarse.arena.bump(); fine.arena.bump()
torch.cuda.synchronize()
