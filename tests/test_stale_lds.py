"""Kernels must not read LDS they have not written in the same launch.  LDS content survives from one kernel to the next on a CU, so a read
of a location the launch never wrote (or wrote later) normally sees what the PREVIOUS launch of the same kernel left there -- the right
values -- and every repeat-and-compare test passes.  Here `snerf_debug_lds_scribble` fills every CU's LDS with a seeded random pattern in
front of EVERY library call of a whole forward / backward pass of each path, twice with different seeds: results that depend on the seed
are such reads.  This catches a missing initialisation every time; a DMA that is merely not ORDERED before its reader (round 4: the K = 128
flavour of gemm_nt8p_kernel, about one launch in 300) only when it fires -- that one has its own repetition test
(tests/test_gpu_kernels.py::test_persistent_gemm_two_k_tiles_after_another_layer_is_reproducible, tools/probes/gemm_k128_bias_race.py);
with the scribble in front such a read at least cannot hide behind equal leftovers."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def scribbled(monkeypatch):
    from snerf_amd import _lib, ops
    real = _lib.call
    state = {"base": 1, "n": 0, "on": True}

    def call(name, *args):
        if state["on"] and name != "snerf_debug_lds_scribble":
            state["n"] += 1
            real("snerf_debug_lds_scribble", (state["base"] * 1000003 + state["n"] * 7919) & 0x7fffffff, ops._stream())
        return real(name, *args)
    monkeypatch.setattr(_lib, "call", call)
    return state


def _twice(state, fn):
    outs = []
    for base in (1, 2):
        state["base"], state["n"] = base, 0
        outs.append([t.detach().clone() for t in fn()])
        assert state["n"] > 0
    for k, (a, b) in enumerate(zip(*outs)):
        same = torch.equal(a, b) or (bool(torch.isnan(a).any()) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) and torch.equal(torch.isnan(a), torch.isnan(b)))
        assert same, f"output {k} depends on what the LDS held before the launches: {int((a != b).sum())} of {a.numel()} elements differ"
    return outs[0]


def test_path_a_forward_and_backward_do_not_read_stale_lds(scribbled):
    from snerf_amd import mipnerf
    from oracle import common
    torch.manual_seed(0)
    m = mipnerf.MipNerfModel(n_samples=64, N_fine=129, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True, rgb_layer=3,
                             hidden_layer=1024, density_noise=0., max_deg_point=16, proposal_hidden_layer=256, proposal_loss=True, compute="bf16")
    m.set_deterministic(True)
    n = 1500                                                          # 192 000 fine rows: not a multiple of the 256-row tile
    rays = mipnerf.Rays(**{k: v.cuda() for k, v in common.synthetic_rays(n, seed=3).items()})
    tgt = torch.rand(n, 3, device="cuda")

    def infer():
        with torch.no_grad():
            ret = m(rays, False, False, 0.)
        return [ret[1][0], ret[1][1], ret[1][2], ret[0][1]]

    def train():
        for p in m.parameters():
            p.grad = None
        ret = m(rays, False, False, 0.)
        (((ret[1][0] - tgt) ** 2).mean() + 0.05 * (1 / ret[0][1]).mean()).backward()
        return [ret[1][0]] + [p.grad for p in m.parameters() if p.grad is not None]
    _twice(scribbled, infer)
    _twice(scribbled, train)


def test_path_b_forward_and_backward_do_not_read_stale_lds(scribbled):
    from snerf_amd import classic
    torch.manual_seed(0)
    net = classic.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True, compute="bf16", device="cuda")
    net.set_deterministic(True)
    M, S = 40 * 96, 96
    pts = torch.rand(M, 3, device="cuda") * 4 - 2
    vd = torch.nn.functional.normalize(torch.randn(M // S, 3, device="cuda"), dim=-1)
    n = net.net

    def infer():
        with torch.no_grad():
            return [n.forward(pts, vd, S, False)[0]]

    def train():
        raw, saved = n.forward(pts, vd, S, True)
        n.a.grad.zero_()
        n.backward(torch.ones_like(raw) * 1e-3, saved)
        return [raw, n.a.grad]
    _twice(scribbled, infer)
    scribbled["on"] = False
    n.deterministic = False                                            # the fused gradient chain (its bias gradients add in arrival order: compared to rounding)
    scribbled["on"] = True
    outs = []
    for base in (1, 2):
        scribbled["base"], scribbled["n"] = base, 0
        raw, saved = n.forward(pts, vd, S, True)
        n.a.grad.zero_()
        n.backward(torch.ones_like(raw) * 1e-3, saved)
        outs.append((raw.clone(), n.a.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    rel = float((outs[0][1] - outs[1][1]).norm() / outs[0][1].norm())
    assert rel < 1e-5, rel
    n.deterministic = True
    _twice(scribbled, train)


def test_path_c_forward_and_backward_do_not_read_stale_lds(scribbled):
    from snerf_amd import zipnerf
    torch.manual_seed(0)
    R = 1000
    g = torch.Generator().manual_seed(4)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.tensor([0.0, 1.0, 0.0]).expand(R, 3), dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    batch = {k: v.cuda() for k, v in dict(origins=torch.randn(R, 3, generator=g) * 0.05, directions=d, viewdirs=d, radii=torch.full((R, 1), 5e-4),
                                          near=torch.full((R, 1), 0.1), far=torch.full((R, 1), 10.0), base_x=bx, base_y=by).items()}
    tgt = torch.rand(R, 3, generator=g).cuda()
    for compute, table in (("fp16", "f16"), ("bf16", "ref")):
        m = zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute=compute, table_dtype=table,
                          grid_log2_hashmap_size=16, init_std=0.1, use_semantic=(compute == "fp16"))     # (without the semantic head inference takes the fused NeRF MLP)
        for net in m.nets:
            net.deterministic = True
        draws = m._draws(R, False, m.arena.flat.device, 7)

        def infer():
            with torch.no_grad():
                ren, hist = m(False, batch, 1.0, False, draws=draws)
            return [ren[2]["rgb"], ren[2]["depth"], hist[0]["weights"], hist[1]["weights"]] + ([ren[2]["semantic"]] if m.use_semantic else [])

        def train():
            for p in m.parameters():
                p.grad = None
            ren, hist = m(False, batch, 1.0, False, draws=draws)
            (((ren[2]["rgb"] - tgt) ** 2).mean() + 0.05 * sum((h["weights"] ** 2).sum() for h in hist[:2]) / R).backward()
            return [ren[2]["rgb"]] + [p.grad for p in m.parameters() if p.grad is not None]
        _twice(scribbled, infer)
        _twice(scribbled, train)
