"""N > 1 path on CPU: two gloo ranks, ray batch sharded across them, ONE all-reduce of the flat gradient arena, fused
Adam with the 1/world mean folded in.  After two steps the parameters must equal a single-process run on the whole
batch (host logic; kernels emulated -- tests/cpu_ops_emulation.py)."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _build(n_samples=8, n_fine=9, hidden=64, compute="f32"):
    from snerf_amd.mipnerf import MipNerfModel
    torch.manual_seed(0)
    return MipNerfModel(n_samples=n_samples, N_fine=n_fine, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0,
                        real=True, rgb_layer=3, hidden_layer=hidden, density_noise=0., max_deg_point=16, proposal_hidden_layer=64,
                        proposal_loss=True, compute=compute, device="cpu")


def _data(n):
    from oracle import common
    from snerf_amd.mipnerf import Rays
    rays = Rays(**common.synthetic_rays(n, seed=9))
    tgt = torch.rand(n, 3, generator=torch.Generator().manual_seed(10))
    return rays, tgt


def _worker(rank, world, init_file, n, out_file, compute="f32"):
    from cpu_ops_emulation import emulate_ops
    from snerf_amd.trainer import MipTrainer, shard_rays
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(2)
    with emulate_ops():
        model = _build(compute=compute)
        if rank != 0:
            with torch.no_grad():
                model.arena.flat.add_(1.0)          # deliberately different: broadcast must fix it
        tr = MipTrainer(model, lr=1e-2)
        tr.broadcast_parameters(0)
        rays, tgt = _data(n)
        my = shard_rays(rays, rank, world)
        per = n // world
        for _ in range(2):
            tr.step(my, tgt[rank * per:(rank + 1) * per], randomized=False)
        flat = model.arena.flat.clone()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        if rank == 0:
            torch.save(gathered, out_file)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("compute,tol", [("f32", 2e-5), ("f16f8", 1e-3), ("bf16x3_fwd", 1e-3)])
def test_ray_sharded_data_parallel_matches_single_process(compute, tol):
    """compute="f16f8" / "bf16x3_fwd": the modes whose backward differs from their forward's layout (scaled fp16 / plain bf16 gradients through a
    scratch arena, one exchange announcement per network): the ranks must still agree bit for bit and follow the single-process run."""
    from cpu_ops_emulation import emulate_ops
    from snerf_amd.trainer import MipTrainer
    n, world = 24, 2
    with tempfile.TemporaryDirectory() as td:
        init_file, out_file = os.path.join(td, "init"), os.path.join(td, "out.pt")
        mp.spawn(_worker, args=(world, init_file, n, out_file, compute), nprocs=world, join=True)
        gathered = torch.load(out_file)
    assert torch.equal(gathered[0], gathered[1]), "ranks diverged"
    with emulate_ops():
        model = _build(compute=compute)
        tr = MipTrainer(model, lr=1e-2)
        rays, tgt = _data(n)
        for _ in range(2):
            tr.step(rays, tgt, randomized=False)
    ref = model.arena.flat
    err = (gathered[0] - ref).abs().max().item()
    assert err < tol, f"data-parallel parameters differ from the single-process run by {err:.3e}"
    assert (ref - _build().arena.flat).abs().max().item() > 1e-3, "the optimiser did not move the parameters"


# ---------------------------------------------------------------------------- path C (zipnerf) ----
def _zip_build(log2T=12):
    from snerf_amd import zipnerf
    torch.manual_seed(0)
    return zipnerf.Model(config=None, raydist_fn='power_transformation', opaque_background=True, compute="f32", table_dtype="f32", device="cpu",
                         grid_log2_hashmap_size=log2T, init_std=0.1)


def _zip_data(n):
    g = torch.Generator().manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    bx = torch.nn.functional.normalize(torch.cross(d, torch.tensor([0.0, 1.0, 0.0]).expand(n, 3), dim=-1), dim=-1)
    by = torch.nn.functional.normalize(torch.cross(d, bx, dim=-1), dim=-1)
    batch = dict(origins=torch.randn(n, 3, generator=g) * 0.05, directions=d, viewdirs=d, radii=torch.full((n, 1), 5e-4),
                 near=torch.full((n, 1), 0.1), far=torch.full((n, 1), 10.0), base_x=bx, base_y=by)
    return batch, torch.rand(n, 3, generator=g)


def _zip_aux(hist):
    # stand-in for the caller's interlevel loss: any differentiable function of the levels' weights
    return sum((h["weights"] ** 2).sum() for h in hist) * 1e-3


def _zip_worker(rank, world, init_file, n, out_file, table_exchange="sharded", log2T=12):
    from cpu_ops_emulation import emulate_ops
    from snerf_amd.trainer import ZipTrainer
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(2)
    with emulate_ops():
        model = _zip_build(log2T)
        if rank != 0:
            with torch.no_grad():
                model.arena.flat.add_(0.5)
        tr = ZipTrainer(model, lr=1e-2, eps=1e-4, table_exchange=table_exchange)    # a large eps: with the reference's 1e-15 Adam is a sign function of rounding noise on near-zero gradients
        assert (tr.shards is not None) == (table_exchange == "sharded")
        tr.broadcast_parameters(0)
        batch, tgt = _zip_data(n)
        per = n // world
        mine = {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
        for _ in range(2):
            # the per-rank aux loss is a SUM over its rays: scale by world so that the all-reduced mean equals the full-batch sum
            tr.step(mine, tgt[rank * per:(rank + 1) * per], rand=False, aux_loss_fn=lambda h: _zip_aux(h) * world)
        flat = model.arena.flat.clone()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        sd = tr.state_dict()                        # (collective with sharded tables: the other ranks' moment slices are gathered)
        moments = torch.cat([torch.cat([st["exp_avg"].reshape(-1), st["exp_avg_sq"].reshape(-1)]) for st in sd["state"].values()])
        all_m = [torch.empty_like(moments) for _ in range(world)]
        dist.all_gather(all_m, moments)
        if rank == 0:
            torch.save((gathered, all_m), out_file)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_zip_ray_sharded_data_parallel_matches_single_process():
    from cpu_ops_emulation import emulate_ops
    from snerf_amd.trainer import ZipTrainer
    n, world = 8, 2
    with tempfile.TemporaryDirectory() as td:
        init_file, out_file = os.path.join(td, "init"), os.path.join(td, "out.pt")
        mp.spawn(_zip_worker, args=(world, init_file, n, out_file), nprocs=world, join=True)
        gathered, moments = torch.load(out_file)
        # the hash tables through reduce-scatter + sharded Adam + all-gather (default) and through the dense all-reduce: the same sums
        # (two addends commute), the same optimiser arithmetic on every element -- identical parameters
        init2, out2 = os.path.join(td, "init2"), os.path.join(td, "out2.pt")
        mp.spawn(_zip_worker, args=(world, init2, n, out2, "allreduce"), nprocs=world, join=True)
        dense, dense_moments = torch.load(out2)
    # a checkpoint of the sharded run holds every rank's Adam moments (gathered), the same on both ranks and the same as the dense run's
    assert torch.equal(moments[0], moments[1]) and torch.equal(moments[0], dense_moments[0]) and float(moments[0].abs().sum()) > 0
    assert torch.equal(gathered[0], gathered[1]), "ranks diverged"
    assert torch.equal(gathered[0], dense[0]) and torch.equal(dense[0], dense[1]), "sharded and all-reduced table updates differ"
    with emulate_ops():
        model = _zip_build()
        start = model.arena.flat.clone()
        tr = ZipTrainer(model, lr=1e-2, eps=1e-4)    # a large eps: with the reference's 1e-15 Adam is a sign function of rounding noise on near-zero gradients
        batch, tgt = _zip_data(n)
        for _ in range(2):
            tr.step(batch, tgt, rand=False, aux_loss_fn=_zip_aux)
    err = (gathered[0] - model.arena.flat).abs().max().item()
    assert err < 5e-5, f"data-parallel parameters differ from the single-process run by {err:.3e}"
    assert (model.arena.flat - start).abs().max().item() > 1e-3


def _zip_dynamic_scale_worker(rank, world, init_file, n, out_file):
    from cpu_ops_emulation import emulate_ops
    from snerf_amd.trainer import ZipTrainer, LossScaler
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(2)
    with emulate_ops():
        model = _zip_build()
        tr = ZipTrainer(model, lr=1e-2, eps=1e-4, loss_scale=LossScaler(init_scale=256.0, growth_interval=2))
        assert tr.shards is not None
        tr.broadcast_parameters(0)
        batch, tgt = _zip_data(n)
        per = n // world
        mine = {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
        log = []
        for i in range(4):
            # step 1: ONE rank's gradients overflow (its aux term is infinite) -- every rank has to skip
            mult = float("inf") if (i == 1 and rank == 1) else float(world)
            before = model.arena.flat.clone()
            tr.step(mine, tgt[rank * per:(rank + 1) * per], rand=False, aux_loss_fn=lambda h: _zip_aux(h) * mult)
            log.append((bool(torch.equal(before, model.arena.flat)), tr.t, tr.scaler.scale, tr.scaler.skipped_steps, bool(tr.model.arena.grad.any())))
        flat = model.arena.flat.clone()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        logs = [None] * world
        dist.all_gather_object(logs, log)
        if rank == 0:
            torch.save((gathered, logs), out_file)
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_zip_sharded_tables_on_four_ranks_with_slices_off_the_16_byte_grid():
    """at world 4 / 8 a rank's slice of a single-channel table is not a multiple of 4 floats (the default proposal tables:
    6 606 952 / 8 = 825 869 rows per rank) -- the sharded update (reduce-scatter, Adam on the slice, all-gather) has to agree with
    the dense all-reduce there too.  Four gloo ranks on a small model whose C = 1 tables are cut off the 16-byte grid (the HIP Adam
    kernel takes such slices since round 5: tests/test_gpu_kernels.py::test_adam_on_misaligned_slices_matches_aligned)."""
    from cpu_ops_emulation import emulate_ops
    n, world = 8, 4
    with emulate_ops():
        m = _zip_build(13)            # 2^13-row levels: the C = 1 tables hold 45 880 / 62 264 rows = 11 470 / 15 566 per rank (2 mod 4)
        spans = [m.arena.span(nm + "encoder.embeddings") for nm in m.names]
        per = [(b - a) // world for a, b in spans]
        assert all((b - a) % world == 0 for a, b in spans) and any(p % 4 != 0 or (a % 4) != 0 for p, (a, b) in zip(per, spans)), (spans, per)
    with tempfile.TemporaryDirectory() as td:
        init_file, out_file = os.path.join(td, "init"), os.path.join(td, "out.pt")
        mp.spawn(_zip_worker, args=(world, init_file, n, out_file, "sharded", 13), nprocs=world, join=True)
        gathered, moments = torch.load(out_file)
        init2, out2 = os.path.join(td, "init2"), os.path.join(td, "out2.pt")
        mp.spawn(_zip_worker, args=(world, init2, n, out2, "allreduce", 13), nprocs=world, join=True)
        dense, dense_moments = torch.load(out2)
    for r in range(1, world):
        assert torch.equal(gathered[0], gathered[r]), "ranks diverged"
    # four addends: the sharded path sums them in the reduce's order, the dense path in the all-reduce's -- equal up to fp32 rounding
    assert float((gathered[0] - dense[0]).abs().max()) < 2e-6 and float((moments[0] - dense_moments[0]).abs().max()) < 1e-6


@pytest.mark.timeout(600)
def test_zip_dynamic_loss_scale_skips_on_every_rank_when_one_rank_overflows():
    """ZipTrainer(loss_scale=LossScaler(...)) with sharded table updates on two ranks: an overflow in ONE rank's gradients (step 1) is
    seen by both (the all-reduced sums are non-finite / the 4-byte MAX all-reduce of the flag) -- both skip the step whole, halve the
    scale, keep t; the other steps update, the scale grows after 2 clean steps, and the ranks stay bit-identical."""
    n, world = 8, 2
    with tempfile.TemporaryDirectory() as td:
        init_file, out_file = os.path.join(td, "init"), os.path.join(td, "out.pt")
        mp.spawn(_zip_dynamic_scale_worker, args=(world, init_file, n, out_file), nprocs=world, join=True)
        gathered, logs = torch.load(out_file, weights_only=False)
    assert torch.equal(gathered[0], gathered[1]), "ranks diverged"
    assert bool(torch.isfinite(gathered[0]).all())
    assert logs[0] == logs[1]
    #            unchanged, t, scale, skipped, grads left
    assert logs[0] == [(False, 1, 256.0, 0, False), (True, 1, 128.0, 1, False), (False, 2, 128.0, 1, False), (False, 3, 256.0, 1, False)], logs[0]


def _zip_render_worker(rank, world, init_file, H, W, out_file):
    import types
    from cpu_ops_emulation import emulate_ops
    from snerf_amd import zipnerf
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(2)
    with emulate_ops():
        model = _zip_build()
        batch, _ = _zip_data(H * W)
        frame = {k: v.reshape(H, W, -1) for k, v in batch.items()}
        cfg = types.SimpleNamespace(render_chunk_size=4, vis_num_rays=2)
        model.config = cfg
        out = zipnerf.render_image(lambda rand, b: model(rand, b, train_frac=1.0, compute_extras=True), None, frame, False, cfg)
        if rank == 0:
            torch.save({k: v for k, v in out.items()}, out_file)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_zip_render_image_row_blocks_across_ranks_match_single_process():
    """render_image on 2 ranks (contiguous ray blocks, one all-gather per buffer; 15 rays -> blocks of 8 and 7, ragged chunks of 4)
    assembles the same frame as one process."""
    import types
    from cpu_ops_emulation import emulate_ops
    from snerf_amd import zipnerf
    H, W, world = 3, 5, 2
    with tempfile.TemporaryDirectory() as td:
        init_file, out_file = os.path.join(td, "init"), os.path.join(td, "out.pt")
        mp.spawn(_zip_render_worker, args=(world, init_file, H, W, out_file), nprocs=world, join=True)
        got = torch.load(out_file)
    with emulate_ops():
        model = _zip_build()
        batch, _ = _zip_data(H * W)
        cfg = types.SimpleNamespace(render_chunk_size=1 << 20, vis_num_rays=2)
        model.config = cfg
        ref = zipnerf.render_image(lambda rand, b: model(rand, b, train_frac=1.0, compute_extras=True), None,
                                   {k: v.reshape(H, W, -1) for k, v in batch.items()}, False, cfg)
    assert set(got) == set(ref)
    for k in ("rgb", "depth", "acc", "distance_mean", "distance_median"):
        assert got[k].shape == ref[k].shape and got[k].shape[:2] == (H, W)
        assert torch.allclose(got[k], ref[k], rtol=1e-4, atol=1e-5), (k, float((got[k] - ref[k]).abs().max()))
    assert len(got["ray_weights"]) == 3 and got["ray_weights"][2].shape == (2, 32)


def _render_worker(rank, world, init_file, out_file):
    from cpu_ops_emulation import emulate_ops
    from snerf_amd import mipnerf
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(2)
    with emulate_ops(), torch.no_grad():
        model = _build()
        rays, _ = _data(35)                               # 5 x 7 frame: 35 rays do not divide by 2 (17 + 18)
        grid = mipnerf.Rays(*[r.reshape(5, 7, -1) for r in rays])
        out = mipnerf.render_image(lambda r: model(r, False, False, 0.), grid, rank, chunk=4, world=world)
        if rank == 1:                                     # every rank holds the whole frame after the all-gather
            torch.save([o for o in out[:3]], out_file)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_mip_render_image_sharded_across_ranks():
    """SURVEY.md section 8e, inference: the frame's rays are split into one contiguous block per rank (uneven split), one all-gather
    per output buffer; the assembled frame equals the single-process render."""
    from cpu_ops_emulation import emulate_ops
    from snerf_amd import mipnerf
    with tempfile.TemporaryDirectory() as td:
        init_file, out_file = os.path.join(td, "init"), os.path.join(td, "out.pt")
        mp.spawn(_render_worker, args=(2, init_file, out_file), nprocs=2, join=True)
        got = torch.load(out_file)
    with emulate_ops(), torch.no_grad():
        model = _build()
        rays, _ = _data(35)
        grid = mipnerf.Rays(*[r.reshape(5, 7, -1) for r in rays])
        ref = mipnerf.render_image(lambda r: model(r, False, False, 0.), grid, 0, chunk=35)
    for a, b in zip(got, ref[:3]):                        # (CPU BLAS blocks differently for different chunk sizes: last-bit differences)
        assert a.shape == b.shape and float((a - b).abs().max()) < 1e-5 * (1 + float(b.abs().max()))


_RCCL_ONE_RANK = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SNERF_REPO"])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["SNERF_PORT"], RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from snerf_amd import mipnerf
from snerf_amd.trainer import MipTrainer
from oracle import common

def run(single):
    torch.manual_seed(0)
    m = mipnerf.MipNerfModel(n_samples=32, N_fine=33, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True, rgb_layer=3,
                             hidden_layer=256, density_noise=0., max_deg_point=16, proposal_loss=True, compute="bf16")
    m.set_deterministic(True)      # weight gradients folded in a fixed order: the two runs can be compared exactly
    tr = MipTrainer(m, lr=1e-3, proposal_loss=True, exchange_when_single=single)
    tr.broadcast_parameters()
    rays = mipnerf.Rays(**{k: v.cuda() for k, v in common.synthetic_rays(1024, seed=3).items()})
    tgt = torch.rand(1024, 3, generator=torch.Generator().manual_seed(1)).cuda()
    out = []
    for _ in range(3):
        loss, _ = tr.step(rays, tgt, randomized=False)
        out.append(float(loss))
    return out, m.arena.flat.clone()

la, pa = run(True)          # gradients go through RCCL all_reduce (async, per network block) on this GPU
lb, pb = run(False)         # no collective
assert max(abs(a - b) for a, b in zip(la, lb)) < 1e-6 and float((pa - pb).norm() / pb.norm()) < 1e-6, (la, lb, float((pa - pb).norm() / pb.norm()))
x = torch.arange(8, device="cuda", dtype=torch.float32)
parts = [torch.empty_like(x)]
dist.all_gather(parts, x)
assert torch.equal(parts[0], x)
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", la)
"""


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_collectives_run_on_the_gpu_with_one_rank(tmp_path):
    """The N > 1 tests above run gloo on CPU tensors; the 1-GPU box cannot host two RCCL ranks (duplicate device).  What it can do is
    put the SAME code path -- nccl backend (= RCCL), asynchronous per-block all_reduce of the gradient arena from the backward's
    callbacks, wait before Adam, all_gather -- on real device memory with a communicator of one rank, and check that the training
    trajectory is the one of the run without collectives."""
    import subprocess
    import sys
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_ONE_RANK)
    env = dict(os.environ, SNERF_REPO=REPO, SNERF_PORT=str(29650 + os.getpid() % 200))
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=550, env=env)
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


_CAPTURED_TWO_RANKS = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SNERF_REPO"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["SNERF_PORT"], rank=rank, world_size=world)
torch.cuda.set_device(0)
import bench
from snerf_amd.trainer import MipTrainer, shard_batch

def build():
    torch.manual_seed(0)
    from snerf_amd.mipnerf import MipNerfModel
    return MipNerfModel(n_samples=16, N_fine=17, no_warp_sample=0, ray_shape="cone", fn=1, radius=3., transform_idx=0, real=True, rgb_layer=3,
                        hidden_layer=256, density_noise=0., max_deg_point=16, proposal_hidden_layer=256, proposal_loss=True, compute="bf16", device="cuda")
n = 256
rays = bench.synth_rays(n, 5, "cuda")
g = torch.Generator().manual_seed(6)
tgt = torch.rand(n, 3, generator=g).cuda()
depth = (torch.rand(n, generator=g) * 50 + 2).cuda()
conf = torch.rand(n, generator=g).cuda()
my, mt, md, mc = shard_batch(rays, rank, world, tgt, depth, conf)

def run(captured):
    m = build()
    tr = MipTrainer(m, lr=1e-3)
    tr.broadcast_parameters(0)
    losses = []
    if captured:
        tr.capture(my, mt, md, mc, randomized=False, warmup=2)
        for _ in range(3):
            losses.append(float(tr.replay()[0]))
    else:
        for _ in range(3):
            losses.append(float(tr.step(my, mt, md, mc, randomized=False)[0]))
    return losses, m.arena.flat.clone(), tr.t
le, pe, te = run(False)       # eager: per-layer buckets overlapped with the backward
lc, pc, tc = run(True)        # captured forward + backward, exchange + Adam outside the graph
rel = float((pe - pc).norm() / pe.norm())
assert te == tc == 3 and max(abs(a - b) for a, b in zip(le, lc)) < 2e-3 * max(abs(x) for x in le) and rel < 2e-3, (le, lc, rel)
other = [torch.empty_like(pc) for _ in range(world)]
dist.all_gather(other, pc)
assert torch.equal(other[0], other[1]), "ranks diverged"
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
if rank == 0:
    print("CAPTURED_TWO_RANKS_OK", le, lc, rel)
"""


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_graph_captured_step_with_the_exchange_outside_the_graph_two_ranks(tmp_path):
    """Strong-scaling step: forward + loss tail + backward of a rank's shard as ONE hipGraph launch, the gradient
    all-reduce and Adam outside it.  Two ranks (gloo; both on cuda:0 -- the 1-GPU box cannot host two RCCL ranks) against the eager
    data-parallel step with its per-layer buckets: same trajectory (bf16 atomics order aside), identical parameters on both ranks."""
    import subprocess
    import sys
    script = tmp_path / "captured_two_ranks.py"
    script.write_text(_CAPTURED_TWO_RANKS)
    port = str(29850 + os.getpid() % 100)
    procs = []
    for r in range(2):
        env = dict(os.environ, SNERF_REPO=REPO, SNERF_PORT=port, RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=800) for p in procs]
    assert all(p.returncode == 0 for p in procs) and "CAPTURED_TWO_RANKS_OK" in outs[0][0], [(o[0][-800:], o[1][-2500:]) for o in outs]


def test_grad_exchange_sends_every_element_exactly_once(monkeypatch):
    """_GradExchange with queries that only partly overlap what is already on the wire: the pieces that go out are the
    interval difference [a, b) minus the union of the sent spans, so no element of the gradient arena is all-reduced (summed) twice and
    finish() covers the rest -- every element exactly once whatever the order of the announcements."""
    import types
    from snerf_amd import trainer

    class _Work:
        def wait(self):
            pass

    sent = []

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        sent.append((t.storage_offset(), t.storage_offset() + t.numel()))
        return _Work()
    monkeypatch.setattr(trainer.dist, "all_reduce", fake_all_reduce)
    numel = 120
    spans = {"a": (80, 100), "b": (60, 80), "c": (40, 60), "all": (0, 100), "mid": (50, 90), "tail": (95, 120), "x": (10, 20)}
    arena = types.SimpleNamespace(grad=torch.zeros(numel), numel=numel, span=lambda p: spans[p])
    for order in (["a", "b", "c", "all"], ["x", "mid", "all", "tail"], [["a", "c"], "mid", "x"], ["all", "all", "tail"], []):
        sent.clear()
        ex = trainer._GradExchange(arena, 2, None)
        for q in order:
            ex(q)
        ex.finish()
        cover = torch.zeros(numel, dtype=torch.int32)
        for x, y in sent:
            assert y > x
            cover[x:y] += 1
        assert bool((cover == 1).all()), (order, sent)


def test_trainer_checkpoints_are_torch_adam_state_dicts():
    """MipTrainer / ZipTrainer.state_dict() is torch.optim.Adam's layout over model.parameters() -- what the reference saves as 'optimzer'
    (s-nerf/train.py:269) / 'optimizer' (zipnerf/train.py:437) and resumes from (model_utils.py:53, checkpoints.py:54): it loads into a
    torch Adam, a torch Adam's state loads into the trainer, and a resumed trainer continues bit for bit."""
    from cpu_ops_emulation import emulate_ops
    from snerf_amd.trainer import ZipTrainer, MipTrainer, LossScaler
    n = 8
    with emulate_ops():
        model = _zip_build()
        tr = ZipTrainer(model, lr=1e-2, eps=1e-4, loss_scale=LossScaler(init_scale=8.0, growth_interval=2))
        batch, tgt = _zip_data(n)
        for _ in range(2):
            tr.step(batch, tgt, rand=False, aux_loss_fn=_zip_aux)
        sd = tr.state_dict()
        assert sd["param_names"] == [k for k, _ in model.named_parameters()] and sd["loss_scaler"]["scale"] == 16.0
        # -> torch.optim.Adam
        opt = torch.optim.Adam(model.parameters(), lr=1.0)
        opt.load_state_dict(sd)                                      # (torch ignores the extra top-level entries)
        assert opt.param_groups[0]["lr"] == 1e-2 and tuple(opt.param_groups[0]["betas"]) == (0.9, 0.99)
        for i, p_ in enumerate(model.parameters()):
            assert torch.equal(opt.state[p_]["exp_avg"], sd["state"][i]["exp_avg"]) and int(opt.state[p_]["step"]) == 2
        # resume: a fresh model + trainer loaded from the checkpoint takes the same third step
        model2 = _zip_build()
        model2.load_state_dict(model.state_dict())
        tr2 = ZipTrainer(model2, lr=3e-3, loss_scale="dynamic")
        tr2.load_state_dict(sd)
        assert tr2.t == 2 and tr2.lr == 1e-2 and tr2.eps == 1e-4 and tr2.scaler.scale == 16.0 and tr2.scaler.good_steps == 0
        for t_, m_ in ((tr, model), (tr2, model2)):
            t_.step(batch, tgt, rand=False, aux_loss_fn=_zip_aux)
        assert torch.equal(model.arena.flat, model2.arena.flat) and torch.equal(tr.m, tr2.m) and torch.equal(tr.v, tr2.v) and tr2.t == 3
        # torch.optim.Adam -> trainer
        model3 = _zip_build()
        opt3 = torch.optim.Adam(model3.parameters(), lr=5e-3, betas=(0.8, 0.9), eps=1e-6)
        g = torch.Generator().manual_seed(1)
        for _ in range(3):
            for p_ in model3.parameters():
                p_.grad = torch.randn(p_.shape, generator=g)
            opt3.step()
        tr3 = ZipTrainer(model3)
        tr3.load_state_dict(opt3.state_dict())
        assert tr3.t == 3 and tr3.lr == 5e-3 and tr3.betas == (0.8, 0.9) and tr3.eps == 1e-6
        for nme, p_ in model3.named_parameters():
            lo, cnt = model3.arena._offs[nme]
            assert torch.equal(tr3.m[lo:lo + cnt].view(p_.shape), opt3.state[p_]["exp_avg"])
            assert torch.equal(tr3.v[lo:lo + cnt].view(p_.shape), opt3.state[p_]["exp_avg_sq"])
        bad = opt3.state_dict()
        bad["param_groups"][0]["params"] = bad["param_groups"][0]["params"][:-1]
        with pytest.raises(ValueError):
            tr3.load_state_dict(bad)
        wd = opt3.state_dict()
        wd["param_groups"][0]["weight_decay"] = 0.1
        with pytest.raises(ValueError):
            tr3.load_state_dict(wd)
        ps = list(model3.parameters())
        two = torch.optim.Adam([{"params": ps[:3], "lr": 1e-3}, {"params": ps[3:], "lr": 1e-4}])
        with pytest.raises(ValueError):
            tr3.load_state_dict(two.state_dict())           # two learning rates cannot ride in one launch
        same = torch.optim.Adam([{"params": ps[:3]}, {"params": ps[3:]}], lr=2e-3)
        tr3.load_state_dict(same.state_dict())              # (never stepped: empty state -> zero moments, t = 0)
        assert tr3.t == 0 and tr3.lr == 2e-3 and not bool(tr3.m.any()) and not bool(tr3.v.any())
        # path A's trainer: the same layout
        ma = _build()
        ta = MipTrainer(ma, lr=1e-3)
        rays, tgt_a = _data(16)
        ta.step(rays, tgt_a, randomized=False)
        sa = ta.state_dict()
        torch.optim.Adam(ma.parameters()).load_state_dict(sa)
        mb = _build()
        mb.load_state_dict(ma.state_dict())
        tb = MipTrainer(mb, lr=1.0)
        tb.load_state_dict(sa)
        for t_ in (ta, tb):
            t_.step(rays, tgt_a, randomized=False)
        assert tb.t == 2 and torch.equal(ma.arena.flat, mb.arena.flat)
