"""Training step for the mip path without the autograd round trip: forward, per-ray loss tail, backward,
gradient exchange and fused Adam on the flat arenas.

Reference counterpart: the hot loop of s-nerf/train.py:110-221 (model forward :112-115, RGB MSE
loss_factory.py:5-11, disparity-L1 depth loss with per-ray confidence loss_factory.py:26-37 /
confidence.py:209-224, ProposalLoss loss_factory.py:59-74, loss.backward() :213, optimizer.step() :217-221).
The per-ray loss tail is one kernel that returns the loss terms and
dL/d(rgb, distance, coarse weights) -- no [N]-sized torch expressions, no device->host sync in the step.

Multi-GPU: one process per GPU, every rank renders its own shard of the ray batch; the only exchange is the
RCCL all-reduce of the flat fp32 gradient arena (35.9 MB for the shipped model), issued network by network as
the backward pass completes each block so that it overlaps with the rest of the backward, before the Adam
kernel, which folds the 1/world_size mean into its pass.
"""
import torch
import torch.distributed as dist

from . import ops


class _GradExchange:
    """Gradient all-reduce overlapped with the backward pass: the arena is laid out network by network and layer by layer, and as soon
    as a block's gradients are final -- the NeRF MLP's heads, then each 4 MB trunk layer as the backward reaches it, then the proposal
    network -- it is all-reduced asynchronously (RCCL runs it on its own stream) while the remaining backward kernels keep the compute
    stream busy: only the last trunk layer's bucket (~0.4 MB) is left exposed in front of Adam.  `finish()` reduces whatever was not
    announced and waits for everything."""

    def __init__(self, arena, world, group, single_rank=False):
        """`single_rank`: run the collectives even in a process group of one rank (diagnostics: exercises the RCCL path on a 1-GPU box)"""
        self.arena, self.world, self.group = arena, world, group
        self.active = world > 1 or single_rank
        self.works, self.done = [], []

    def __call__(self, prefix):
        """`prefix`: a parameter-name prefix or a list of them; neighbouring spans go out as ONE collective, and what an earlier call
        already covered is not sent twice (the backward announces the NeRF MLP layer by layer and then once more as a whole)."""
        if not self.active:
            return
        spans = sorted(self.arena.span(p) for p in ([prefix] if isinstance(prefix, str) else prefix))
        merged = []
        for a, b in spans:
            if merged and a <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], b)
            else:
                merged.append([a, b])
        for a, b in merged:
            # [a, b) minus the union of what is already on the wire, in one sweep over the sent spans in ascending order
            pos = a
            for c, d in sorted(self.done):
                if d <= pos:
                    continue
                if c >= b:
                    break
                if c > pos:
                    self._send(pos, c)
                pos = max(pos, d)
            if pos < b:
                self._send(pos, b)

    def _send(self, x, y):
        self.done.append((x, y))
        self.works.append(dist.all_reduce(self.arena.grad[x:y], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        if not self.active:
            return
        pos = 0
        for a, b in sorted(self.done) + [(self.arena.numel, self.arena.numel)]:
            if a > pos:
                self.works.append(dist.all_reduce(self.arena.grad[pos:a], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            pos = max(pos, b)
        for w in self.works:
            w.wait()
        self.works, self.done = [], []


class _AdamState:
    """Checkpointing of the fused Adam launch's state in torch.optim.Adam's `state_dict()` layout over `model.parameters()` -- what the
    reference stores next to the weights (s-nerf/train.py:264-273 'optimzer', resumed at utils/model_utils.py:44-63; zipnerf/train.py:432-440
    'optimizer', internal/checkpoints.py:52-54): a checkpoint written by the reference's training loop resumes here and the other way round
    (`torch.optim.Adam(model.parameters()).load_state_dict(trainer.state_dict())`)."""

    def _moments(self):
        return self.m, self.v

    def state_dict(self):
        a = self.model.arena
        m, v = self._moments()
        names = [n for n, _ in self.model.named_parameters()]
        state = {}
        for i, n in enumerate(names):
            lo, cnt = a._offs[n]
            state[i] = {"step": torch.tensor(float(self.t)), "exp_avg": m[lo:lo + cnt].view(a.shapes[n]).clone(),
                        "exp_avg_sq": v[lo:lo + cnt].view(a.shapes[n]).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(names)))}
        sd = {"state": state, "param_groups": [group], "param_names": names}
        if getattr(self, "scaler", None) is not None:
            sd["loss_scaler"] = self.scaler.state_dict()
        elif getattr(self, "loss_scale", None) is not None:
            sd["static_loss_scale"] = float(self.loss_scale)
        return sd

    def load_state_dict(self, sd):
        a = self.model.arena
        names = [n for n, _ in self.model.named_parameters()]
        groups = sd["param_groups"]
        if sum(len(g["params"]) for g in groups) != len(names):
            raise ValueError(f"optimizer state for {sum(len(g['params']) for g in groups)} parameters, the model has {len(names)}")
        if "param_names" in sd and list(sd["param_names"]) != names:
            raise ValueError("optimizer state was saved for other parameter names")
        order = [i for g in groups for i in g["params"]]
        steps = set()
        with torch.no_grad():
            for pos, n in zip(order, names):
                lo, cnt = a._offs[n]
                st = sd["state"].get(pos, sd["state"].get(str(pos)))
                if st is None:                              # (a parameter the saved optimizer never stepped)
                    self.m[lo:lo + cnt].zero_(); self.v[lo:lo + cnt].zero_()
                    continue
                if tuple(st["exp_avg"].shape) != tuple(a.shapes[n]):
                    raise ValueError(f"optimizer state of {n}: shape {tuple(st['exp_avg'].shape)}, the parameter has {tuple(a.shapes[n])}")
                self.m[lo:lo + cnt].copy_(st["exp_avg"].reshape(-1).to(self.m.device, torch.float32))
                self.v[lo:lo + cnt].copy_(st["exp_avg_sq"].reshape(-1).to(self.v.device, torch.float32))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"one Adam launch covers the whole arena: the saved parameters disagree on the step count ({sorted(steps)})")
        self.t = steps.pop() if steps else 0
        g0 = groups[0]
        for g in groups[1:]:                                # one launch, one set of hyper-parameters
            if (float(g["lr"]), tuple(g["betas"]), float(g["eps"])) != (float(g0["lr"]), tuple(g0["betas"]), float(g0["eps"])):
                raise ValueError("the fused Adam launch covers the whole arena with one lr / betas / eps: the saved param_groups differ")
        new_betas, new_eps = tuple(float(b) for b in g0["betas"]), float(g0["eps"])
        if getattr(self, "_graph", None) is not None and (new_betas != tuple(float(b) for b in self.betas) or new_eps != float(self.eps)):
            # betas / eps are host kernel arguments baked into the captured hipGraph (lr and the step count live on the device)
            raise ValueError("load_state_dict after capture(): the checkpoint's betas / eps differ from the captured step's -- load before capturing")
        self.lr, self.betas, self.eps = float(g0["lr"]), new_betas, new_eps
        if float(g0.get("weight_decay", 0) or 0) != 0 or g0.get("amsgrad", False):
            raise ValueError("the fused Adam launch has no weight decay / amsgrad")
        if getattr(self, "_step_dev", None) is not None:
            self._step_dev.fill_(self.t)
        if getattr(self, "scaler", None) is not None and "loss_scaler" in sd:
            self.scaler.load_state_dict(sd["loss_scaler"])
            self.loss_scale = self.scaler.scale
        elif getattr(self, "scaler", None) is None and "static_loss_scale" in sd and getattr(self, "loss_scale", None) is not None \
                and float(sd["static_loss_scale"]) != float(self.loss_scale):
            import warnings
            warnings.warn(f"resuming with a static loss scale of {self.loss_scale:g}; the checkpoint was trained with {float(sd['static_loss_scale']):g}")


class MipTrainer(_AdamState):
    def __init__(self, model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, depth_lambda=0.2, coarse_depth_mult=0.2,
                 proposal_loss=False, proposal_lambda=0.05, disparity_depth=True, process_group=None,
                 nonfinite="zero", grad_max_val=0.0, grad_max_norm=0.0, exchange_when_single=False):
        """`nonfinite` / `grad_max_val` / `grad_max_norm`: gradient hygiene folded into the Adam launch (ops.adam_step).  The s-nerf
        reference has none (train.py:212-215 drops into pdb on a failing backward); the default "zero" keeps one NaN / Inf gradient
        (1/(depth + eps), a bf16 overflow) from poisoning m, v and the parameters of the whole arena on every rank.  "keep" = plain Adam."""
        self.model = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.nonfinite, self.grad_max_val, self.grad_max_norm = nonfinite, float(grad_max_val), float(grad_max_norm)
        self.depth_lambda, self.coarse_depth_mult = depth_lambda, coarse_depth_mult
        self.proposal_loss, self.proposal_lambda, self.disparity_depth = proposal_loss, proposal_lambda, disparity_depth
        a = model.arena
        self.m = torch.zeros_like(a.flat)
        self.v = torch.zeros_like(a.flat)
        self.t = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.single_rank_exchange = bool(exchange_when_single) and dist.is_available() and dist.is_initialized()
        self._step_dev = self._lr_dev = self._graph = None
        a.grad.zero_()

    def broadcast_parameters(self, src=0):
        if self.world > 1:
            dist.broadcast(self.model.arena.flat, src=src, group=self.pg)
            self.model.arena.bump()

    def _adam(self):
        """One fused Adam launch over the flat arena; folds the data-parallel mean, the gradient hygiene and zero_grad."""
        a = self.model.arena
        coef = ops.grad_clip_coef(a.grad, 1.0 / self.world, self.grad_max_norm) if self.grad_max_norm > 0 else None
        ops.adam_step(a.flat, a.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t, grad_scale=1.0 / self.world,
                      zero_grad=True, nonfinite=self.nonfinite, grad_max_val=self.grad_max_val, clip_coef=coef,
                      step_dev=self._step_dev, lr_dev=self._lr_dev)
        a.bump()

    def loss_and_grads(self, outs, target_rgb, target_depth, conf):
        """The reference's per-ray loss tail (train.py:150-208) in ONE kernel, `snerf_mip_loss_tail`: RGB MSE, the
        confidence-weighted (disparity) depth loss on both levels masked to rays with a LiDAR target, and -- with
        `proposal_loss` -- ProposalLoss on the two histograms.  Returns (loss [device scalar], output gradients)."""
        dist0, acc0, s0, w0, rgb1, dist1, acc1, s1, w1 = outs[:9]
        prop = self.proposal_loss
        out, g_rgb1, g_dist1, g_dist0, g_w0 = ops.mip_loss_tail(
            rgb1, target_rgb, dist1 if target_depth is not None else None, dist0 if target_depth is not None else None,
            target_depth, conf, s1 if prop else None, w1 if prop else None, s0 if prop else None, w0 if prop else None,
            self.disparity_depth, self.depth_lambda, self.coarse_depth_mult, self.proposal_lambda)
        self.last_losses = out[:4]                               # {#valid depth rays, rgb, depth, proposal}: stays on the device
        return out[4], (g_dist0, None, g_w0, g_rgb1, g_dist1, None, None)      # out[4]: their sum, formed by the same launch

    def step(self, rays, target_rgb, target_depth=None, conf=None, randomized=True, s_rand=None, u=None, ray_grads=False, viewc=None):
        """`ray_grads=True` (pose refinement, configs: pose_refine = True): `last_ray_grads` = d loss / d (origins, directions, viewdirs)
        of this rank's rays, for the caller's pose optimiser (`rays.origins.backward(g_o)` etc. chains them into the pose parameters).
        `viewc` (fn = 0 models, the view-centred warp): the warp centre the reference passes to every forward (train.py:36,112); a model
        built with fn = 0 refuses to step until a centre has been given here or through `model.set_viewc`."""
        self._set_viewc(viewc)
        ex = _GradExchange(self.model.arena, self.world, self.pg, self.single_rank_exchange)
        # every block of the gradient arena goes on the wire as soon as the backward pass has finished it (heads, then trunk layer by
        # trunk layer, then the proposal network): the collectives run under the remaining backward kernels
        loss, outs = self._forward_backward(rays, target_rgb, target_depth, conf, randomized, s_rand, u, ray_grads, ex)
        ex.finish()
        self.t += 1
        self._adam()                                      # graph mode: step count and lr live on the device (capture / replay below)
        return loss, outs

    def _set_viewc(self, viewc):
        m = self.model
        if viewc is not None:
            m.set_viewc(viewc)
        elif m.fn == 0 and not getattr(m, "_viewc_given", False):
            raise ValueError("MipTrainer: the model warps around a view centre (fn = 0): pass viewc= (train.py:36,112) or call model.set_viewc first")

    def _forward_backward(self, rays, target_rgb, target_depth, conf, randomized, s_rand, u, ray_grads, on_done):
        """draws, both levels forward, the fused loss tail, the backward pass into the gradient arena (no exchange, no optimiser)"""
        m = self.model
        dev = m.arena.flat.device
        n = rays.origins.shape[0]
        ds_rand, du, noise0, noise1 = m._draws(n, randomized, dev)
        s_rand = ds_rand if s_rand is None else s_rand
        u = du if u is None else u
        outs, ctx = m._run(rays, True, False, s_rand, u.contiguous(), noise0, noise1)
        loss, g = self.loss_and_grads(outs, target_rgb, target_depth, conf)
        self.last_ray_grads = m._backward(ctx, *g, on_done=on_done, ray_grads=ray_grads)
        return loss, outs

    # ---- hipGraph capture of the whole step --------------------------------------------------------------------------------------
    def capture(self, rays, target_rgb, target_depth=None, conf=None, randomized=True, warmup=3, viewc=None):
        """Capture one full training step (draws, forward, loss tail, backward, Adam: ~130 launches) in a hipGraph over the GIVEN tensors;
        afterwards `replay()` runs a step with one graph launch -- the caller refreshes the batch by copying into those tensors
        (`rays.origins.copy_(...)` etc.).  Worth it when the step is launch-bound: at 512 rays per GPU (the 8-GPU split of the
        reference's 4096-ray batch) the kernels are 10-50 us each.  The packed-weight refresh, the torch RNG draws (graph-safe Philox
        offsets), the Adam bias corrections (step count in device memory) and the learning rate (device scalar, refreshed from
        `self.lr` by every `replay()`, so an lr schedule keeps working) are all inside the graph.

        The `warmup` steps (one-time kernel attributes, allocator pools) are REAL steps on the capture batch, so parameters, Adam
        moments and the step count are snapshotted before and restored after them: capturing does not train.  (The torch RNG does
        advance.)

        Data parallel (world > 1; every rank calls capture / replay together): the graph holds forward + loss tail + backward -- the
        launch-bound part -- and `replay()` follows it with the gradient all-reduce and the Adam launch OUTSIDE the graph (one
        collective over the whole arena: nothing is left to overlap it with once the backward is a single graph launch; the
        collective stays out of the graph so that any backend works, gloo on the 1-GPU box included)."""
        self._set_viewc(viewc)                           # (fn = 0: the centre is a kernel argument, i.e. a constant of the captured graph)
        a = self.model.arena
        dev = a.flat.device
        snap = (a.flat.clone(), self.m.clone(), self.v.clone(), self.t)
        if self.world == 1:
            self._step_dev = torch.tensor([self.t], dtype=torch.int32, device=dev)
            self._lr_dev = torch.tensor([self.lr], dtype=torch.float32, device=dev)
        args = (rays, target_rgb, target_depth, conf)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                    # warm-up on a side stream
            for _ in range(warmup):
                self.step(*args, randomized=randomized)
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():
            a.flat.copy_(snap[0]); self.m.copy_(snap[1]); self.v.copy_(snap[2])
            self.t = snap[3]
            if self._step_dev is not None:
                self._step_dev.fill_(self.t)
            a.grad.zero_()
        a.bump()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            if self.world == 1:
                self._graph_loss, self._graph_outs = self.step(*args, randomized=randomized)
            else:
                self._graph_loss, self._graph_outs = self._forward_backward(*args, randomized, None, None, False, None)
        if self.world == 1:
            self.t -= 1                                  # capturing records the step, it does not run it
        return self._graph_loss

    def replay(self):
        """One captured step.  -> (loss, outs): the same device tensors every time, overwritten by each replay."""
        if self.world == 1:
            self._lr_dev.fill_(self.lr)                  # the graph reads the learning rate from the device
            self._graph.replay()
            self.t += 1
            return self._graph_loss, self._graph_outs
        self._graph.replay()                             # forward + loss tail + backward of this rank's shard
        ex = _GradExchange(self.model.arena, self.world, self.pg)
        ex.finish()                                      # one all-reduce of the flat gradient arena
        self.t += 1
        self._adam()
        return self._graph_loss, self._graph_outs


def shard_bounds(n: int, rank: int, world: int):
    """[start, end) of rank's contiguous share of n rays; the first n % world ranks take one extra ray, nothing is dropped."""
    per, rem = divmod(n, world)
    start = rank * per + min(rank, rem)
    return start, start + per + (1 if rank < rem else 0)


def shard_rays(rays, rank: int, world: int):
    """Contiguous split of a ray batch (namedtuple of [N,.] tensors) across ranks; see shard_bounds."""
    a, b = shard_bounds(rays[0].shape[0], rank, world)
    return type(rays)(*[r[a:b] for r in rays])


def shard_batch(rays, rank: int, world: int, *per_ray):
    """shard_rays for the rays AND every per-ray target (target_rgb, target_depth, conf, ...; None passes through) with the same
    bounds, so that a caller cannot slice them inconsistently.  -> (rays_shard, *target_shards)"""
    a, b = shard_bounds(rays[0].shape[0], rank, world)
    for t in per_ray:
        if t is not None and t.shape[0] != rays[0].shape[0]:
            raise ValueError("per-ray tensor does not match the ray batch")
    return (type(rays)(*[r[a:b] for r in rays]),) + tuple(None if t is None else t[a:b] for t in per_ray)


class _TableShards:
    """Hash-table gradients without the dense all-reduce (the reference all-reduces the dense ~310 MB through DDP,
    zipnerf/train.py:334).  Touched-rows-only exchange does not apply at the reference's batch sizes -- a rank's 8 192 rays put 14.7 M
    multisample cells on every 2^21-row level: every row is touched -- so the table spans go the ZeRO-1 way instead: REDUCE-SCATTER of
    the gradient span (each rank receives the sum of its 1 / world slice), Adam on that slice only (its m / v slices; 1 / world of the
    optimiser pass), ALL-GATHER of the updated parameter slices.  Same bytes on the wire as the all-reduce it replaces (a ring all-reduce
    is exactly these two phases) with the optimiser work of the tables sharded; the MLP parameters keep the bucketed all-reduce.
    Level sizes are multiples of 8 rows (grid.py:130), so the spans divide evenly for world sizes 2 / 4 / 8; other world sizes, and
    global-norm clipping (which needs the norm of the whole reduced gradient), use the all-reduce path.  A rank's slice of a
    single-channel table is NOT a multiple of 4 floats at world 4 / 8 (6 606 952 / 8 = 825 869): ops.adam_step takes its four
    pointers at any 4-byte alignment (scalar flavour of the kernel; tests/test_gpu_kernels.py::test_adam_on_misaligned_slices...)."""

    def __init__(self, arena, names, world, group):
        self.arena, self.world, self.group = arena, world, group
        self.rank = dist.get_rank(group)
        self.spans = {}
        for n in names:
            a, b = arena.span(n)
            if (b - a) % world != 0:
                raise ValueError("table span does not divide by the world size")
            self.spans[n] = (a, b, (b - a) // world)
        self.nccl = dist.get_backend(group) == "nccl"
        self.pending = []

    def reduce_scatter(self, name):
        """start the reduce-scatter of one table's gradient span; -> nothing (finish() waits)"""
        a, b, per = self.spans[name]
        g = self.arena.grad[a:b]
        mine = g[self.rank * per:(self.rank + 1) * per]
        if self.nccl:
            out = torch.empty_like(mine)
            w = dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((name, out, [w]))
        else:       # (gloo has no reduce-scatter: one reduce per destination rank; functional tests only)
            ws = [dist.reduce(g[r * per:(r + 1) * per], dst=dist.get_global_rank(self.group, r) if self.group is not None else r,
                              op=dist.ReduceOp.SUM, group=self.group, async_op=True) for r in range(self.world)]
            self.pending.append((name, mine, ws))

    def finish(self):
        """-> [(name, flat slice [lo, hi) of this rank, reduced gradient slice)]"""
        out = []
        for name, gshard, ws in self.pending:
            for w in ws:
                w.wait()
            a, b, per = self.spans[name]
            out.append((name, a + self.rank * per, a + (self.rank + 1) * per, gshard.clone() if not self.nccl else gshard))
        self.pending = []
        return out

    def all_gather_params(self, name):
        self.all_gather_span(self.arena.flat, name)

    def all_gather_span(self, flat, name):
        """every rank's slice of the table span of `flat` (the parameters, or an Adam moment of the same layout) to every rank"""
        a, b, per = self.spans[name]
        full = flat[a:b]
        mine = full[self.rank * per:(self.rank + 1) * per]
        dist.all_gather_into_tensor(full, mine if self.nccl else mine.clone(), group=self.group)


class LossScaler:
    """The policy of torch.cuda.amp.GradScaler (what accelerate wraps around the reference's fp16 training, zipnerf/train.py:44,215,331) as
    host state: the scale starts at `init_scale`, a step whose gradients hold a NaN / +-Inf is skipped and multiplies it by
    `backoff_factor`, `growth_interval` consecutive clean steps multiply it by `growth_factor`.  Same defaults as torch."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        if not (init_scale > 0 and growth_factor > 1.0 and 0.0 < backoff_factor < 1.0 and growth_interval >= 1):
            raise ValueError("LossScaler: init_scale > 0, growth_factor > 1, 0 < backoff_factor < 1, growth_interval >= 1")
        self.scale, self.growth_factor, self.backoff_factor = float(init_scale), float(growth_factor), float(backoff_factor)
        self.growth_interval, self.good_steps, self.skipped_steps = int(growth_interval), 0, 0

    def update(self, found_inf: bool) -> bool:
        """-> True when the optimizer step has to be skipped"""
        if found_inf:
            self.scale *= self.backoff_factor
            self.good_steps = 0
            self.skipped_steps += 1
            return True
        self.good_steps += 1
        if self.good_steps >= self.growth_interval:
            self.scale *= self.growth_factor
            self.good_steps = 0
        return False

    def state_dict(self):
        return dict(scale=self.scale, growth_factor=self.growth_factor, backoff_factor=self.backoff_factor, growth_interval=self.growth_interval,
                    good_steps=self.good_steps, skipped_steps=self.skipped_steps)

    def load_state_dict(self, d):
        self.scale, self.growth_factor, self.backoff_factor = float(d["scale"]), float(d["growth_factor"]), float(d["backoff_factor"])
        self.growth_interval, self.good_steps, self.skipped_steps = int(d["growth_interval"]), int(d["good_steps"]), int(d.get("skipped_steps", 0))


class ZipTrainer(_AdamState):
    """Train step of the S-NeRF++ / zipnerf background model (s-nerfpp/zipnerf/train.py hot loop :218-331: Model.forward, the loss
    terms, loss.backward(), optimizer.step()) on the flat arenas: forward, ONE fused loss-tail launch (ops.zip_loss_tail: Charbonnier
    data term, disparity-L1 depth terms, semantic NLL, anti-interlevel and distortion regularisers, with their gradients), backward
    through the fused kernels, the RCCL all-reduce of the flat gradient arena (MLPs + the 3 hash tables, ~310 MB for waymo.gin) issued
    level by level so that it overlaps with the remaining backward, and one fused Adam launch with the 1/world mean folded in.  No device->host sync anywhere in the step: the loss terms stay on the
    device in `last_losses` (ops.ZIP_LOSS_NAMES order).

    `loss_cfg` overrides the reference defaults (internal/configs.py:60-66,85; train.py:253,272,298): charb_padding 0.001, data_mult
    1, depth_lambda 0.5, com_mult 0.2, sem_mult 0.04, pulse_width (0.03, 0.003), interlevel_mult 0.01, distortion_mult 0.005, mse False,
    hash_decay_mult 0.1 (`snerf_hash_decay`, one pass over the three tables, value appended to `last_losses`).
    Further caller-defined terms on the ray histories go through `aux_loss_fn(ray_history) -> scalar` (torch autograd on detached
    leaf copies of every level's `weights`; its d(loss)/d(weights) is added to the fused tail's)."""

    def __init__(self, model, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, charb_padding=0.001, process_group=None, loss_cfg=None,
                 nonfinite="zero", grad_max_val=0.0, grad_max_norm=0.0, table_exchange="sharded", loss_scale=None):
        """`loss_scale` (a number = static; default 4096 when the model computes in fp16, else 1): the gradients of the rendered outputs are
        multiplied by it before the backward -- so that the fp16 gradient buffers of the networks stay in fp16's normal range -- and the
        factor is undone inside the Adam launch (grad_scale), before clipping.  Overflowed (non-finite) gradients are dropped by
        `nonfinite` and counted in `dropped_nonfinite` (device int64: watch it in fp16 runs -- GradScaler would have skipped those steps).
        `loss_scale="dynamic"` (or a LossScaler): torch's GradScaler policy -- one pass over the gradient arena after the
        exchange (snerf_nonfinite_flag, + a 4-byte MAX all-reduce when the tables are sharded), ONE device->host read of the flag per
        step (as GradScaler.step does), a step with an overflow is skipped whole (parameters, m, v and the step count t untouched,
        gradients zeroed) and halves the scale; `scaler.skipped_steps` counts them.  The static scale keeps the step free of host syncs.
        `nonfinite`, `grad_max_val`, `grad_max_norm` = train_utils.clip_gradients (train_utils.py:234-243, run every step at
        zipnerf/train.py:336; configs.py:83-84 defaults 0 = off), folded into the Adam launch.  The reference always ends with
        param.grad.nan_to_num_(); "zero" (default) also drops +-Inf instead of mapping it to +-FLT_MAX (which would leave v = inf,
        i.e. that parameter frozen for good); "nan_to_num" reproduces the reference exactly."""
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.nonfinite, self.grad_max_val, self.grad_max_norm = nonfinite, float(grad_max_val), float(grad_max_norm)
        self.loss_cfg = dict(charb_padding=charb_padding)
        self.loss_cfg.update(loss_cfg or {})
        # hash-grid weight decay (train_utils.py:184-203; configs.py:74): a term on the parameters, not on the rays
        self.hash_decay_mult = float(self.loss_cfg.pop("hash_decay_mult", 0.1))
        a = model.arena
        self.m, self.v, self.t = torch.zeros_like(a.flat), torch.zeros_like(a.flat), 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.last_losses = None
        # running count of NaN / +-Inf gradient elements the Adam launches met (device int64, no host sync; read it when logging): a
        # persistently overflowing fp16 run with the static scale looks healthy otherwise -- the `nonfinite` policy drops them silently
        self.dropped_nonfinite = torch.zeros(1, dtype=torch.int64, device=a.flat.device)
        self.scaler = LossScaler() if isinstance(loss_scale, str) and loss_scale == "dynamic" else (loss_scale if isinstance(loss_scale, LossScaler) else None)
        if isinstance(loss_scale, str) and self.scaler is None:
            raise ValueError(f"loss_scale: a number, 'dynamic' or a LossScaler, not {loss_scale!r}")
        if self.scaler is not None:
            self.loss_scale = self.scaler.scale
            self._found_inf = torch.zeros(1, dtype=torch.int32, device=a.flat.device)
        else:
            self.loss_scale = float(loss_scale) if loss_scale is not None else (4096.0 if getattr(model, "dt", None) == ops.F16 else 1.0)
        a.grad.zero_()
        # the hash tables' gradients: "sharded" (default for N > 1) = reduce-scatter + Adam on the rank's slice + all-gather of the updated
        # parameters (_TableShards); "allreduce" = part of the bucketed all-reduce like the MLP parameters (the reference's DDP behaviour)
        if table_exchange not in ("sharded", "allreduce"):
            raise ValueError(table_exchange)
        self.tables = [n + "encoder.embeddings" for n in model.names]
        self.shards = None
        if table_exchange == "sharded" and self.world > 1 and self.grad_max_norm <= 0 and \
                all((a.span(n)[1] - a.span(n)[0]) % self.world == 0 for n in self.tables):
            self.shards = _TableShards(a, self.tables, self.world, process_group)

    def broadcast_parameters(self, src=0):
        if self.world > 1:
            dist.broadcast(self.model.arena.flat, src=src, group=self.pg)
            self.model.arena.bump()

    def _moments(self):
        """sharded table updates keep a rank's Adam moments current on its own slices only: a checkpoint gathers them (collective: every
        rank has to call state_dict())"""
        if self.shards is None:
            return self.m, self.v
        m, v = self.m.clone(), self.v.clone()
        for name in self.tables:
            self.shards.all_gather_span(m, name)
            self.shards.all_gather_span(v, name)
        return m, v

    def _overflowed(self, a, shards):
        """dynamic loss scaling: the found-inf pass over the exchanged gradients and the scaler's update; -> True = the step is skipped
        (gradients zeroed; parameters, moments and t stay as they were)"""
        self._found_inf.zero_()
        ops.nonfinite_flag(a.grad, self._found_inf)
        if shards is not None:         # sharded tables: each rank sees the reduced gradient of its slices only -> agree on the flag
            for _, _, _, gshard in shards:
                ops.nonfinite_flag(gshard, self._found_inf)
            dist.all_reduce(self._found_inf, op=dist.ReduceOp.MAX, group=self.pg)
        if not self.scaler.update(bool(int(self._found_inf.item()))):
            return False
        a.grad.zero_()
        return True

    def step(self, batch, target_rgb, train_frac=1.0, rand=True, aux_loss_fn=None, draws=None, sample_n=7, sample_m=3, targets=None, zero_glo=False):
        """`targets` (all optional, per ray): lossmult [R] (the reference's mask_rgb as 0/1 floats), depth [R] + depth_mask [R]
        (+ complete_mask [R]), semantic int32 labels [R] + semantic_mask [R].  `zero_glo` (models with GLO vectors): train with zero
        vectors instead of the rows batch['cam_idx'] selects (zipnerf/train.py:235 passes zero_glo=False)."""
        m = self.model
        m._zero_glo = bool(zero_glo)
        dev = m.arena.flat.device
        R = batch['origins'].shape[0]
        t = targets or {}
        if draws is None:
            draws = m._draws(R, rand, dev, sample_n)
        levels, ctx = m._run(batch, True, float(train_frac), draws, sample_n, sample_m)
        fin = levels[2]
        with_depth = t.get("depth") is not None
        with_sem = t.get("semantic") is not None and fin.get("semantic") is not None
        out, G = ops.zip_loss_tail(fin["rgb"], target_rgb, t.get("lossmult"), depth=fin["depth"] if with_depth else None,
                                   tdepth=t.get("depth"), dmask=t.get("depth_mask"), cmask=t.get("complete_mask"),
                                   sem=fin["semantic"] if with_sem else None, labels=t.get("semantic") if with_sem else None,
                                   smask=t.get("semantic_mask"), hist=[(levels[l]["sdist"], levels[l]["weights"]) for l in range(3)],
                                   **self.loss_cfg)
        decay = torch.zeros(1, dtype=torch.float32, device=dev)
        if self.scaler is not None:
            self.loss_scale = self.scaler.scale
        ls = self.loss_scale
        if ls != 1.0:
            for g in G.values():
                if g is not None:
                    g.mul_(ls)
        if self.hash_decay_mult > 0:
            for lvl, pre in enumerate(m.names):            # identical on every rank: the all-reduced mean leaves it unchanged
                e = m.encs[lvl]
                ops.hash_decay(m.arena.p[pre + "encoder.embeddings"], m.arena.g[pre + "encoder.embeddings"], m.dev_offsets[lvl], e.L, e.C,
                               self.hash_decay_mult, decay, grad_mult=ls)
        self.last_losses = torch.cat([out[4:], decay])     # ops.ZIP_LOSS_NAMES + ("hash_decay",)
        loss = out[4] + out[6:].sum() + decay[0]
        g_w = [G["w0"], G["w1"], G["w2"]]
        if aux_loss_fn is not None:
            with torch.enable_grad():
                leaves = [levels[l]["weights"].detach().requires_grad_(True) for l in range(3)]
                hist = [dict(sdist=levels[l]["sdist"].detach(), tdist=levels[l]["tdist"].detach(), weights=leaves[l]) for l in range(3)]
                aux = aux_loss_fn(hist)
                gs = torch.autograd.grad(aux, leaves, allow_unused=True)
            gs = [b if (b is None or ls == 1.0) else b * ls for b in gs]
            g_w = [a if b is None else (b if a is None else a + b) for a, b in zip(g_w, gs)]
            loss = loss + aux.detach()
        ex = _GradExchange(m.arena, self.world, self.pg)
        sh = self.shards
        if sh is not None:
            for n in self.tables:                      # the table spans do not ride in the all-reduce buckets
                ex.done.append(m.arena.span(n))

        def level_done(prefix):
            # a level's gradients are final: its table starts its reduce-scatter, its MLP parameters their all-reduce -- the NeRF level
            # (240 MB of table gradient) is on the wire while the two proposal levels' backward runs
            if sh is not None and isinstance(prefix, str) and prefix + "encoder.embeddings" in sh.spans:
                sh.reduce_scatter(prefix + "encoder.embeddings")
            ex(prefix)
        m._backward(ctx, [(None, None, None, g_w[0]), (None, None, None, g_w[1]), (G["rgb"], G["depth"], None, g_w[2], G["semantic"])], on_done=level_done)
        ex.finish()
        a = m.arena
        mine = sh.finish() if sh is not None else None
        if self.scaler is not None and self._overflowed(a, mine):
            return loss, levels
        self.t += 1
        adam = lambda lo, hi, g=None: ops.adam_step(a.flat[lo:hi], a.grad[lo:hi] if g is None else g, self.m[lo:hi], self.v[lo:hi], self.lr, self.betas[0],
                                                    self.betas[1], self.eps, self.t, grad_scale=1.0 / (self.world * ls), zero_grad=True,
                                                    nonfinite=self.nonfinite, grad_max_val=self.grad_max_val, dropped=self.dropped_nonfinite)
        if sh is None:
            coef = ops.grad_clip_coef(a.grad, 1.0 / (self.world * ls), self.grad_max_norm) if self.grad_max_norm > 0 else None
            ops.adam_step(a.flat, a.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t,
                          grad_scale=1.0 / (self.world * ls), zero_grad=True, nonfinite=self.nonfinite, grad_max_val=self.grad_max_val, clip_coef=coef,
                          dropped=self.dropped_nonfinite)
        else:
            # tables: this rank's slice only, then the updated slices are gathered; everything between the table spans: the usual pass
            for name, lo, hi, gshard in mine:
                adam(lo, hi, gshard)
            pos = 0
            for ta, tb in sorted(a.span(n) for n in self.tables) + [(a.numel, a.numel)]:
                if ta > pos:
                    adam(pos, ta)
                if tb > ta:
                    a.grad[ta:tb].zero_()
                pos = max(pos, tb)
            for name in self.tables:
                sh.all_gather_params(name)
        m.arena.bump()
        return loss, levels
