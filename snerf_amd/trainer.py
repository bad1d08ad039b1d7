"""Training step for the mip path without the autograd round trip: forward, per-ray loss tail, backward,
gradient exchange and fused Adam on the flat arenas.

Reference counterpart: the hot loop of s-nerf/train.py:110-221 (model forward :112-115, RGB MSE
loss_factory.py:5-11, disparity-L1 depth loss with per-ray confidence loss_factory.py:26-37 /
confidence.py:209-224, ProposalLoss loss_factory.py:59-74, loss.backward() :213, optimizer.step() :217-221).
The per-ray loss tail is one kernel that returns the loss terms and
dL/d(rgb, distance, coarse weights) -- no [N]-sized torch expressions, no device->host sync in the step.

Multi-GPU: one process per GPU, every rank renders its own shard of the ray batch; the only exchange is ONE
RCCL all-reduce of the flat fp32 gradient arena (35.9 MB for the shipped model) before the Adam kernel, which
folds the 1/world_size mean into its pass.
"""
import torch
import torch.distributed as dist

from . import ops


class MipTrainer:
    def __init__(self, model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, depth_lambda=0.2, coarse_depth_mult=0.2,
                 proposal_loss=False, proposal_lambda=0.05, disparity_depth=True, process_group=None):
        self.model = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.depth_lambda, self.coarse_depth_mult = depth_lambda, coarse_depth_mult
        self.proposal_loss, self.proposal_lambda, self.disparity_depth = proposal_loss, proposal_lambda, disparity_depth
        a = model.arena
        self.m = torch.zeros_like(a.flat)
        self.v = torch.zeros_like(a.flat)
        self.t = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        a.grad.zero_()

    def broadcast_parameters(self, src=0):
        if self.world > 1:
            dist.broadcast(self.model.arena.flat, src=src, group=self.pg)
            self.model.arena.bump()

    def loss_and_grads(self, outs, target_rgb, target_depth, conf):
        """The reference's per-ray loss tail (train.py:150-208) in ONE kernel, `snerf_mip_loss_tail`: RGB MSE, the
        confidence-weighted (disparity) depth loss on both levels masked to rays with a LiDAR target, and -- with
        `proposal_loss` -- ProposalLoss on the two histograms.  Returns (loss [device scalar], output gradients)."""
        dist0, acc0, s0, w0, rgb1, dist1, acc1, s1, w1 = outs
        prop = self.proposal_loss
        out, g_rgb1, g_dist1, g_dist0, g_w0 = ops.mip_loss_tail(
            rgb1, target_rgb, dist1 if target_depth is not None else None, dist0 if target_depth is not None else None,
            target_depth, conf, s1 if prop else None, w1 if prop else None, s0 if prop else None, w0 if prop else None,
            self.disparity_depth, self.depth_lambda, self.coarse_depth_mult, self.proposal_lambda)
        self.last_losses = out                                   # {#valid depth rays, rgb, depth, proposal}: stays on the device
        return out[1:].sum(), (g_dist0, None, g_w0, g_rgb1, g_dist1, None, None)

    def step(self, rays, target_rgb, target_depth=None, conf=None, randomized=True, s_rand=None, u=None):
        m = self.model
        dev = m.arena.flat.device
        n = rays.origins.shape[0]
        ds_rand, du, noise0, noise1 = m._draws(n, randomized, dev)
        s_rand = ds_rand if s_rand is None else s_rand
        u = du if u is None else u
        outs, ctx = m._run(rays, True, False, s_rand, u.contiguous(), noise0, noise1)
        loss, g = self.loss_and_grads(outs, target_rgb, target_depth, conf)
        m._backward(ctx, *g)
        if self.world > 1:
            dist.all_reduce(m.arena.grad, op=dist.ReduceOp.SUM, group=self.pg)
        self.t += 1
        ops.adam_step(m.arena.flat, m.arena.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t,
                      grad_scale=1.0 / self.world, zero_grad=True)
        m.arena.bump()
        return loss, outs


def shard_rays(rays, rank: int, world: int):
    """Contiguous, equal split of a ray batch (namedtuple of [N,.] tensors) across ranks."""
    n = rays[0].shape[0]
    per = n // world
    return type(rays)(*[r[rank * per:(rank + 1) * per] for r in rays])
