"""Drop-in ``GridEncoder`` of S-NeRF++ / zipnerf backed by the HIP kernels of libsnerf_hip.so.

Mirrors s-nerfpp/zipnerf/gridencoder/grid.py: ``_grid_encode`` (:24-89), ``grid_encode`` (:93), ``GridEncoder``
(:96-201) -- same constructor keywords, attributes (``num_levels``, ``output_dim``, ``offsets``, ``idx``,
``grid_sizes``, ``embeddings``, ``per_level_scale`` ...), forward signature and ``state_dict`` keys.

Differences (MI355X-first, documented in DESIGN.md): the kernels write ``[B, L*C]`` directly (no ``[L,B,C]`` buffer +
permute, grid.py:47,57 / :74), take the current stream, and under autocast the table is cast to fp16 exactly like the
reference (even C only).  There is no fallback: without the library every call raises.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops

_gridtype_to_id = {'hash': 0, 'tiled': 1}
_interp_to_id = {'linear': 0, 'smoothstep': 1}


class _grid_encode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0):
        inputs = inputs.contiguous().float()
        L = offsets.shape[0] - 1
        S = np.log2(per_level_scale)
        H = base_resolution
        C = embeddings.shape[1]
        table = embeddings
        if torch.is_autocast_enabled() and C % 2 == 0:   # grid.py:41-44
            table = embeddings.to(torch.half)
        table = table.contiguous()
        outputs, dy_dx = ops.grid_encode_fwd(inputs, table, offsets, L, S, H, gridtype, align_corners, interpolation, calc_grad_inputs)
        ctx.save_for_backward(inputs, table, offsets, dy_dx if dy_dx is not None else torch.empty(0, device=inputs.device))
        ctx.dims = [L, S, H, gridtype, interpolation, dy_dx is not None, embeddings.dtype]
        # D = 3 / C in {1, 2, 4, 8} / hash / linear without input gradients (what internal/models.py:413-421 constructs, and the module's default C = 2): the forward above ran
        # the corner-cached gather and the backward takes the binned table gradient (csrc/zip.hip g3_*) instead of the atomic scatter
        ctx.fast = ops.grid_fast_ok(inputs.shape[1], C, gridtype, align_corners, interpolation, table.dtype, calc_grad_inputs)
        ctx.offsets_host = ops.grid_host_offsets(offsets) if ctx.fast and ctx.needs_input_grad[1] else None   # (read here: `offsets` is the caller's tensor object)
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, table, offsets, dy_dx = ctx.saved_tensors
        L, S, H, gridtype, interpolation, has_dd, emb_dtype = ctx.dims
        grad = grad.contiguous().to(table.dtype)
        if ctx.fast:
            out_dt = emb_dtype if emb_dtype in (torch.float32, torch.float16) else torch.float32
            # (a level-major copy of the gradient first, as grid.py:74 makes, would let the record writer read it coalesced: measured 19.4 vs
            # 18.3 ms per backward at 14.7 M points -- the copy costs more than the 8-byte strided reads, profiles/r5_y_grid_encoder_mapping_ab.txt)
            try:
                g_emb = ops.grid_encode_bwd_binned(grad, inputs, offsets, table.shape[1], L, S, H, out_dtype=out_dt, offsets_host=ctx.offsets_host)
            except ValueError:       # a level with more than 1024 row ranges (> 2^22 rows at C = 4, > 2^24 at C = 1): the atomic scatter takes it
                g_emb = ops.grid_encode_bwd(grad, inputs, table, offsets, L, S, H, gridtype, ctx.align_corners, interpolation, None)[0]
            return None, g_emb.to(emb_dtype), None, None, None, None, None, None, None
        g_emb, g_in = ops.grid_encode_bwd(grad, inputs, table, offsets, L, S, H, gridtype, ctx.align_corners, interpolation,
                                          dy_dx if has_dd else None)
        if g_in is not None:
            g_in = g_in.to(inputs.dtype)
        return g_in, g_emb.to(emb_dtype), None, None, None, None, None, None, None


grid_encode = _grid_encode.apply


class GridEncoder(nn.Module):
    """Attribute names, buffer names (`offsets`, `idx`, `grid_sizes`) and the `embeddings` parameter are the reference module's: callers
    read them (internal/models.py:413-421, 494; train_utils.py:184-203) and checkpoints store them."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype='hash', align_corners=False, interpolation='linear', init_std=1e-4,
                 device="cuda"):
        super().__init__()
        if input_dim not in (2, 3, 4, 5) or level_dim not in (1, 2, 4, 8):
            raise NotImplementedError("GridEncoder: input_dim in {2,3,4,5}, level_dim in {1,2,4,8} (the reference's instantiations, gridencoder.cu:376-399)")
        if desired_resolution is not None:               # the finest level lands on desired_resolution (grid.py:104-106)
            per_level_scale = ops.grid_per_level_scale(base_resolution, desired_resolution, num_levels)
        offsets, sizes = ops.grid_level_layout(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        vars(self).update(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, per_level_scale=per_level_scale,
                          log2_hashmap_size=log2_hashmap_size, base_resolution=base_resolution, output_dim=num_levels * level_dim,
                          gridtype=gridtype, gridtype_id=_gridtype_to_id[gridtype], interpolation=interpolation,
                          interp_id=_interp_to_id[interpolation], align_corners=align_corners, init_std=init_std,
                          max_params=2 ** log2_hashmap_size)
        rows = int(offsets[-1])
        self.register_buffer('offsets', torch.from_numpy(offsets).to(device))
        # level of every table row (the hash-decay loss segments the table with it, train_utils.py:207-211)
        self.register_buffer('idx', torch.repeat_interleave(torch.arange(num_levels), torch.from_numpy(np.diff(offsets).astype(np.int64))).to(device))
        self.register_buffer('grid_sizes', torch.from_numpy(sizes).to(device))
        self.n_params = rows * level_dim
        self.embeddings = nn.Parameter(torch.empty(rows, level_dim, device=device))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-self.init_std, self.init_std)

    def extra_repr(self):
        finest = int(self.grid_sizes[-1]) - (0 if self.align_corners else 1)
        return (f"D={self.input_dim}, L={self.num_levels}, C={self.level_dim}, base {self.base_resolution} -> {finest} "
                f"(x{self.per_level_scale:.4f} per level), table {tuple(self.embeddings.shape)}, {self.gridtype}/{self.interpolation}"
                f"{', align_corners' if self.align_corners else ''}")

    def forward(self, inputs, bound=1, cal_input_grad=False):
        inputs = (inputs + bound) / (2 * bound)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                              inputs.requires_grad, self.gridtype_id, self.align_corners, self.interp_id)
        return outputs.view(prefix_shape + [self.output_dim])

    @torch.no_grad()
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
        if self.embeddings.grad is None:
            raise ValueError('grad is None, should be called after loss.backward() and before optimizer.step()!')
        ops.grid_tv_grad(inputs.contiguous().float(), self.embeddings.data.contiguous(), self.embeddings.grad, self.offsets, weight,
                         self.num_levels, np.log2(self.per_level_scale), self.base_resolution, self.gridtype_id, self.align_corners)
