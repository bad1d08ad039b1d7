"""Device-agnostic torch restatements for the BASELINE legs of bench.py (test infrastructure, like the rest of oracle/):

  grid_features_torch   GridEncoder.forward (gridencoder/grid.py:158-177 -> gridencoder.cu:87-245) as plain torch ops (index
                        arithmetic in int64, `embeddings[index]` gathers, trilinear weights): differentiable through autograd
                        (the table gradient becomes torch's index_add), runs on the CPU and on a GPU.  Checked against oracle/grid.py
                        in tests/test_grid_oracle.py.
  sample_pdf_torch      run_nerf_helpers.sample_pdf (:336-379) with torch.searchsorted -- what the reference itself calls --
                        instead of oracle/classic.py's numpy canonical-order sums (which exist to pin bit-exact indices).

Used only by bench.py's `cpu_baseline` / `eager_baseline` legs of paths B and C: never by the product, never as the thing measured
as the build's own number."""
import numpy as np
import torch

_PRIMES = (1, 2654435761, 805459861)


def grid_features_torch(spec, emb, means):
    """features [..., L, C] of positions `means` [..., 3] in [-1, 1] (bound = 1) on the level layout `spec` (oracle.zip.GridSpec);
    emb [rows, C].  Hash grid type, linear interpolation, align_corners = False (what zipnerf constructs)."""
    x = ((means.reshape(-1, 3) + 1) / 2).to(torch.float32)
    inb = ((x >= 0) & (x <= 1)).all(-1, keepdim=True)
    outs = []
    for l in range(spec.L):
        off0, hs = int(spec.offsets[l]), int(spec.offsets[l + 1] - spec.offsets[l])
        scale = float(np.float32(np.exp2(np.float32(l) * np.float32(spec.S)) * np.float32(spec.H) - np.float32(1.0)))
        res = int(np.ceil(scale)) + 1
        pos = x * scale + 0.5
        pg = torch.floor(pos)
        fr = pos - pg
        pg = pg.to(torch.int64)
        acc = 0
        for idx in range(8):
            w = 1.0
            stride, index, dense = 1, 0, True
            corner = []
            for d in range(3):
                bit = (idx >> d) & 1
                w = w * (fr[:, d] if bit else 1 - fr[:, d])
                corner.append(pg[:, d] + bit)
            for d in range(3):                                   # gridencoder.cu:66-84
                if stride <= hs:
                    index = index + corner[d] * stride
                    stride *= res + 1
            if stride > hs:
                h = 0
                for d in range(3):
                    h = h ^ ((corner[d] * _PRIMES[d]) & 0xFFFFFFFF)
                index = h
            index = index % hs
            acc = acc + w[:, None] * emb[off0 + index]
        outs.append(torch.where(inb, acc, torch.zeros_like(acc)))
    return torch.stack(outs, 1).reshape(list(means.shape[:-1]) + [spec.L, emb.shape[1]])


def sample_pdf_torch(bins, weights, u, sum_mode="torch"):
    """-> (samples [N, Nf], inds int64 [N, Nf]); u [N, Nf] explicit uniforms"""
    w = weights.detach() + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    u = u.contiguous().to(cdf.device)
    inds = torch.searchsorted(cdf, u, right=True)
    below, above = torch.clamp(inds - 1, min=0), torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0), inds


def mip_resample_torch(s_vals, weights, u, resample_padding=0.01):
    """oracle/mip.warp_resample_s (mip.py:294-320, math_ops.py:19-76) as torch ops on the tensors' device -- the numpy restatement with
    its canonical accumulation order is the parity oracle; this form is what a PyTorch-ROCm eager run of the reference executes.
    -> (s_vals of the resampled fence posts, interval indices)"""
    dev = s_vals.device
    w = weights.detach()
    wp = torch.cat([w[..., :1], w, w[..., -1:]], -1)
    wmax = torch.maximum(wp[..., :-1], wp[..., 1:])
    w = 0.5 * (wmax[..., :-1] + wmax[..., 1:]) + resample_padding
    wsum = w.sum(-1, keepdim=True)
    pad = torch.clamp(1e-5 - wsum, min=0)
    w = w + pad / w.shape[-1]
    wsum = wsum + pad
    cdf = torch.clamp(torch.cumsum((w / wsum)[..., :-1], -1), max=1.0)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf, torch.ones_like(cdf[..., :1])], -1)
    uu = u.to(dev).expand(s_vals.shape[0], -1).contiguous()
    idx = torch.searchsorted(cdf, uu, right=True) - 1
    idx = idx.clamp(0, cdf.shape[-1] - 2)
    b0, b1 = torch.gather(s_vals, -1, idx), torch.gather(s_vals, -1, idx + 1)
    c0, c1 = torch.gather(cdf, -1, idx), torch.gather(cdf, -1, idx + 1)
    t = torch.clip(torch.nan_to_num((uu - c0) / (c1 - c0), 0.0), 0, 1)
    return b0 + t * (b1 - b0), idx
