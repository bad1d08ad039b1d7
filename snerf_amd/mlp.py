"""Tiny-MLP executors on top of the MFMA GEMM kernels (forward, data- and
weight-gradient), one class per network of the reference:

  ClassicNeRFNet  -- ``NeRF``        s-nerf/model/run_nerf_helpers.py:74-126
  MipProposalNet  -- ``proposal``    s-nerf/model/models.py:299-325
  MipNerfNet      -- ``MLP``         s-nerf/model/models.py:217-296

Design (MI355X-first, not a translation of nn.Linear chains):
  * parameters live in ONE flat fp32 arena (one Adam launch, one RCCL
    all-reduce bucket); ``nn.Parameter`` objects of the drop-in modules are
    views into it, gradients are views into a second flat arena;
  * every layer's weight is packed once per parameter version into the GEMM
    operand layout: [N padded to 128, K padded to the 128-byte tile row] in
    the compute dtype, plus the transposed pack used by the data gradient;
  * concatenations (skip connections, [bottleneck | view encoding]) are never
    materialised: producers write into column ranges of one wide buffer;
  * ReLU is fused into the forward epilogue, its mask and the bias gradient
    (column sums) into the data-gradient epilogue; weight gradients are
    accumulated with fp32 atomics straight into the flat gradient arena.
"""
import contextlib

import torch

from . import ops
from .ops import ACT_MASK, ACT_NONE, ACT_RELU, gran, roundup


class ParamArena:
    """Flat fp32 parameter + gradient storage with named views."""

    def __init__(self, shapes, device):
        self.names = [n for n, _ in shapes]
        self.shapes = dict(shapes)
        offs, o = {}, 0
        for n, s in shapes:
            numel = 1
            for d in s:
                numel *= d
            offs[n] = (o, numel)
            o += roundup(numel, 4)  # keep every view 16-byte aligned
        self.numel = roundup(o, 4)
        self._offs = offs
        self.epoch = 0
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.p = {n: self.flat[a:a + c].view(self.shapes[n]) for n, (a, c) in offs.items()}
        self.g = {n: self.grad[a:a + c].view(self.shapes[n]) for n, (a, c) in offs.items()}

    def load(self, sd: dict, prefix: str = ""):
        with torch.no_grad():
            for n in self.names:
                self.p[n].copy_(sd[prefix + n].to(self.flat.device, torch.float32))

    def span(self, prefix: str):
        """[start, end) of the flat arena covered by the parameters whose name starts with `prefix` (they are laid out together)."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefix)]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError(f"parameters with prefix {prefix!r} are not one contiguous block of the arena")
        a = self._offs[self.names[idx[0]]][0]
        b = self._offs[self.names[idx[-1] + 1]][0] if idx[-1] + 1 < len(self.names) else self.numel
        return a, b

    def index_image(self, name):
        """float64 tensor of the parameter's shape holding (flat arena position + 2) of every element: what the packing code slices,
        transposes and concatenates once to learn where each element of a packed operand comes from (0 = padding, 1 = constant one)."""
        a, c = self._offs[name]
        return (torch.arange(a, a + c, dtype=torch.float64, device=self.flat.device) + 2.0).view(self.shapes[name])

    def bump(self):
        """Call after the parameters were modified behind torch's back (fused Adam kernel, RCCL broadcast)."""
        self.epoch += 1

    def version(self) -> int:
        return self.flat._version + self.epoch


def _w2(t):
    return t if t.dim() == 2 else t.view(1, -1)


# ---------------------------------------------------------------------------------------------------------------------------------
# Weight stream of the fused register-resident MLP kernels (csrc/fmlp.hip)
# ---------------------------------------------------------------------------------------------------------------------------------
# Order in which the 16 reduction indices of a k-step sit in the lanes when the B operand of an MFMA is the previous layer's
# accumulator (lane half 0 holds outputs {0-3, 8-11}, lane half 1 {4-7, 12-15} of every 16): position p of the fragment carries
# logical index FMLP_PERM[p].  Layers fed from memory keep the natural order.
FMLP_PERM = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)
FMLP_CHUNK = 16


def fmlp_pack(layers, device):
    """layers = [(W fp32 [N,K], bias fp32 [N] or None, segments[, k_major])] in the order the kernel consumes them, segments =
    [(first weight column, columns, from_accumulator)] in the order of the kernel's input segments; `k_major`: the layer's fragments
    are consumed k-step by k-step over ALL its blocks (the colour head's wide first layer keeps its four accumulators live and
    streams the input row once) instead of block by block.
    -> (stream bf16 [n_frags, 512]: 1 KiB MFMA A-operand fragments -- lane (n = lane & 31, half = lane >> 5) holds the 8 reduction
    positions 8 half .. 8 half + 7 of output n of its 32-output block -- block by block, k-step by k-step, padded to whole chunks;
    bias fp32 [n_blocks * 32])."""
    perm = torch.tensor(FMLP_PERM, device=device)
    frags, biases = [], []
    dt = layers[0][0].dtype if layers[0][0].dtype == torch.float64 else torch.float32   # float64: index images (_Net._build_plan)
    for W, b, segs, *opt in layers:
        kmajor = bool(opt and opt[0])
        W = _w2(W).detach().to(device, dt)
        N = W.shape[0]
        NB = (N + 31) // 32
        per_seg = []
        for c0, cnt, from_acc in segs:
            ks = (cnt + 15) // 16
            Ws = torch.zeros(NB * 32, ks * 16, dtype=dt, device=device)
            Ws[:N, :cnt] = W[:, c0:c0 + cnt]
            Ws = Ws.view(NB * 32, ks, 16)
            if from_acc:
                Ws = Ws[:, :, perm]
            # [NB, 32 n, ks, 2 halves, 8] -> [NB, ks, half, n, 8]: one fragment = 64 lanes x 8 values, lane = half * 32 + n
            per_seg.append(Ws.reshape(NB, 32, ks, 2, 8).permute(0, 2, 3, 1, 4).reshape(NB, ks, 512))
        fr = torch.cat(per_seg, 1)                                     # [NB, k-steps, 512]
        frags.append((fr.permute(1, 0, 2) if kmajor else fr).reshape(-1, 512))
        bb = torch.zeros(NB * 32, dtype=dt, device=device)
        if b is not None:
            bb[:N] = b.detach().to(device, dt).reshape(-1)
        biases.append(bb)
    stream = torch.cat(frags, 0)
    pad = (-stream.shape[0]) % FMLP_CHUNK
    if pad:
        stream = torch.cat([stream, torch.zeros(pad, 512, dtype=dt, device=device)], 0)
    if dt == torch.float64:
        return stream.contiguous(), torch.cat(biases).contiguous()
    return stream.to(torch.bfloat16).contiguous(), torch.cat(biases).contiguous()


LO_FLAG = float(1 << 40)      # added to an index-image entry: "the low part of the split-bf16 pair of this parameter" (see split_w_image)


def split_w_image(img):
    """index image of a GEMM weight operand [N, K] (K % 64 == 0) -> image of its split-bf16 form [N, 3 K]: per 64 columns
    [hi | hi | lo] (csrc/gemm.hip, GemmNT::split); padding stays padding, a constant one has no low part"""
    N, K = img.shape
    t = img.reshape(N, K // 64, 1, 64)
    lo = torch.where(t > 1.5, t + LO_FLAG, torch.zeros_like(t))
    return torch.cat([t, t, lo], 2).reshape(N, 3 * K).contiguous()


class _PackPlan:
    """Persistent packed operands of one network + the gather map that refreshes them from the parameter arena."""

    def __init__(self, flat):
        self.flat = flat
        self.items = {}                    # dtype -> [(index image, offset)]
        self.size = {}

    def add(self, img, dtype):
        off = self.size.get(dtype, 0)
        self.items.setdefault(dtype, []).append((img, off))
        self.size[dtype] = off + roundup(img.numel(), 128)        # every operand 256-byte aligned inside its pool
        return (dtype, off, tuple(img.shape))

    @staticmethod
    def _transposed_tiles(k2, off):
        """k2: int64 [R, C] arena positions of one packed operand (negative: padding / ones; bit 30: split-bf16 low parts).  -> (int32 [n, 4] tile
        descriptors of snerf_gather_pack_tiles, bool [R, C] mask of the covered elements) for the 16 x 64 tiles with
        k2[r0 + i, c0 + j] = base + j * stride + i, base % 4 == stride % 4 == 0, stride >= 16 -- the transposed weight images."""
        R, C = k2.shape
        R16, C64 = R // 16 * 16, C // 64 * 64
        if R16 == 0 or C64 == 0:
            return None, None
        t = k2[:R16, :C64].reshape(R16 // 16, 16, C64 // 64, 64)
        base = t[:, 0, :, 0]
        s = t[:, 0, :, 1] - base
        i = torch.arange(16, device=k2.device).view(1, 16, 1, 1)
        j = torch.arange(64, device=k2.device).view(1, 1, 1, 64)
        ok = (t == base[:, None, :, None] + i + j * s[:, None, :, None]).all(3).all(1)
        ok &= (base >= 0) & (base % 4 == 0) & (s % 4 == 0) & (s >= 16) & (base + 63 * s + 15 < (1 << 30))
        if not bool(ok.any()):
            return None, None
        tr, tc = torch.nonzero(ok, as_tuple=True)
        tiles = torch.stack([off + tr * 16 * C + tc * 64, base[tr, tc], s[tr, tc], torch.full_like(tr, C)], 1).to(torch.int32)
        mask = torch.zeros(R, C, dtype=torch.bool, device=k2.device)
        mask[:R16, :C64] = ok[:, None, :, None].expand(-1, 16, -1, 64).reshape(R16, C64)
        return tiles, mask

    def finish(self):
        self.pools, self.maps, self.tiles = {}, {}, {}
        for dtype, items in self.items.items():
            n = self.size[dtype]
            idx = torch.full((n,), -1, dtype=torch.int32, device=self.flat.device)
            tl = []
            for img, off in items:
                v = img.reshape(-1)
                assert bool(((v == v.round()) & (v >= 0)).all()), "pack() may only copy parameters, zeros and ones"
                lo = v >= LO_FLAG                                  # split-bf16 operands: the low part bf16(x - bf16(x)) of parameter x
                v = torch.where(lo, v - LO_FLAG, v)
                k = (v - 2.0).to(torch.int32)                      # image value = arena position + 2; 0 = padding, 1 = constant one
                k = torch.where(lo, k | (1 << 30), k)
                minus1, minus2 = torch.full_like(k, -1), torch.full_like(k, -2)
                k = torch.where(v == 0, minus1, torch.where(v == 1, minus2, k))
                if img.dim() == 2 and dtype in (torch.bfloat16, torch.float16) and n < (1 << 31):
                    # transposed weight images (W^T of the data-gradient GEMMs) leave the element-wise gather: 16 x 64 tiles, one 64-byte source line per lane
                    tiles, mask = self._transposed_tiles(k.view(img.shape).long(), off)
                    if tiles is not None:
                        tl.append(tiles)
                        k = torch.where(mask.reshape(-1), torch.full_like(k, -3), k)
                idx[off:off + v.numel()] = k
            self.pools[dtype] = torch.zeros(n, dtype=dtype, device=self.flat.device)
            self.maps[dtype] = idx
            self.tiles[dtype] = torch.cat(tl, 0).contiguous() if tl else None
        self.items = None
        self.post = []                     # fp16 + fp8 weights: (fp32 image [N, K] in the fp32 pool, its [N, 2 K] split operand), converted after the gather

    def view(self, handle):
        dtype, off, shape = handle
        n = 1
        for d in shape:
            n *= d
        return self.pools[dtype][off:off + n].view(shape)

    def refresh(self):
        half = [d for d in self.maps if d in (torch.bfloat16, torch.float16)]
        if len(half) == 1 and torch.float32 in self.maps and len(self.maps) == 2:
            # the usual case -- one 16-bit operand pool + the fp32 pool (biases, fused-kernel tables) -- is ONE launch
            d = half[0]
            ops.gather_pack_pair(self.flat, self.maps[d], self.pools[d], self.tiles[d], self.maps[torch.float32], self.pools[torch.float32])
        else:
            for dtype, idx in self.maps.items():
                ops.gather_pack(self.flat, idx, self.pools[dtype], self.tiles[dtype])
        for src, dst in self.post:
            ops.split8_cast(src, src.shape[1], dst, src.shape[1], weight=True)


class _Net:
    """Shared machinery: packing, buffer helpers, layer calls."""

    def __init__(self, arena: ParamArena, prefix: str, dt: int, variant: int = 8, bwd_plain: bool = False):
        self.a, self.pre, self.dt, self.variant = arena, prefix, dt, variant
        self.g = gran(dt)
        self.km = 2 if dt in ops.SPLIT_DTS else 1        # physical columns per logical column of an activation buffer (split-bf16: hi / lo interleaved)
        self.tdt = ops.torch_dtype(dt)
        # compute="bf16x3_fwd": the FORWARD in split-bf16 (three MFMA passes: renders inside the fp32 contract), the BACKWARD as ONE plain bf16
        # pass per product -- data gradients on plain bf16 buffers with plainly packed W^T, weight gradients from the hi half of the saved
        # split activations (snerf_linear_wgrad variant bit 14), ReLU masks from the forward's bit masks or the saved hi halves (bit 14 of
        # snerf_linear_fwd).  The gradients then carry the bf16 mode's rounding (of exact forward values), at a third of the split backward.
        # compute="f16f8" (dtype F16F8: fp16 tiles + e4m3 correction tiles, two pass-equivalents) has a forward only: its backward is ALWAYS the
        # plain one, in fp16 -- on gradients scaled by a power of two chosen per call from their largest entry (fp16 has 5 exponent bits: the
        # raw gradients of a mean-reduced loss sit in its subnormals), accumulated into a scratch gradient arena and folded into the real one
        # with the inverse scale (_scaled_backward).
        self.bwd_plain = (bool(bwd_plain) and dt == ops.BF16X3) or dt == ops.F16F8
        self.bwd_dt = ops.F16 if dt == ops.F16F8 else ops.BF16
        self._in_plain_bwd = False
        self._g = arena.g                                 # where gW / gB point: the arena's gradient views, or the scratch ones of a scaled backward
        self.dev = arena.flat.device
        self._packed_version = -1
        self.deterministic = False        # bit-reproducible gradients: partial tiles folded in a fixed order instead of fp32 atomics
        self.version_fn = arena.version   # drop-in modules override this with their nn.Parameter versions
        self.fw, self.fb, self.tw = {}, {}, {}
        self._rec = None                  # while a plan is being recorded: [(index image, dtype of the operand)]
        self._plans = {}                  # what was packed -> _PackPlan

    # ---- packing -------------------------------------------------------------
    # The pack() methods below DESCRIBE the operand layouts with ordinary slicing code.  They run once per network (and per train
    # flag) on index images of the parameters (ParamArena.index_image); the result is one int32 gather map per network, and every
    # later refresh (the parameters change each optimiser step) is ONE snerf_gather_pack launch per dtype into persistent buffers.
    def W(self, name):
        if self._rec is not None:
            return _w2(self.a.index_image(self.pre + name + ".weight"))
        return _w2(self.a.p[self.pre + name + ".weight"])

    def B(self, name):
        if self._rec is not None:
            return self.a.index_image(self.pre + name + ".bias")
        return self.a.p[self.pre + name + ".bias"]

    def _zeros(self, *shape, f32=False, plain=False):
        """operand buffer of a pack() method (recorded: its content becomes part of the network's gather map).  `plain`: a compute-dtype
        operand that stays ONE bf16 value per parameter in a split-bf16 network (the backward's W^T under `bwd_plain`)"""
        assert self._rec is not None, "pack() runs only while a plan is recorded"
        t = torch.zeros(*shape, dtype=torch.float64, device=self.dev)
        self._rec.append((t, torch.float32 if f32 else self.tdt))
        if plain:
            self._rec_plain.add(id(t))
        return t

    @contextlib.contextmanager
    def _bwd(self):
        """the body of a backward(): under `bwd_plain` every helper below (buf / cs / head_grad / dgrad / wgrad / input_grad) works on plain
        bf16 buffers while it runs; a no-op otherwise"""
        if not self.bwd_plain:
            yield
            return
        saved = (self.dt, self.km, self.tdt, self._in_plain_bwd)
        self.dt, self.km, self.tdt, self._in_plain_bwd = self.bwd_dt, 1, ops.torch_dtype(self.bwd_dt), True
        try:
            yield
        finally:
            self.dt, self.km, self.tdt, self._in_plain_bwd = saved

    def _fp16_backward(self):
        """fp16 gradient GEMMs (behind the fp16 + fp8 forward, or compute="fp16" on the mip path): they run on scaled gradients (_scaled_backward)"""
        return (self.bwd_plain and self.bwd_dt == ops.F16) or self.dt == ops.F16

    def _scaled_backward(self, grads, run, on_done=None):
        """fp16 backward (behind the fp16 + fp8 forward): `grads` = the fp32 gradients entering the network (None entries allowed), `run(scaled
        grads)` the backward body.  The gradients are multiplied by S = 2^k with max |g| S in [512, 1024) -- device-side, no read-back --, the body
        accumulates S x (parameter gradients) into a zeroed scratch copy of this network's span of the gradient arena, which is then added to
        the real one times 1 / S; tensors `run` returns (gradients w.r.t. the encodings) are scaled back too."""
        live = [g for g in grads if g is not None]
        amax = torch.stack([g.abs().amax() for g in live]).amax()
        S = torch.exp2(torch.floor(torch.log2(1024.0 / amax.clamp(min=1e-30)))).clamp(2.0 ** -24, 2.0 ** 40)
        a, b = self.a.span(self.pre)
        if getattr(self, "_gscratch", None) is None:
            self._gscratch = torch.zeros(b - a, dtype=torch.float32, device=self.dev)
            self._gs_views = {n: self._gscratch[o - a:o - a + c].view(self.a.shapes[n]) for n, (o, c) in self.a._offs.items() if n.startswith(self.pre)}
        else:
            self._gscratch.zero_()
        self._g = self._gs_views
        try:
            out = run([None if g is None else g * S for g in grads])
        finally:
            self._g = self.a.g
        inv = 1.0 / S
        self.a.grad[a:b].addcmul_(self._gscratch, inv.expand_as(self._gscratch))
        if on_done is not None:
            on_done([""])
        unscale = lambda t: t * inv if torch.is_tensor(t) else t
        return tuple(unscale(t) for t in out) if isinstance(out, tuple) else unscale(out)

    def _build_plan(self, fill):
        """Run `fill()` (a pack method writing operands obtained from _zeros / returned as extra (image, dtype) pairs) on index images
        and turn every recorded operand into a view of a persistent pool + its slice of the gather map."""
        assert self.a.numel < (1 << 31) - 2
        self._rec, self._rec_plain = [], set()
        try:
            extra = fill() or []
            rec = self._rec + list(extra)
            plain = self._rec_plain
        finally:
            self._rec = None
        plan = _PackPlan(self.a.flat)
        if self.dt == ops.F16F8:  # GEMM weights [N, K]: gathered as fp32 images, then converted into the fp16 + fp8 operand [N, 2 K] (ops.split8_cast)
            is_w = lambda img, dtype: dtype == self.tdt and img.dim() == 2 and id(img) not in plain
            handles = [(id(img), plan.add(img, torch.float32 if is_w(img, dtype) else dtype), is_w(img, dtype)) for img, dtype in rec]
            plan.finish()
            real = {}
            for i, h, w in handles:
                real[i] = plan.view(h)
                if w:
                    assert real[i].shape[1] % 64 == 0
                    dst = torch.zeros(real[i].shape[0], 2 * real[i].shape[1], dtype=self.tdt, device=self.dev)
                    plan.post.append((real[i], dst))
                    real[i] = dst
            handles = None
        elif self.km == 2:        # every compute-dtype operand of the per-layer plans is a GEMM weight [N, K]: its [hi | hi | lo] form
            handles = [(id(img), plan.add(split_w_image(img) if dtype == self.tdt and img.dim() == 2 and id(img) not in plain else img, dtype))
                       for img, dtype in rec]
        else:
            handles = [(id(img), plan.add(img, dtype)) for img, dtype in rec]
        if handles is not None:
            plan.finish()
            real = {i: plan.view(h) for i, h in handles}

        def swap(v):
            if isinstance(v, torch.Tensor):
                return real.get(id(v), v)
            if isinstance(v, dict):
                return {k: swap(x) for k, x in v.items()}
            if isinstance(v, tuple):
                return tuple(swap(x) for x in v)
            return v
        return plan, swap

    def gW(self, name):
        return _w2(self._g[self.pre + name + ".weight"])

    def gB(self, name):
        return self._g[self.pre + name + ".bias"]

    def _refresh_fused(self, pack_fused, key="fused", dtype=torch.bfloat16):
        """weight stream / bias table of a fused kernel (fmlp_pack's layout), refreshed by the same gather.  -> (stream, bias)"""
        if key not in self._plans:
            def fill():
                st, bi = pack_fused()                      # index images (float64) in record mode
                self._fimg = (st, bi)
                return [(st, dtype), (bi, torch.float32)]
            plan, swap = self._build_plan(fill)
            self._plans[key] = (plan, swap(self._fimg))
            del self._fimg
        plan, sb = self._plans[key]
        plan.refresh()
        if key == "fused":
            self.fstream, self.fbias = sb
        return sb

    def _pack_fwd(self, key, name, segs, kbuf):
        W = self.W(name)
        N = W.shape[0]
        out = self._zeros(roundup(N, 128), kbuf)
        for bc, wc, cnt in segs:
            out[:N, bc:bc + cnt] = W[:, wc:wc + cnt]
        b = self._zeros(roundup(N, 128), f32=True)
        b[:N] = self.B(name)
        self.fw[key], self.fb[key] = out, b

    def _pack_dgrad(self, key, parts, wc, cnt):
        """W^T restricted to weight columns [wc, wc+cnt) for one or more layers that read the same
        activation: rows = those input columns (padded to 128), cols = concat of each layer's outputs
        padded to the tile granularity (matches the [dZ_a | dZ_b] gradient buffer)."""
        cols = sum(roundup(self.W(n).shape[0], self.g) for n in parts)
        out = self._zeros(roundup(cnt, 128), cols, plain=self.bwd_plain)
        c = 0
        for n in parts:
            W = self.W(n)
            out[:cnt, c:c + W.shape[0]] = W[:, wc:wc + cnt].t()
            c += roundup(W.shape[0], self.g)
        self.tw[key] = out

    def _pack_dgrad_cols(self, key, parts, cnt):
        """like _pack_dgrad with a weight-column offset per layer: parts = [(layer, first weight column)]; used for the gradient
        w.r.t. an INPUT encoding that several layers read at different column positions (rows = the encoding's columns)."""
        cols = sum(roundup(self.W(n).shape[0], self.g) for n, _ in parts)
        out = self._zeros(roundup(cnt, 128), cols, plain=self.bwd_plain)
        c = 0
        for n, wc in parts:
            W = self.W(n)
            out[:cnt, c:c + W.shape[0]] = W[:, wc:wc + cnt].t()
            c += roundup(W.shape[0], self.g)
        self.tw[key] = out

    def input_grad(self, key, dZ, K, width):
        """fp32 [M, width] = dZ[:, :K] @ tw[key]^T: the data gradient that reaches an input encoding (pose refinement only)."""
        out = self.buf(dZ.shape[0], width, f32=True)
        ops.linear_fwd(dZ, self.tw[key], None, out, K, width, ACT_NONE, self.dt, out_f32=True, variant=self.variant)
        return out

    def colsum(self, x, C, out):
        ops.colsum_f32(x, C, out, deterministic=self.deterministic)

    def ensure_packed(self, train: bool):
        # ReLU bit masks written by this forward's layers (keyed by the activation view), read by the data-gradient GEMMs
        self._bits = {} if train else None
        v = self.version_fn()
        key = "train" if train else "infer"
        if self._packed_version != (v, key) and not (key == "infer" and self._packed_version == (v, "train")):
            with torch.no_grad():
                if key not in self._plans:
                    self.fw, self.fb, self.tw = {}, {}, {}
                    plan, swap = self._build_plan(lambda: self.pack(train))
                    self._plans[key] = (plan, swap(self.fw), swap(self.fb), swap(self.tw))
                plan, self.fw, self.fb, self.tw = self._plans[key]
                plan.refresh()
            self._packed_version = (v, key)

    # ---- kernels ---------------------------------------------------------------
    def buf(self, M, cols, f32=False):
        """activation buffer of `cols` LOGICAL columns (split-bf16: twice as many physical ones)"""
        return torch.empty(M, cols if f32 else cols * self.km, dtype=torch.float32 if f32 else self.tdt, device=self.dev)

    def cs(self, t, a, b=None):
        """logical column range [a, b) of an activation buffer (multiples of the tile granularity): a plain slice, of twice the width in
        the split-bf16 layout (64 logical columns = one 128-column physical group)"""
        return t[:, a * self.km:(None if b is None else b * self.km)]

    def fwd(self, key, A, K, Y, n_store, act=ACT_RELU, out_f32=False):
        W = self.fw[key]
        bits = getattr(self, "_bits", None)
        if act == ACT_RELU and bits is not None and not out_f32 and ops.relu_bits_ok(A, W, Y, K, n_store, self.dt, self.variant):
            # training: the ReLU also leaves a 1-bit mask (1/16 of the activation bytes) for the data gradient of the next layer
            words = torch.empty(ops.mask_bits_words(A.shape[0], W.shape[0]), dtype=torch.int32, device=A.device)
            ops.linear_fwd(A, W, self.fb[key], Y, K, n_store, ops.ACT_RELU_BITS, self.dt, aux=words, variant=self.variant)
            bits[(Y.data_ptr(), Y.shape[0])] = (words, W.shape[0])
            return
        ops.linear_fwd(A, W, self.fb[key], Y, K, n_store, act, self.dt, out_f32=out_f32, variant=self.variant)

    def dgrad(self, key, dZ, K, dX, n_store, mask=None, colsum=None):
        W = self.tw[key]
        bits = getattr(self, "_bits", None)
        if mask is not None and bits:
            ent = bits.get((mask.data_ptr(), mask.shape[0]))
            if ent is not None and ent[1] == W.shape[0] and ops.relu_bits_ok(dZ, W, dX, K, n_store, self.dt, self.variant, consumer=True):
                ops.linear_fwd(dZ, W, None, dX, K, n_store, ops.ACT_MASK_BITS, self.dt, aux=ent[0], colsum=colsum, variant=self.variant,
                               deterministic=self.deterministic)
                return
        act = ACT_MASK if mask is not None else ACT_NONE
        if self.deterministic and colsum is not None and not ops.fast_epilogue_ok(dX, n_store, self.dt, aux=mask, act=act):
            # deterministic mode and a destination the 16-byte epilogue does not cover (an unaligned view, n_store % 8 != 0): the
            # direct-store epilogue could only add its column sums with atomics (snerf_linear_fwd refuses the combination), so the
            # bias gradient comes from a fixed-order column sum of the stored data gradient instead
            ops.linear_fwd(dZ, W, None, dX, K, n_store, act, self.dt, aux=mask, variant=self.variant, aux_split=getattr(self, "_in_plain_bwd", False) and mask is not None)
            ops.colsum_wide_f32(dX[:, :n_store].float(), n_store, colsum, deterministic=True)     # (any width; colsum_f32_det stops at 8 columns)
            return
        ops.linear_fwd(dZ, W, None, dX, K, n_store, act, self.dt,
                       aux=mask, colsum=colsum, variant=self.variant, deterministic=self.deterministic,
                       aux_split=getattr(self, "_in_plain_bwd", False) and mask is not None)      # (the mask source is an activation the split forward saved)

    def wgrad(self, name, dZ, X, n_valid, k_valid, wcol=0):
        gw = self.gW(name)
        ops.linear_wgrad(dZ, X, gw[:, wcol:], n_valid, k_valid, self.dt, variant=3,   # 2: 8-phase 256x256 tiles where they fit, else 1: transposing LDS reads
                         deterministic=self.deterministic, x_split_hi=self._in_plain_bwd)

    def head_grad(self, d_raw_f32, C):
        """fp32 head gradient [M,C] -> compute-dtype buffer padded to the tile granularity."""
        M = d_raw_f32.shape[0]
        out = self.buf(M, roundup(C, self.g))
        ops.cast_pad(d_raw_f32, C, out, roundup(C, self.g), self.dt)
        return out


# =============================================================================
# classic NeRF (path B)
# =============================================================================
class ClassicNeRFNet(_Net):
    """8 x W trunk, cat([input_pts, h]) after layer `skip`, alpha / feature / views / rgb heads.
    Buffers: E [M, Pw] embedding (63 + pad); SK [M, Pw + W] = [embedding | layer-skip output];
    V [M, W + Vw] = [feature | view embedding (27 + pad)]; OUT [M,4] fp32 = [rgb | alpha]."""

    def __init__(self, arena, prefix, dt, D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,), variant=8, alpha_head=True, output_ch=0, bwd_plain=False):
        """`alpha_head=False`: the ``NeRF_RGB`` variant (run_nerf_helpers.py:157-212) -- no alpha_linear; column 3 of the output is left
        for the caller (the frozen alpha model's density).  `output_ch` > 0: the ``use_viewdirs=False`` network (run_nerf_helpers.py:100-101,
        122-124): the trunk's output goes through ``output_linear`` (W -> output_ch) alone; ``views_linears.0`` exists (the reference
        constructs it regardless, :90) but is never evaluated."""
        super().__init__(arena, prefix, dt, variant, bwd_plain)
        self.output_ch = int(output_ch)
        assert len(skips) == 1 and 0 <= skips[0] < D - 1 and W % self.g == 0 and (W // 2) % self.g == 0
        self.D, self.Wd, self.ic, self.icv, self.skip = D, W, input_ch, input_ch_views, skips[0]
        self.Pw, self.Vw = roundup(input_ch, self.g), roundup(input_ch_views, self.g)
        self.alpha_head = bool(alpha_head)
        self.fused = True                 # inference through the fused register-resident kernel where it applies (fused_ok)
        self.fused_embed = True           # ... with the positional encodings computed inside it (False: separate embedding kernel)

    @staticmethod
    def param_shapes(D=8, W=256, input_ch=63, input_ch_views=27, skips=(4,), alpha_head=True, output_ch=0):
        out = []
        for i in range(D):
            k = input_ch if i == 0 else (W + input_ch if (i - 1) in skips else W)
            out += [(f"pts_linears.{i}.weight", (W, k)), (f"pts_linears.{i}.bias", (W,))]
        out += [("views_linears.0.weight", (W // 2, input_ch_views + W)), ("views_linears.0.bias", (W // 2,))]
        if output_ch > 0:
            return out + [("output_linear.weight", (output_ch, W)), ("output_linear.bias", (output_ch,))]
        out += [("feature_linear.weight", (W, W)), ("feature_linear.bias", (W,))]
        if alpha_head:
            out += [("alpha_linear.weight", (1, W)), ("alpha_linear.bias", (1,))]
        out += [("rgb_linear.weight", (3, W // 2)), ("rgb_linear.bias", (3,))]
        return out

    def pack(self, train):
        W, ic, Pw = self.Wd, self.ic, self.Pw
        for i in range(self.D):
            n = f"pts_linears.{i}"
            if i == 0:
                self._pack_fwd(n, n, [(0, 0, ic)], Pw)
            elif i == self.skip + 1:
                self._pack_fwd(n, n, [(0, 0, ic), (Pw, ic, W)], Pw + W)     # param [pts | h] -> buffer [pts pad | h]
            else:
                self._pack_fwd(n, n, [(0, 0, W)], W)
        if self.output_ch > 0:
            self._pack_fwd("out", "output_linear", [(0, 0, W)], W)
            if train:
                self._pack_dgrad("out", ["output_linear"], 0, W)
                for i in range(1, self.D):
                    n = f"pts_linears.{i}"
                    self._pack_dgrad(n, [n], ic if i == self.skip + 1 else 0, W)
            return
        if self.alpha_head:
            self._pack_fwd("alpha", "alpha_linear", [(0, 0, W)], W)
        self._pack_fwd("feature", "feature_linear", [(0, 0, W)], W)
        self._pack_fwd("views", "views_linears.0", [(0, 0, W + self.icv)], W + self.Vw)
        self._pack_fwd("rgb", "rgb_linear", [(0, 0, W // 2)], W // 2)
        if train:
            self._pack_dgrad("rgb", ["rgb_linear"], 0, W // 2)
            self._pack_dgrad("views", ["views_linears.0"], 0, W)             # only the feature columns need a gradient
            self._pack_dgrad("fa", ["feature_linear"] + (["alpha_linear"] if self.alpha_head else []), 0, W)
            for i in range(1, self.D):
                n = f"pts_linears.{i}"
                self._pack_dgrad(n, [n], ic if i == self.skip + 1 else 0, W)

    def fused_ok(self):
        """the register-resident fused kernel (csrc/fmlp.hip) covers the S-NeRF configuration: bf16, 8 x 256, skip after layer 4,
        63 + 27 input channels, alpha head"""
        return (self.fused and self.dt == ops.BF16 and self.D == 8 and self.Wd == 256 and self.skip == 4 and self.ic == 63 and self.icv == 27
                and self.alpha_head and self.output_ch == 0)

    def _pack_fused(self):
        W, ic = self.Wd, self.ic
        L = []
        for i in range(self.D):
            n = f"pts_linears.{i}"
            if i == 0:
                segs = [(0, ic, False)]
            elif i == self.skip + 1:
                segs = [(0, ic, False), (ic, W, True)]               # cat([input_pts, h]): pts from memory, h from the accumulators
            else:
                segs = [(0, W, True)]
            L.append((self.W(n), self.B(n), segs))
        L.append((self.W("alpha_linear"), self.B("alpha_linear"), [(0, W, True)]))
        L.append((self.W("feature_linear"), self.B("feature_linear"), [(0, W, True)]))
        L.append((self.W("views_linears.0"), self.B("views_linears.0"), [(0, W, True), (W, self.icv, False)]))
        L.append((self.W("rgb_linear"), self.B("rgb_linear"), [(0, W // 2, True)]))
        return fmlp_pack(L, self.dev)

    def _fused_ready(self):
        v = self.version_fn()
        if getattr(self, "_fused_version", None) != v:
            with torch.no_grad():
                self._refresh_fused(self._pack_fused)
            self._fused_version = v

    def chain_ok(self):
        """the data-gradient chain as ONE launch (csrc/fmlp.hip fchain_bwd_kernel); its bias gradients meet in LDS atomics, so the
        deterministic mode keeps the per-layer kernels"""
        return self.fused_ok() and getattr(self, "fused_chain", True) and not self.deterministic

    def _pack_chain(self):
        """transposed weights in the order the data-gradient chain consumes them (no biases)"""
        W, ic = self.Wd, self.ic
        L = [(self.W("rgb_linear").t(), None, [(0, 3, False)]),                                      # d rgb -> d views_linears.0
             (self.W("views_linears.0")[:, :W].t(), None, [(0, W // 2, True)]),                      # -> d feature (the view columns: no gradient)
             (torch.cat([self.W("feature_linear").t(), self.W("alpha_linear").t()], 1), None, [(0, W, True), (W, 1, False)])]   # [d feature | d alpha] -> d pts_linears.7
        for i in range(self.D - 1, 0, -1):
            Wi = self.W(f"pts_linears.{i}")
            L.append(((Wi[:, ic:] if i == self.skip + 1 else Wi).t(), None, [(0, W, True)]))         # d pts_linears.i -> d pts_linears.(i-1)
        return fmlp_pack(L, self.dev)

    def _chain_stream(self):
        v = self.version_fn()
        if getattr(self, "_chain_version", None) != v:
            with torch.no_grad():
                self._chain = self._refresh_fused(self._pack_chain, "chain")
            self._chain_version = v
        return self._chain[0]

    def forward_fused_train(self, pts, viewdirs, S):
        """Training forward: the exact embedding kernel + ONE fused kernel that also stores the hidden activations in the buffers the
        per-layer backward reads (same `saved` structure as the per-layer forward)."""
        self._fused_ready()
        self.ensure_packed(True)
        M, W, Pw = pts.shape[0], self.Wd, self.Pw
        E, SK, V = self.buf(M, Pw), self.buf(M, Pw + W), self.buf(M, W + self.Vw)
        ops.classic_embed(pts, viewdirs, S, (self.ic - 3) // 6, (self.icv - 3) // 6, E, SK[:, :Pw], Pw, V[:, W:], self.Vw, self.dt)
        ys = [SK[:, Pw:] if i == self.skip else self.buf(M, W) for i in range(self.D)]
        HV, OUT = self.buf(M, W // 2), self.buf(M, 4, f32=True)
        words = [torch.empty(ops.mask_bits_words(M, W), dtype=torch.int32, device=self.dev) for _ in range(self.D)]
        words.append(torch.empty(ops.mask_bits_words(M, W // 2), dtype=torch.int32, device=self.dev))     # views_linears.0
        ops.fmlp_classic_train_fwd(E, V[:, W:], self.fstream, self.fbias, OUT, ys + [V[:, :W], HV], words)
        for y, w in zip(ys, words):                                   # the data-gradient GEMMs take their ReLU masks from the bits
            self._bits[(y.data_ptr(), M)] = (w, W)
        acts, x, k = [], E, Pw
        for i in range(self.D):
            acts.append((x, k, ys[i]))
            x, k = (SK, Pw + W) if i == self.skip else (ys[i], W)
        return OUT, (acts, V, HV, SK, E, words)

    def forward_fused(self, pts, viewdirs, S):
        """Inference: embedding kernel + ONE fused kernel for the whole network (activations never leave the registers)."""
        self._fused_ready()
        M = pts.shape[0]
        OUT = self.buf(M, 4, f32=True)
        if self.fused_embed and M < (1 << 31):
            ops.fmlp_classic_pts_fwd(pts, viewdirs, S, self.fstream, self.fbias, OUT)     # run_network = one launch
            return OUT
        E, VE = self.buf(M, 64), self.buf(M, 64)
        ops.classic_embed(pts, viewdirs, S, (self.ic - 3) // 6, (self.icv - 3) // 6, E, None, 64, VE, 64, self.dt)
        ops.fmlp_classic_fwd(E, VE, self.fstream, self.fbias, OUT)
        return OUT

    def forward(self, pts, viewdirs, S, keep: bool):
        """pts [M,3] fp32, viewdirs [N,3] -> raw [M,4] fp32 (+ saved activations when keep)."""
        if self.fused_ok():
            if not keep:
                return self.forward_fused(pts, viewdirs, S), None
            return self.forward_fused_train(pts, viewdirs, S)
        self.ensure_packed(keep)
        M, W, Pw = pts.shape[0], self.Wd, self.Pw
        E = self.buf(M, Pw)
        SK = self.buf(M, Pw + W)
        noviews = self.output_ch > 0
        V = None if noviews else self.buf(M, W + self.Vw)
        cs = self.cs                                          # (logical column ranges: twice as wide physically in the split layouts)
        if noviews:
            ops.classic_embed(pts, None, S, (self.ic - 3) // 6, 0, E, cs(SK, 0, Pw), Pw, None, 0, self.dt)
        else:
            ops.classic_embed(pts, viewdirs, S, (self.ic - 3) // 6, (self.icv - 3) // 6, E, cs(SK, 0, Pw), Pw, cs(V, W), self.Vw, self.dt)
        acts = []
        x, k = E, Pw
        pp = [self.buf(M, W), self.buf(M, W)] if not keep else None
        for i in range(self.D):
            if i == self.skip:
                y = cs(SK, Pw)
            else:
                y = self.buf(M, W) if keep else pp[i & 1]
            self.fwd(f"pts_linears.{i}", x, k, y, W)
            acts.append((x, k, y))
            if i == self.skip:
                x, k = SK, Pw + W
            else:
                x, k = y, W
        if noviews:
            OUT = self.buf(M, self.output_ch, f32=True)
            self.fwd("out", x, W, OUT, self.output_ch, ACT_NONE, out_f32=True)
            return OUT, ((acts, None, None, SK, E) if keep else None)
        OUT = self.buf(M, 4, f32=True)
        if self.alpha_head:
            self.fwd("alpha", x, W, OUT[:, 3:], 1, ACT_NONE, out_f32=True)
        self.fwd("feature", x, W, cs(V, 0, W), W, ACT_NONE)
        HV = self.buf(M, W // 2)
        self.fwd("views", V, W + self.Vw, HV, W // 2)
        self.fwd("rgb", HV, W // 2, OUT[:, :3], 3, ACT_NONE, out_f32=True)
        saved = (acts, V, HV, SK, E) if keep else None
        return OUT, saved

    def _wgrad_skip(self, n, dZ, SK):
        """weight gradient of the layer behind the skip connection, which reads SK = [embedding (ic of Pw columns) | trunk output]: ONE launch
        over the physical operand into a [W, Pw + W] scratch (the pad columns multiply zeros), whose two column blocks are then added to
        the weight gradient -- instead of one launch per block, each re-reading the [M, W] gradient matrix (3.2 GB at 6.3 M samples)."""
        W, Pw = self.Wd, self.Pw
        tmp = torch.zeros(W, Pw + W, dtype=torch.float32, device=self.dev)
        ops.linear_wgrad(dZ, SK, tmp, W, Pw + W, self.dt, variant=3, deterministic=self.deterministic, x_split_hi=self._in_plain_bwd)
        gw = self.gW(n)
        gw[:, :self.ic] += tmp[:, :self.ic]
        gw[:, self.ic:] += tmp[:, Pw:]

    def backward(self, d_raw, saved):
        """d_raw [M,4] fp32 -> accumulates parameter gradients into the arena (compute="bf16x3_fwd" / "f16f8" / "fp16": _Net._bwd, _scaled_backward)."""
        with self._bwd():
            if self._fp16_backward():
                return self._scaled_backward([d_raw], lambda g: self._backward(g[0], saved))
            return self._backward(d_raw, saved)

    def _backward(self, d_raw, saved):
        acts, V, HV, SK, E = saved[:5]
        W, Pw, g, M = self.Wd, self.Pw, self.g, d_raw.shape[0]
        ah = self.alpha_head
        if self.output_ch > 0:
            oc, x7 = self.output_ch, acts[-1][2]
            self.colsum(d_raw, oc, self.gB("output_linear"))
            dz = self.head_grad(d_raw, oc)
            self.wgrad("output_linear", dz, x7, oc, W)
            dZ = self.buf(M, W)
            self.dgrad("out", dz, roundup(oc, g), dZ, W, mask=x7, colsum=self.gB(f"pts_linears.{self.D - 1}"))
            self._trunk_backward(dZ, acts, SK, E)
            return
        self.colsum(d_raw, 3, self.gB("rgb_linear"))
        if ah:
            self.colsum(d_raw[:, 3:], 1, self.gB("alpha_linear"))
        dz = self.head_grad(d_raw, 3)
        self.wgrad("rgb_linear", dz, HV, 3, W // 2)
        if len(saved) > 5 and self.chain_ok():
            # every data gradient in ONE launch; the weight gradients below read what it stored
            dHV, DB = self.buf(M, W // 2), self.buf(M, W + g)
            dZs = [self.buf(M, W) for _ in range(self.D)]                 # d pts_linears.7 .. .0
            ops.fchain_bwd(ops.CHAIN_CLASSIC, d_raw, self._chain_stream(), saved[5], [dHV, DB] + dZs,
                           [self.gB("views_linears.0"), self.gB("feature_linear")] + [self.gB(f"pts_linears.{i}") for i in range(self.D - 1, -1, -1)])
            x7 = acts[-1][2]
            self.wgrad("views_linears.0", dHV, V, W // 2, W + self.icv)
            self.wgrad("feature_linear", self.cs(DB, 0, W), x7, W, W)
            ops.cast_pad(d_raw[:, 3:], 1, self.cs(DB, W), g, self.dt)
            self.wgrad("alpha_linear", self.cs(DB, W), x7, 1, W)
            for i in range(self.D - 1, -1, -1):
                x, k, y = acts[i]
                n, dZ = f"pts_linears.{i}", dZs[self.D - 1 - i]
                if i == self.skip + 1:
                    self._wgrad_skip(n, dZ, SK)
                elif i == 0:
                    self.wgrad(n, dZ, E, W, self.ic)
                else:
                    self.wgrad(n, dZ, x, W, W)
            return
        dHV = self.buf(M, W // 2)
        self.dgrad("rgb", dz, g, dHV, W // 2, mask=HV, colsum=self.gB("views_linears.0"))
        self.wgrad("views_linears.0", dHV, V, W // 2, W + self.icv)
        DB = self.buf(M, W + (g if ah else 0))                    # [d feature | d alpha (+pad)]
        self.dgrad("views", dHV, W // 2, DB, W, colsum=self.gB("feature_linear"))
        x7 = acts[-1][2]
        self.wgrad("feature_linear", self.cs(DB, 0, W), x7, W, W)
        if ah:
            ops.cast_pad(d_raw[:, 3:], 1, self.cs(DB, W), g, self.dt)
            self.wgrad("alpha_linear", self.cs(DB, W), x7, 1, W)
        dZ = self.buf(M, W)
        self.dgrad("fa", DB, W + (g if ah else 0), dZ, W, mask=x7, colsum=self.gB(f"pts_linears.{self.D - 1}"))
        self._trunk_backward(dZ, acts, SK, E)

    def _trunk_backward(self, dZ, acts, SK, E):
        """weight gradients of pts_linears.{D-1 .. 0} and the data gradients between them, from dZ = d loss / d (pre-activation of the
        last trunk layer)"""
        W, M = self.Wd, dZ.shape[0]
        for i in range(self.D - 1, -1, -1):
            x, k, y = acts[i]
            n = f"pts_linears.{i}"
            if i == self.skip + 1:
                self._wgrad_skip(n, dZ, SK)
            elif i == 0:
                self.wgrad(n, dZ, E, W, self.ic)
            else:
                self.wgrad(n, dZ, x, W, W)
            if i > 0:
                xin = acts[i - 1][2]                              # the trunk activation feeding layer i
                dX = self.buf(M, W)
                self.dgrad(n, dZ, W, dX, W, mask=xin, colsum=self.gB(f"pts_linears.{i - 1}"))
                dZ = dX


# =============================================================================
# mip path (path A)
# =============================================================================
class MipProposalNet(_Net):
    """proposal MLP: n_layers x (Linear+ReLU) width H, density head -> [M,1] fp32 (models.py:299-325)."""

    def __init__(self, arena, prefix, dt, hidden=256, n_layers=4, feature_dim=96, variant=8, bwd_plain=False):
        super().__init__(arena, prefix, dt, variant, bwd_plain)
        assert hidden % self.g == 0
        self.H, self.L, self.fd, self.Ew = hidden, n_layers, feature_dim, roundup(feature_dim, self.g)
        self.fused = True                 # inference through the fused register-resident kernel where it applies (fused_ok)

    @staticmethod
    def param_shapes(hidden=256, n_layers=4, feature_dim=96):
        out = []
        for i in range(n_layers):
            out += [(f"layers.{i}.layers.0.weight", (hidden, feature_dim if i == 0 else hidden)), (f"layers.{i}.layers.0.bias", (hidden,))]
        return out + [("density_layer.weight", (1, hidden)), ("density_layer.bias", (1,))]

    def pack(self, train):
        for i in range(self.L):
            n = f"layers.{i}.layers.0"
            self._pack_fwd(n, n, [(0, 0, self.fd if i == 0 else self.H)], self.Ew if i == 0 else self.H)
            if train and i > 0:
                self._pack_dgrad(n, [n], 0, self.H)
        self._pack_fwd("density", "density_layer", [(0, 0, self.H)], self.H)
        if train:
            self._pack_dgrad("density", ["density_layer"], 0, self.H)
            self._pack_dgrad_cols("enc", [("layers.0.layers.0", 0)], self.fd)

    def fused_ok(self):
        return self.fused and self.dt == ops.BF16 and self.H == 256 and self.L == 4 and self.fd == 96

    def _pack_fused(self):
        L = [(self.W(f"layers.{i}.layers.0"), self.B(f"layers.{i}.layers.0"), [(0, self.fd if i == 0 else self.H, i > 0)])
             for i in range(self.L)]
        L.append((self.W("density_layer"), self.B("density_layer"), [(0, self.H, True)]))
        return fmlp_pack(L, self.dev)

    def _fused_ready(self):
        v = self.version_fn()
        if getattr(self, "_fused_version", None) != v:
            with torch.no_grad():
                self._refresh_fused(self._pack_fused)
            self._fused_version = v

    def chain_ok(self):
        return self.fused_ok() and getattr(self, "fused_chain", True) and not self.deterministic

    def _pack_chain(self):
        L = [(self.W("density_layer").t(), None, [(0, 1, False)])]
        L += [(self.W(f"layers.{i}.layers.0").t(), None, [(0, self.H, True)]) for i in range(self.L - 1, 0, -1)]
        return fmlp_pack(L, self.dev)

    def _chain_stream(self):
        v = self.version_fn()
        if getattr(self, "_chain_version", None) != v:
            with torch.no_grad():
                self._chain = self._refresh_fused(self._pack_chain, "chain")
            self._chain_version = v
        return self._chain[0]

    def forward_fused(self, E):
        self._fused_ready()
        out = self.buf(E.shape[0], 1, f32=True)
        ops.fmlp_proposal_fwd(E, self.fstream, self.fbias, out)
        return out

    def forward_fused_train(self, E):
        """one launch; the four hidden activations are stored for the per-layer backward"""
        self._fused_ready()
        self.ensure_packed(True)
        M, H = E.shape[0], self.H
        ys = [self.buf(M, H) for _ in range(self.L)]
        out = self.buf(M, 1, f32=True)
        words = [torch.empty(ops.mask_bits_words(M, H), dtype=torch.int32, device=self.dev) for _ in range(self.L)]
        ops.fmlp_proposal_train_fwd(E, self.fstream, self.fbias, out, ys, words)
        for y, w in zip(ys, words):
            self._bits[(y.data_ptr(), M)] = (w, H)
        acts, x, k = [], E, self.Ew
        for y in ys:
            acts.append((x, k, y))
            x, k = y, H
        self._chain_bits = (ys[0].data_ptr(), words)        # the fused data-gradient chain reads all four masks
        return out, acts

    def forward(self, E, keep: bool):
        """E [M, Ew] encoded samples (compute dtype) -> raw density [M,1] fp32."""
        if self.fused_ok():
            return self.forward_fused_train(E) if keep else (self.forward_fused(E), None)
        self.ensure_packed(keep)
        M, H = E.shape[0], self.H
        acts, x, k = [], E, self.Ew
        pp = [self.buf(M, H), self.buf(M, H)] if not keep else None
        for i in range(self.L):
            y = self.buf(M, H) if keep else pp[i & 1]
            self.fwd(f"layers.{i}.layers.0", x, k, y, H)
            acts.append((x, k, y))
            x, k = y, H
        out = self.buf(M, 1, f32=True)
        self.fwd("density", x, H, out, 1, ACT_NONE, out_f32=True)
        return out, (acts if keep else None)

    def backward(self, d_raw_density, acts, want_input_grad=False):
        """-> None, or with `want_input_grad` the fp32 gradient [M, Ew] w.r.t. the encoded samples."""
        with self._bwd():
            if self._fp16_backward():
                return self._scaled_backward([d_raw_density], lambda g: self._backward(g[0], acts, want_input_grad))
            return self._backward(d_raw_density, acts, want_input_grad)

    def _backward(self, d_raw_density, acts, want_input_grad):
        H, M = self.H, d_raw_density.shape[0]
        self.colsum(d_raw_density, 1, self.gB("density_layer"))
        dz = self.head_grad(d_raw_density, 1)
        xl = acts[-1][2]
        self.wgrad("density_layer", dz, xl, 1, H)
        cb = getattr(self, "_chain_bits", None)
        if cb is not None and cb[0] == acts[0][2].data_ptr() and self.chain_ok():
            dZs = [self.buf(M, H) for _ in range(self.L)]                 # d layers.3 .. .0 in ONE launch
            ops.fchain_bwd(ops.CHAIN_PROPOSAL, d_raw_density, self._chain_stream(), cb[1], dZs,
                           [self.gB(f"layers.{i}.layers.0") for i in range(self.L - 1, -1, -1)])
            for i in range(self.L - 1, -1, -1):
                self.wgrad(f"layers.{i}.layers.0", dZs[self.L - 1 - i], acts[i][0], H, self.fd if i == 0 else H)
            return self.input_grad("enc", dZs[-1], H, self.Ew) if want_input_grad else None
        dZ = self.buf(M, H)
        self.dgrad("density", dz, self.g, dZ, H, mask=xl, colsum=self.gB(f"layers.{self.L - 1}.layers.0"))
        for i in range(self.L - 1, -1, -1):
            x, k, y = acts[i]
            n = f"layers.{i}.layers.0"
            self.wgrad(n, dZ, x, H, self.fd if i == 0 else H)
            if i > 0:
                dX = self.buf(M, H)
                self.dgrad(n, dZ, H, dX, H, mask=x, colsum=self.gB(f"layers.{i - 1}.layers.0"))
                dZ = dX
        return self.input_grad("enc", dZ, H, self.Ew) if want_input_grad else None


class MipNerfNet(_Net):
    """NeRF MLP of the mip path (models.py:217-296): 8 DenseBlocks width H with cat([x, inputs]) after
    layer `skip_layer` (trunk first), density head from the trunk, bottleneck, cat([bottleneck, cond]),
    `n_cond` cond layers of 128, rgb head.
    Buffers: SKIP [M, H + Ew] = [layer-4 output | IPE (+pad)] (layer 0 reads the IPE columns in place);
             CB [M, H + Cw] = [bottleneck | view encoding (+pad)]."""

    def __init__(self, arena, prefix, dt, hidden=1024, n_layers=8, skip_layer=4, feature_dim=96, cond_dim=27,
                 n_cond=3, cond_units=128, variant=8, semantic_classes=0, bwd_plain=False):
        super().__init__(arena, prefix, dt, variant, bwd_plain)
        self.sc, self.Hs = int(semantic_classes), hidden // 2    # optional semantic head: trunk -> H/2 ReLU -> C (models.py:258-260)
        assert hidden % self.g == 0 and cond_units % self.g == 0 and n_layers > skip_layer + 1
        self.H, self.L, self.skip, self.fd, self.cd, self.nc, self.cu = hidden, n_layers, skip_layer, feature_dim, cond_dim, n_cond, cond_units
        self.Ew, self.Cw = roundup(feature_dim, self.g), roundup(cond_dim, self.g)

    @staticmethod
    def param_shapes(hidden=1024, n_layers=8, skip_layer=4, feature_dim=96, cond_dim=27, n_cond=3, cond_units=128, semantic_classes=0):
        out = []
        for i in range(n_layers):
            k = feature_dim if i == 0 else (hidden + feature_dim if ((i - 1) % skip_layer == 0 and i - 1 > 0) else hidden)
            out += [(f"layers.{i}.layers.0.weight", (hidden, k)), (f"layers.{i}.layers.0.bias", (hidden,))]
        out += [("density_layer.weight", (1, hidden)), ("density_layer.bias", (1,)),
                ("bottleneck_layer.layers.0.weight", (hidden, hidden)), ("bottleneck_layer.layers.0.bias", (hidden,))]
        for j in range(n_cond):
            out += [(f"cond_layers.{j}.layers.0.weight", (cond_units, hidden + cond_dim if j == 0 else cond_units)),
                    (f"cond_layers.{j}.layers.0.bias", (cond_units,))]
        out += [("rgb_layer.weight", (3, cond_units)), ("rgb_layer.bias", (3,))]
        if semantic_classes > 0:
            out += [("semantic_layer.0.layers.0.weight", (hidden // 2, hidden)), ("semantic_layer.0.layers.0.bias", (hidden // 2,)),
                    ("semantic_layer.1.weight", (semantic_classes, hidden // 2)), ("semantic_layer.1.bias", (semantic_classes,))]
        return out

    def _is_skip_in(self, i):  # layer i consumes the concatenated [trunk | enc] buffer
        return i >= 1 and (i - 1) % self.skip == 0 and (i - 1) > 0

    def _is_skip_out(self, i):  # layer i's output is followed by the concat
        return i % self.skip == 0 and i > 0

    def pack(self, train):
        H = self.H
        for i in range(self.L):
            n = f"layers.{i}.layers.0"
            if i == 0:
                self._pack_fwd(n, n, [(0, 0, self.fd)], self.Ew)
            elif self._is_skip_in(i):
                self._pack_fwd(n, n, [(0, 0, H + self.fd)], H + self.Ew)
            else:
                self._pack_fwd(n, n, [(0, 0, H)], H)
            if train and i > 0:
                self._pack_dgrad(n, [n], 0, H)
        self._pack_fwd("density", "density_layer", [(0, 0, H)], H)
        self._pack_fwd("bottleneck", "bottleneck_layer.layers.0", [(0, 0, H)], H)
        for j in range(self.nc):
            n = f"cond_layers.{j}.layers.0"
            self._pack_fwd(n, n, [(0, 0, H + self.cd if j == 0 else self.cu)], H + self.Cw if j == 0 else self.cu)
            if train:
                self._pack_dgrad(n, [n], 0, H if j == 0 else self.cu)
        self._pack_fwd("rgb", "rgb_layer", [(0, 0, self.cu)], self.cu)
        if self.sc:
            self._pack_fwd("sem0", "semantic_layer.0.layers.0", [(0, 0, H)], H)
            self._pack_fwd("sem1", "semantic_layer.1", [(0, 0, self.Hs)], self.Hs)
        if train:
            self._pack_dgrad("rgb", ["rgb_layer"], 0, self.cu)
            # the trunk output feeds the bottleneck, the density head and (optionally) the semantic head: one K-concatenated GEMM
            self._pack_dgrad("bd", ["bottleneck_layer.layers.0", "density_layer"] + (["semantic_layer.0.layers.0"] if self.sc else []), 0, H)
            if self.sc:
                self._pack_dgrad("sem1", ["semantic_layer.1"], 0, self.Hs)
            # gradient w.r.t. the input encodings (pose refinement): IPE columns of layer 0 and of every skip layer, one K-concatenated
            # GEMM over [dZ_0 | dZ_skip]; view encoding columns of the first cond layer
            self.enc_layers = [0] + [i for i in range(self.L) if self._is_skip_in(i)]
            self._pack_dgrad_cols("enc", [(f"layers.{i}.layers.0", 0 if i == 0 else H) for i in self.enc_layers], self.fd)
            self._pack_dgrad_cols("cenc", [("cond_layers.0.layers.0", H)], self.cd)

    # ---- fused colour head (csrc/fmlp.hip: fcolour_fwd_kernel / fcolour_bwd_kernel) ------------------------------------------------
    def colour_fused_ok(self):
        """cat([bottleneck, view encoding]) -> 3 x 128 -> rgb in ONE launch each way: the shipped configuration (hidden 1024, 27
        view-encoding columns, rgb_layer 3 x 128, bf16)"""
        return (getattr(self, "fused_colour", True) and self.dt == ops.BF16 and self.H == 1024 and self.cd == 27 and self.nc == 3 and self.cu == 128
                and self.Cw >= 32)

    def _pack_colour_fwd(self):
        H, cu = self.H, self.cu
        L = [(self.W("cond_layers.0.layers.0"), self.B("cond_layers.0.layers.0"), [(0, H, False), (H, self.cd, False)], True),
             (self.W("cond_layers.1.layers.0"), self.B("cond_layers.1.layers.0"), [(0, cu, True)]),
             (self.W("cond_layers.2.layers.0"), self.B("cond_layers.2.layers.0"), [(0, cu, True)]),
             (self.W("rgb_layer"), self.B("rgb_layer"), [(0, cu, True)])]
        return fmlp_pack(L, self.dev)

    def _pack_colour_bwd(self):
        """transposed weights in the order the data-gradient chain consumes them (no biases)"""
        H, cu = self.H, self.cu
        L = [(self.W("rgb_layer").t(), None, [(0, 3, False)]),                    # d raw_rgb (3 values from memory) -> dC2
             (self.W("cond_layers.2.layers.0").t(), None, [(0, cu, True)]),       # dC2 -> dC1
             (self.W("cond_layers.1.layers.0").t(), None, [(0, cu, True)]),       # dC1 -> dC0
             (self.W("cond_layers.0.layers.0")[:, :H].t(), None, [(0, cu, True)])]  # dC0 -> d bottleneck (the view-encoding columns: input_grad)
        return fmlp_pack(L, self.dev)

    def _colour_streams(self, train):
        v = self.version_fn()
        if getattr(self, "_colour_version", None) != (v, train) and not (not train and getattr(self, "_colour_version", None) == (v, True)):
            with torch.no_grad():
                self._cfwd = self._refresh_fused(self._pack_colour_fwd, "colour_fwd")
                if train:
                    self._cbwd = self._refresh_fused(self._pack_colour_bwd, "colour_bwd")
            self._colour_version = (v, train)

    def alloc_inputs(self, M):
        """-> (SKIP, CB); the encoders write SKIP[:, H:] and CB[:, H:] in place."""
        return self.buf(M, self.H + self.Ew), self.buf(M, self.H + self.Cw)

    def forward(self, SKIP, CB, keep: bool):
        """-> raw_rgb [M,3] fp32, raw_density [M,1] fp32."""
        self.ensure_packed(keep)
        M, H = SKIP.shape[0], self.H
        acts = []
        x, k = self.cs(SKIP, H), self.Ew
        pp = [self.buf(M, H), self.buf(M, H)] if not keep else None
        for i in range(self.L):
            if self._is_skip_out(i):
                y = self.cs(SKIP, 0, H)
            else:
                y = self.buf(M, H) if keep else pp[i & 1]
            self.fwd(f"layers.{i}.layers.0", x, k, y, H)
            acts.append((x, k, y))
            if self._is_skip_out(i):
                x, k = SKIP, H + self.Ew
            else:
                x, k = y, H
        assert k == H, "a skip concat directly before the heads is not supported"
        raw_d = self.buf(M, 1, f32=True)
        self.fwd("density", x, H, raw_d, 1, ACT_NONE, out_f32=True)
        self.fwd("bottleneck", x, H, self.cs(CB, 0, H), H)
        cacts = []
        raw_rgb = self.buf(M, 3, f32=True)
        bbits = self._bits.get((CB.data_ptr(), M)) if keep else None
        if self.colour_fused_ok() and (not keep or (bbits is not None and bbits[1] == H)):
            # ONE launch for cond_layers.0..2 + rgb_layer; training also stores the three hidden activations + their ReLU bit masks
            self._colour_streams(keep)
            cys = cbits = None
            if keep:
                cys = [self.buf(M, self.cu) for _ in range(self.nc)]
                cbits = [torch.empty(ops.mask_bits_words(M, self.cu), dtype=torch.int32, device=self.dev) for _ in range(self.nc)]
            ops.fcolour_fwd(CB, self._cfwd[0], self._cfwd[1], raw_rgb, cys, cbits)
            if keep:
                cx, ck = CB, H + self.Cw
                for cy in cys:
                    cacts.append((cx, ck, cy))
                    cx, ck = cy, self.cu
                cacts.append(("fused", cbits, bbits[0]))
        else:
            cx, ck = CB, H + self.Cw
            for j in range(self.nc):
                cy = self.buf(M, self.cu)
                self.fwd(f"cond_layers.{j}.layers.0", cx, ck, cy, self.cu)
                cacts.append((cx, ck, cy))
                cx, ck = cy, self.cu
            self.fwd("rgb", cx, self.cu, raw_rgb, 3, ACT_NONE, out_f32=True)
        self.raw_sem, S0 = None, None
        if self.sc:
            S0 = self.buf(M, self.Hs)
            self.fwd("sem0", x, H, S0, self.Hs)
            self.raw_sem = self.buf(M, self.sc, f32=True)
            self.fwd("sem1", S0, self.Hs, self.raw_sem, self.sc, ACT_NONE, out_f32=True)
        return raw_rgb, raw_d, ((acts, cacts, SKIP, CB, S0) if keep else None)

    def backward(self, d_raw_rgb, d_raw_density, saved, d_raw_sem=None, want_input_grad=False, want_cond_grad=False, on_done=None):
        """see _backward; under `bwd_plain` (compute="bf16x3_fwd" / "f16f8") the whole pass runs as plain bf16 / scaled fp16 launches (_Net._bwd)"""
        with self._bwd():
            if self._fp16_backward():
                return self._scaled_backward([d_raw_rgb, d_raw_density, d_raw_sem],
                                             lambda g: self._backward(g[0], g[1], saved, g[2], want_input_grad, want_cond_grad, None), on_done)
            return self._backward(d_raw_rgb, d_raw_density, saved, d_raw_sem, want_input_grad, want_cond_grad, on_done)

    def _backward(self, d_raw_rgb, d_raw_density, saved, d_raw_sem=None, want_input_grad=False, want_cond_grad=False, on_done=None):
        """-> None, or with `want_input_grad` (dE fp32 [M, Ew], dV fp32 [M, Cw]): the gradients w.r.t. the IPE and view encodings;
        with `want_cond_grad` alone: dV (the appearance embedding's columns of the condition block need it in every training step).
        `on_done(names)`: called with a list of parameter-name prefixes (relative to this network) as soon as their gradients are final --
        the heads first, then every trunk layer right after its weight gradient -- so that a data-parallel trainer can put each block
        on the wire while the rest of the backward pass still runs (the trunk is back-propagated last layer first)."""
        acts, cacts, SKIP, CB, S0 = saved
        H, g, cu, M = self.H, self.g, self.cu, d_raw_rgb.shape[0]
        dV = None
        DZE = self.buf(M, H * len(self.enc_layers)) if want_input_grad else None
        self.colsum(d_raw_rgb, 3, self.gB("rgb_layer"))
        self.colsum(d_raw_density, 1, self.gB("density_layer"))
        dz = self.head_grad(d_raw_rgb, 3)
        fused = cacts[-1] if (len(cacts) == self.nc + 1 and cacts[-1][0] == "fused") else None
        if fused is not None:
            cacts = cacts[:-1]
        clast = cacts[-1][2]
        self.wgrad("rgb_layer", dz, clast, 3, cu)
        DB = self.buf(M, H + g + (self.Hs if self.sc else 0))       # [d bottleneck | d raw density (+pad) | d semantic hidden]
        if fused is not None:
            # the whole data-gradient chain d raw_rgb -> dC2 -> dC1 -> dC0 -> d bottleneck, masks and the four bias gradients: one launch
            _, cbits, bbits = fused
            dCs = [self.buf(M, cu) for _ in range(self.nc)]                       # [dC2, dC1, dC0]
            ops.fcolour_bwd(d_raw_rgb, self._cbwd[0], [cbits[2], cbits[1], cbits[0], bbits], dCs, self.cs(DB, 0, H),
                            [self.gB("cond_layers.2.layers.0"), self.gB("cond_layers.1.layers.0"), self.gB("cond_layers.0.layers.0"),
                             self.gB("bottleneck_layer.layers.0")])
            for j in range(self.nc - 1, -1, -1):
                cx, ck, cy = cacts[j]
                self.wgrad(f"cond_layers.{j}.layers.0", dCs[self.nc - 1 - j], cx, cu, H + self.cd if j == 0 else cu)
            if want_input_grad or want_cond_grad:
                dV = self.input_grad("cenc", dCs[-1], cu, self.Cw)
        else:
            dC = self.buf(M, cu)
            self.dgrad("rgb", dz, g, dC, cu, mask=clast, colsum=self.gB(f"cond_layers.{self.nc - 1}.layers.0"))
            for j in range(self.nc - 1, -1, -1):
                cx, ck, cy = cacts[j]
                n = f"cond_layers.{j}.layers.0"
                self.wgrad(n, dC, cx, cu, H + self.cd if j == 0 else cu)
                if j > 0:
                    dX = self.buf(M, cu)
                    self.dgrad(n, dC, cu, dX, cu, mask=cx, colsum=self.gB(f"cond_layers.{j - 1}.layers.0"))
                    dC = dX
                else:
                    self.dgrad(n, dC, cu, DB, H, mask=CB, colsum=self.gB("bottleneck_layer.layers.0"))
                    if want_input_grad or want_cond_grad:
                        dV = self.input_grad("cenc", dC, cu, self.Cw)
        ops.cast_pad(d_raw_density, 1, self.cs(DB, H, H + g), g, self.dt)
        xl = acts[-1][2]
        kb = H + g
        if self.sc:
            dS0 = self.cs(DB, H + g)
            if d_raw_sem is None:
                dS0.zero_()
            else:
                self.colsum(d_raw_sem, self.sc, self.gB("semantic_layer.1"))
                dzs = self.head_grad(d_raw_sem, self.sc)
                self.wgrad("semantic_layer.1", dzs, S0, self.sc, self.Hs)
                self.dgrad("sem1", dzs, roundup(self.sc, g), dS0, self.Hs, mask=S0, colsum=self.gB("semantic_layer.0.layers.0"))
                self.wgrad("semantic_layer.0.layers.0", dS0, xl, self.Hs, H)
            kb += self.Hs
        self.wgrad("bottleneck_layer.layers.0", self.cs(DB, 0, H), xl, H, H)
        self.wgrad("density_layer", self.cs(DB, H, H + g), xl, 1, H)
        dZ = self.buf(M, H)
        self.dgrad("bd", DB, kb, dZ, H, mask=xl, colsum=self.gB(f"layers.{self.L - 1}.layers.0"))
        if on_done is not None:      # everything behind the trunk in the arena: density head, bottleneck, colour head (, semantic head)
            on_done(["density_layer", "bottleneck_layer", "cond_layers", "rgb_layer"] + (["semantic_layer"] if self.sc else []))
        for i in range(self.L - 1, -1, -1):
            x, k, y = acts[i]
            n = f"layers.{i}.layers.0"
            self.wgrad(n, dZ, x, H, self.fd if i == 0 else (H + self.fd if self._is_skip_in(i) else H))
            if on_done is not None:  # weight: just now; bias: the column sums of the data gradient that produced dZ
                on_done([f"layers.{i}."])
            if i > 0:
                xin = acts[i - 1][2]
                if want_input_grad and (i - 1) in self.enc_layers:      # lands in its column block of the K-concatenated operand
                    k = self.enc_layers.index(i - 1)
                    dX = self.cs(DZE, k * H, (k + 1) * H)
                else:
                    dX = self.buf(M, H)
                self.dgrad(n, dZ, H, dX, H, mask=xin, colsum=self.gB(f"layers.{i - 1}.layers.0"))
                dZ = dX
        if want_input_grad:
            return self.input_grad("enc", DZE, H * len(self.enc_layers), self.Ew), dV
        return dV if want_cond_grad else None


# =============================================================================
# zipnerf path (path C): small MLPs behind the fused multisample hash-grid featurisation
# =============================================================================
class ZipPropNet(_Net):
    """PropMLP on the waymo.gin branch (internal/models.py:425-427, 481-519 with disable_rgb): features [P, L*C (+pad)]
    -> Linear 64 + ReLU -> Linear 1 = raw density [P,1] fp32."""

    def __init__(self, arena, prefix, dt, feat_dim, hidden=64, variant=8):
        super().__init__(arena, prefix, dt, variant)
        assert self.km == 1, "split-bf16 (compute='bf16x3') is built for the mip path's networks"
        assert hidden % self.g == 0
        # fused route (csrc/zip.hip, snerf_zip_prop_mlp_fwd / _bwd): the whole network in one launch each way on a COMPACT feature
        # buffer (8 or 16 columns instead of the GEMM route's 64): applies to hidden <= 64, feat_dim <= 16; `fused = False` before
        # the first forward selects the per-layer GEMM route (A/B runs, tests)
        import os
        self.fused = hidden <= 64 and feat_dim <= 16 and os.environ.get("SNERF_ZIP_PROP_GEMM", "") == ""     # (the variable: A/B runs of tools/bench_zip.py)
        self.fd, self.H = feat_dim, hidden
        self._Fw_gemm = roundup(feat_dim, self.g)

    @property
    def Fw(self):
        """columns of the feature buffer the caller allocates"""
        return roundup(self.fd, 8) if self.fused else self._Fw_gemm

    def _P(self, name):
        return self.a.p[self.pre + name]

    def _G(self, name):
        return self.a.g[self.pre + name]

    @staticmethod
    def param_shapes(feat_dim, hidden=64):
        return [("density_layer.0.weight", (hidden, feat_dim)), ("density_layer.0.bias", (hidden,)),
                ("density_layer.2.weight", (1, hidden)), ("density_layer.2.bias", (1,))]

    def pack(self, train):
        self._pack_fwd("d0", "density_layer.0", [(0, 0, self.fd)], self.Fw)
        self._pack_fwd("d2", "density_layer.2", [(0, 0, self.H)], self.H)
        if train:
            self._pack_dgrad("d2", ["density_layer.2"], 0, self.H)
            self._pack_dgrad("d0", ["density_layer.0"], 0, self.fd)

    def forward(self, Fb, keep):
        if self.fused:
            raw = ops.zip_prop_mlp_fwd(Fb, self.fd, self._P("density_layer.0.weight"), self._P("density_layer.0.bias"),
                                       self._P("density_layer.2.weight"), self._P("density_layer.2.bias"), self.dt)
            return raw, ((Fb,) if keep else None)
        self.ensure_packed(keep)
        M = Fb.shape[0]
        H1 = self.buf(M, self.H)
        self.fwd("d0", Fb, self.Fw, H1, self.H)
        raw = self.buf(M, 1, f32=True)
        self.fwd("d2", H1, self.H, raw, 1, ACT_NONE, out_f32=True)
        return raw, ((Fb, H1) if keep else None)

    def backward(self, d_raw, saved):
        """-> dF [P, Fw] (gradient w.r.t. the grid features, compute dtype)"""
        if len(saved) == 1:                   # fused route: the hidden activations are recomputed from the features
            return ops.zip_prop_mlp_bwd(saved[0], d_raw.reshape(-1), self.fd, self._P("density_layer.0.weight"), self._P("density_layer.0.bias"),
                                        self._P("density_layer.2.weight"), self._P("density_layer.2.bias"), self.dt,
                                        self._G("density_layer.0.weight"), self._G("density_layer.0.bias"), self._G("density_layer.2.weight"),
                                        self._G("density_layer.2.bias"))
        Fb, H1 = saved
        M = d_raw.shape[0]
        self.colsum(d_raw, 1, self.gB("density_layer.2"))
        dz = self.head_grad(d_raw, 1)
        self.wgrad("density_layer.2", dz, H1, 1, self.H)
        dH1 = self.buf(M, self.H)
        self.dgrad("d2", dz, dz.shape[1], dH1, self.H, mask=H1, colsum=self.gB("density_layer.0"))
        self.wgrad("density_layer.0", dH1, Fb, self.H, self.fd)
        dF = self.buf(M, self.Fw)
        self.dgrad("d0", dH1, self.H, dF, self.Fw)
        return dF


class ZipNerfNet(_Net):
    """NerfMLP on the waymo.gin branch (internal/models.py:425-427, 462-479, 481-519, 586-703; deg_view = 1):
    features 40 -> 64 ReLU -> 256 (x: channel 0 = raw density, all 256 = bottleneck); [x | dir_enc 9] -> 256 ReLU,
    cat([., x, dir_enc]) (skip_layer_dir = 0) -> 256 ReLU -> rgb 3.
    Buffer SB [P, 256 + 256 + Dw] = [lin0 output | x | dir_enc (+pad)]: lin0 reads columns 256.. in place, lin1 the whole row.
    `glo_dim` > 0 (Model.num_glo_features, models.py:454-459, 620-630): the ray's GLO vector -> lin_glo_0 (ReLU) -> lin_glo_1 -> (scale,
    shift) and the second stage reads x * exp(scale) + shift: SB's x block then holds the MODULATED bottleneck, the unmodulated x
    (raw density, semantic logits, the operand of density_layer.2's weight gradient) lives in its own buffer."""

    def __init__(self, arena, prefix, dt, feat_dim=40, hidden=64, bottleneck=256, width=256, dir_dim=9, variant=8, glo_dim=0, glo_width=128):
        super().__init__(arena, prefix, dt, variant)
        assert self.km == 1, "split-bf16 (compute='bf16x3') is built for the mip path's networks"
        assert hidden % self.g == 0 and bottleneck % self.g == 0 and width % self.g == 0
        self.fd, self.H, self.Bw, self.Wd, self.dd = feat_dim, hidden, bottleneck, width, dir_dim
        self.Fw, self.Dw = roundup(feat_dim, self.g), roundup(dir_dim, self.g)
        self.gd, self.Gh, self.Gw = glo_dim, glo_width, roundup(max(glo_dim, 1), self.g)
        assert glo_dim == 0 or (glo_width % self.g == 0 and (2 * bottleneck) % 128 == 0)

    @staticmethod
    def param_shapes(feat_dim=40, hidden=64, bottleneck=256, width=256, dir_dim=9, glo_dim=0, glo_width=128):
        glo = [] if glo_dim <= 0 else [("lin_glo_0.weight", (glo_width, glo_dim)), ("lin_glo_0.bias", (glo_width,)),
                                       ("lin_glo_1.weight", (2 * bottleneck, glo_width)), ("lin_glo_1.bias", (2 * bottleneck,))]
        return [("density_layer.0.weight", (hidden, feat_dim)), ("density_layer.0.bias", (hidden,)),
                ("density_layer.2.weight", (bottleneck, hidden)), ("density_layer.2.bias", (bottleneck,))] + glo + [
                ("lin_second_stage_0.weight", (width, bottleneck + dir_dim)), ("lin_second_stage_0.bias", (width,)),
                ("lin_second_stage_1.weight", (width, width + bottleneck + dir_dim)), ("lin_second_stage_1.bias", (width,)),
                ("rgb_layer.weight", (3, width)), ("rgb_layer.bias", (3,))]

    def pack(self, train):
        B, Wd, dd = self.Bw, self.Wd, self.dd
        self._pack_fwd("d0", "density_layer.0", [(0, 0, self.fd)], self.Fw)
        self._pack_fwd("d2", "density_layer.2", [(0, 0, self.H)], self.H)
        # fp32 density head = row 0 of the second density layer (raw_density = x[..., 0], models.py:511)
        w2, b2 = self.W("density_layer.2"), self.B("density_layer.2")
        hw = self._zeros(128, self.H); hw[0] = w2[0]
        hb = self._zeros(128, f32=True); hb[0] = b2[0]
        self.fw["dhead"], self.fb["dhead"] = hw, hb
        self._pack_fwd("lin0", "lin_second_stage_0", [(0, 0, B + dd)], B + self.Dw)
        self._pack_fwd("lin1", "lin_second_stage_1", [(0, 0, Wd + B + dd)], Wd + B + self.Dw)
        self._pack_fwd("rgb", "rgb_layer", [(0, 0, Wd)], Wd)
        if self.gd:
            self._pack_fwd("glo0", "lin_glo_0", [(0, 0, self.gd)], self.Gw)
            self._pack_fwd("glo1", "lin_glo_1", [(0, 0, self.Gh)], self.Gh)
        if train:
            self._pack_dgrad("rgb", ["rgb_layer"], 0, Wd)
            self._pack_dgrad("lin1a", ["lin_second_stage_1"], 0, Wd)          # columns that multiply lin0's output
            self._pack_dgrad("d2", ["density_layer.2"], 0, self.H)
            self._pack_dgrad("d0", ["density_layer.0"], 0, self.fd)
            # d x = [dZ_lin0 | dZ_lin1 | d raw_density] . [W0[:, x]; W1[:, x]; e_0]
            g = self.g
            W0, W1 = self.W("lin_second_stage_0"), self.W("lin_second_stage_1")
            out = self._zeros(roundup(B, 128), 2 * Wd + g)
            out[:B, :Wd] = W0[:, :B].t()
            out[:B, Wd:2 * Wd] = W1[:, Wd:Wd + B].t()
            # identity rows: column 0 of the trailing block carries d raw_density, columns 1.. the semantic-logit gradients
            # (x[..., 1:1+C] feeds the softmax of the semantic head, models.py:594-597)
            for c in range(min(g, B)):
                out[c, 2 * Wd + c] = 1.0
            self.tw["xcat"] = out
            # gradient w.r.t. the view-direction encoding (pose refinement): both second-stage layers read it, one K-concatenated GEMM
            self._pack_dgrad_cols("denc", [("lin_second_stage_0", B), ("lin_second_stage_1", Wd + B)], dd)
            if self.gd:
                self._pack_dgrad("glo1", ["lin_glo_1"], 0, self.Gh)
                self._pack_dgrad_cols("gloin", [("lin_glo_0", 0)], self.gd)       # d loss / d GLO vector (the embedding rows' gradient)

    # ---- inference as ONE launch (csrc/fmlp.hip, fzip_fwd_kernel): activations in registers, 160 bytes in and 16 bytes out per sample ----
    def fused_infer_ok(self):
        import os
        on = getattr(self, "fused_infer", os.environ.get("SNERF_ZIP_FUSED_INFER", "1") != "0")      # (the variable: A/B runs of tools/bench_zip.py)
        return (on and self.gd == 0 and self.dt in (ops.BF16, ops.F16) and self.fd <= 64 and self.H == 64 and self.Bw == 256 and self.Wd == 256
                and self.dd <= 16)

    def _pack_fused_infer(self):
        """the layers in the order fzip_fwd_kernel consumes them: density_layer.0, density_layer.2, its row 0 once more (fp32 raw density),
        lin_second_stage_0 on [x | dir], then lin_second_stage_1 block by block, each block followed by the two k-steps of rgb_layer
        that consume it"""
        B, Wd, dd, fd, H = self.Bw, self.Wd, self.dd, self.fd, self.H
        W0, b0 = self.W("density_layer.0"), self.B("density_layer.0")
        W2, b2 = self.W("density_layer.2"), self.B("density_layer.2")
        L0, c0 = self.W("lin_second_stage_0"), self.B("lin_second_stage_0")
        L1, c1 = self.W("lin_second_stage_1"), self.B("lin_second_stage_1")
        Wr, br = self.W("rgb_layer"), self.B("rgb_layer")
        W0p = torch.cat([W0, torch.zeros(H, 64 - fd, dtype=W0.dtype, device=W0.device)], 1) if fd < 64 else W0
        layers = [(W0p, b0, [(0, 64, False)]), (W2, b2, [(0, H, True)]), (W2[0:1], b2[0:1], [(0, H, True)]),
                  (L0, c0, [(0, B, True), (B, dd, False)])]
        for j in range(Wd // 32):
            layers.append((L1[32 * j:32 * j + 32], c1[32 * j:32 * j + 32], [(0, Wd, True), (Wd, B, True), (Wd + B, dd, False)]))
            layers.append((Wr[:, 32 * j:32 * j + 32], br if j == 0 else None, [(0, 32, True)]))
        return fmlp_pack(layers, self.dev)

    def _zip_streams(self):
        v = self.version_fn()
        if getattr(self, "_zinfer_version", None) != v:
            with torch.no_grad():
                self._zinfer = self._refresh_fused(self._pack_fused_infer, "zip_infer", dtype=self.tdt)
            self._zinfer_version = v
        return self._zinfer

    def forward_fused(self, Fb, D, want_x=False):
        """Fb [M, 64] grid features, D [M, 16] direction encoding (compute dtype, zero padded) -> raw_rgb [M,3], raw_density [M,1] fp32;
        `want_x`: self.last_x = the first 32 channels of x [M, 32] (the semantic head's logits are its columns 1 .. C)"""
        st, bi = self._zip_streams()
        M = Fb.shape[0]
        raw_rgb, raw_d = self.buf(M, 3, f32=True), self.buf(M, 1, f32=True)
        self.last_x = self.buf(M, 32) if want_x else None
        ops.fmlp_zip_fwd(Fb, D, st, bi, raw_rgb, raw_d, self.last_x)
        return raw_rgb, raw_d

    def forward_fused_train(self, Fb, SB):
        """The training forward as one launch: same `saved` tuple and ReLU bit masks as forward(Fb, SB, keep=True) leaves (the backward does not
        know which of the two ran).  SB [M, Wd + B + Dw] arrives with the direction encoding in its last block; h and x are written into
        its first two."""
        self.ensure_packed(True)                                  # (the backward's operands + the bit-mask registry of this step)
        st, bi = self._zip_streams()
        M, B, Wd = Fb.shape[0], self.Bw, self.Wd
        H1, H3 = self.buf(M, self.H), self.buf(M, Wd)
        h2, X = SB[:, :Wd], SB[:, Wd:Wd + B]
        raw_rgb, raw_d = self.buf(M, 3, f32=True), self.buf(M, 1, f32=True)
        words = [torch.empty(ops.mask_bits_words(M, w), dtype=torch.int32, device=self.dev) for w in (self.H, Wd, Wd)]
        ops.fmlp_zip_train_fwd(Fb, SB[:, Wd + B:], st, bi, raw_rgb, raw_d, [H1, X, h2, H3], words)
        self._bits[(h2.data_ptr(), M)] = (words[1], Wd)          # (what the per-layer data gradients look up by their mask operand)
        self._bits[(H3.data_ptr(), M)] = (words[2], Wd)
        self._zip_bits = (words, H1.data_ptr(), M)                # ... and all three for the fused gradient chain
        self.last_x = X
        return raw_rgb, raw_d, (Fb, H1, SB, H3, None)

    def _pack_fused_chain(self):
        """transposed weights in the order fzip_chain_bwd_kernel consumes them: rgb_layer^T, lin_second_stage_1[:, :Wd]^T, then block by block
        of dx: [lin_second_stage_0[:, :B]^T | lin_second_stage_1[:, Wd:Wd+B]^T | identity on the 32 head-gradient columns] followed by the
        two k-steps of density_layer.2^T (both 32-output blocks of dH1) that consume the block, and density_layer.0^T"""
        B, Wd, fd, H = self.Bw, self.Wd, self.fd, self.H
        W0, W2 = self.W("density_layer.0"), self.W("density_layer.2")
        L0, L1, Wr = self.W("lin_second_stage_0"), self.W("lin_second_stage_1"), self.W("rgb_layer")
        eye = torch.zeros(B, 32, dtype=W0.dtype, device=W0.device)
        eye[:32] = torch.eye(32, dtype=W0.dtype, device=W0.device)                       # (1.0 = "constant one" in an index image as well)
        Wx = torch.cat([L0[:, :B].t(), L1[:, Wd:Wd + B].t(), eye], 1)                     # [B (x channel), Wd (dh) + Wd (dH3) + 32]
        layers = [(Wr.t(), None, [(0, 3, False)]), (L1[:, :Wd].t(), None, [(0, Wd, True)])]
        for j in range(B // 32):
            layers.append((Wx[32 * j:32 * j + 32], None, [(0, Wd, True), (Wd, Wd, True), (2 * Wd, 32, False)]))
            for blk in range(H // 32):
                layers.append((W2[32 * j:32 * j + 32, 32 * blk:32 * blk + 32].t(), None, [(0, 32, True)]))
        W0t = torch.cat([W0.t(), torch.zeros(64 - fd, H, dtype=W0.dtype, device=W0.device)], 0) if fd < 64 else W0.t()
        layers.append((W0t, None, [(0, H, True)]))
        return fmlp_pack(layers, self.dev)

    def fused_chain_ok(self, saved, n_den):
        import os
        zb = getattr(self, "_zip_bits", None)
        on = getattr(self, "fused_chain", os.environ.get("SNERF_ZIP_FUSED_CHAIN", "1") != "0")       # (the variable: A/B runs of tools/bench_zip.py)
        return (on and zb is not None and saved[4] is None and zb[1] == saved[1].data_ptr() and zb[2] == saved[1].shape[0] and not self.deterministic
                and self.fused_infer_ok() and n_den <= 32)

    def alloc(self, M):
        """-> (F, SB): the featurisation kernel writes F[:, :feat_dim] (F arrives zeroed), the view encoder SB[:, Wd+B:]."""
        return torch.zeros(M, self.Fw, dtype=self.tdt, device=self.dev), self.buf(M, self.Wd + self.Bw + self.Dw)

    def forward(self, Fb, SB, keep, glo=None, S=0):
        """`glo` [R, Gw] (compute dtype, zero padded): the rays' GLO vectors, R = rows / S -- required iff glo_dim > 0.  The returned
        `x` [M, B] is the density network's (unmodulated) output: column 0 = raw density, columns 1.. = semantic logits."""
        import os
        if keep and glo is None and self.fused_infer_ok() and self.Dw >= 16 and ops.mask_bits_words(Fb.shape[0], 256) * 4 < (1 << 31) and \
                getattr(self, "fused_train", os.environ.get("SNERF_ZIP_FUSED_TRAIN", "1") != "0"):       # (the variable: A/B runs of tools/bench_zip.py)
            return self.forward_fused_train(Fb, SB)
        self._zip_bits = None                                       # (the fused gradient chain needs the masks the fused forward writes)
        self.ensure_packed(keep)
        M, B, Wd = Fb.shape[0], self.Bw, self.Wd
        H1 = self.buf(M, self.H)
        self.fwd("d0", Fb, self.Fw, H1, self.H)
        glo_saved = None
        if self.gd:
            assert glo is not None and S > 0 and glo.shape[0] * S == M and glo.shape[1] == self.Gw
            X = self.buf(M, B)
            self.fwd("d2", H1, self.H, X, B, ACT_NONE)
            G0 = self.buf(glo.shape[0], self.Gh)
            self.fwd("glo0", glo, self.Gw, G0, self.Gh)
            SS = self.buf(glo.shape[0], 2 * B, f32=True)
            self.fwd("glo1", G0, self.Gh, SS, 2 * B, ACT_NONE, out_f32=True)
            ops.zip_glo_modulate(X, SS, S, SB[:, Wd:Wd + B])
            glo_saved = (X, glo, G0, SS, S)
        else:
            X = SB[:, Wd:Wd + B]
            self.fwd("d2", H1, self.H, X, B, ACT_NONE)
        raw_d = self.buf(M, 1, f32=True)
        self.fwd("dhead", H1, self.H, raw_d, 1, ACT_NONE, out_f32=True)
        self.fwd("lin0", SB[:, Wd:], B + self.Dw, SB[:, :Wd], Wd)
        H3 = self.buf(M, Wd)
        self.fwd("lin1", SB, Wd + B + self.Dw, H3, Wd)
        raw_rgb = self.buf(M, 3, f32=True)
        self.fwd("rgb", H3, Wd, raw_rgb, 3, ACT_NONE, out_f32=True)
        self.last_x = X
        return raw_rgb, raw_d, ((Fb, H1, SB, H3, glo_saved) if keep else None)

    def backward(self, d_raw_rgb, d_raw_density, saved, want_dir_grad=False, want_glo_grad=False):
        """-> dF [P, Fw] (or (dF, dD fp32 [P, Dw]) with `want_dir_grad`: the gradient w.r.t. the direction encoding; with
        `want_glo_grad` additionally d loss / d GLO vectors, fp32 [R, Gw], as the last element).
        d_raw_density [P, 1] or [P, 1 + C]: column 0 = d raw density, columns 1.. = d semantic logits."""
        Fb, H1, SB, H3, glo_saved = saved
        M, B, Wd, g = d_raw_rgb.shape[0], self.Bw, self.Wd, self.g
        self.colsum(d_raw_rgb, 3, self.gB("rgb_layer"))
        dz = self.head_grad(d_raw_rgb, 3)
        self.wgrad("rgb_layer", dz, H3, 3, Wd)
        DZ = self.buf(M, 2 * Wd + g)                                         # [dZ_lin0 | dZ_lin1 | d raw_density (+pad)]
        if self.fused_chain_ok(saved, d_raw_density.shape[1]):
            # the five data gradients as ONE launch (csrc/fmlp.hip, fzip_chain_bwd_kernel) on the bit masks the fused forward wrote; the
            # weight gradients read its outputs exactly like the per-layer chain's
            v = self.version_fn()
            if getattr(self, "_zchain_version", None) != v:
                with torch.no_grad():
                    self._zchain = self._refresh_fused(self._pack_fused_chain, "zip_chain", dtype=self.tdt)
                self._zchain_version = v
            dx, dH1, dF = self.buf(M, B), self.buf(M, self.H), self.buf(M, self.Fw)
            ops.fmlp_zip_chain_bwd(_f32(d_raw_rgb), _f32(d_raw_density), self._zchain[0], self._zip_bits[0], [DZ[:, Wd:2 * Wd], DZ[:, :Wd], dx, dH1, dF],
                                   [self.gB("lin_second_stage_1"), self.gB("lin_second_stage_0"), self.gB("density_layer.2"), self.gB("density_layer.0")])
            self.wgrad("lin_second_stage_1", DZ[:, Wd:2 * Wd], SB, Wd, Wd + B + self.dd)
            self.wgrad("lin_second_stage_0", DZ[:, :Wd], SB[:, Wd:], Wd, B + self.dd)
            self.wgrad("density_layer.2", dx, H1, B, self.H)
            self.wgrad("density_layer.0", dH1, Fb, self.H, self.fd)
            out = (dF, self.input_grad("denc", DZ[:, :2 * Wd], 2 * Wd, self.Dw)) if want_dir_grad else (dF,)
            return out[0] if len(out) == 1 else out
        self.dgrad("rgb", dz, dz.shape[1], DZ[:, Wd:2 * Wd], Wd, mask=H3, colsum=self.gB("lin_second_stage_1"))
        self.wgrad("lin_second_stage_1", DZ[:, Wd:2 * Wd], SB, Wd, Wd + B + self.dd)
        self.dgrad("lin1a", DZ[:, Wd:2 * Wd], Wd, DZ[:, :Wd], Wd, mask=SB[:, :Wd], colsum=self.gB("lin_second_stage_0"))
        self.wgrad("lin_second_stage_0", DZ[:, :Wd], SB[:, Wd:], Wd, B + self.dd)
        assert d_raw_density.shape[1] <= g
        dx = self.buf(M, B)
        d_glo = None
        if glo_saved is None:
            ops.cast_pad(d_raw_density, d_raw_density.shape[1], DZ[:, 2 * Wd:], g, self.dt)
            self.dgrad("xcat", DZ, 2 * Wd + g, dx, B, colsum=self.gB("density_layer.2"))
        else:
            # the second stage read the MODULATED bottleneck: d xm first (the same pack without its identity block), then through the
            # modulation -- d x, d (scale | shift) per ray, and the bias gradient of density_layer.2 from the per-ray column sums
            X, glo, G0, SS, S = glo_saved
            DZ[:, 2 * Wd:].zero_()
            dxm = self.buf(M, B)
            self.dgrad("xcat", DZ, 2 * Wd + g, dxm, B)
            dSS, dxsum = ops.zip_glo_modulate_bwd(dxm, X, SS, _f32(d_raw_density), S, dx)
            ops.colsum_wide_f32(dxsum, B, self.gB("density_layer.2"), deterministic=self.deterministic)
            R = SS.shape[0]
            ops.colsum_wide_f32(dSS, 2 * B, self.gB("lin_glo_1"), deterministic=self.deterministic)
            dS = self.buf(R, 2 * B)
            ops.cast_pad(dSS, 2 * B, dS, 2 * B, self.dt)
            self.wgrad("lin_glo_1", dS, G0, 2 * B, self.Gh)
            dG0 = self.buf(R, self.Gh)
            self.dgrad("glo1", dS, 2 * B, dG0, self.Gh, mask=G0, colsum=self.gB("lin_glo_0"))
            self.wgrad("lin_glo_0", dG0, glo, self.Gh, self.gd)
            if want_glo_grad:
                d_glo = self.input_grad("gloin", dG0, self.Gh, self.Gw)
        X = SB[:, Wd:Wd + B] if glo_saved is None else glo_saved[0]
        self.wgrad("density_layer.2", dx, H1, B, self.H)
        dH1 = self.buf(M, self.H)
        self.dgrad("d2", dx, B, dH1, self.H, mask=H1, colsum=self.gB("density_layer.0"))
        self.wgrad("density_layer.0", dH1, Fb, self.H, self.fd)
        dF = self.buf(M, self.Fw)
        self.dgrad("d0", dH1, self.H, dF, self.Fw)
        out = (dF, self.input_grad("denc", DZ[:, :2 * Wd], 2 * Wd, self.Dw)) if want_dir_grad else (dF,)
        if want_glo_grad:
            out = out + (d_glo,)
        return out[0] if len(out) == 1 else out


def _f32(t):
    return t if t.dtype == torch.float32 else t.float()
