"""Training-mode SparseEncoder on the inference kernels (round 6, VERDICT r5 "next" #3).

The module-by-module path (conv.py / sparse_block.py: a mirror of the reference's, models/backbones/sparse_encoder.py:100-132,
ops/sparse_block.py:88-107, ops/spconv/functional.py:22-60) trains on the gather kernels over int32 neighbour tables that a hash
insert + lookup builds per level, with one host sync per strided convolution and one more per level for the filter gradient's
orderedness probe.  Here the same layers — same parameters, same BatchNorm buffers, same autograd graph shape (one convolution node
and one BatchNorm(+ReLU, +identity) node per layer, so that DDP's bucketed all-reduce still overlaps the backward) — run on what
inference runs on:

  * the rulebook chain of spconv/fused.py (sorted-key index at level 1, rank index below: no hash insert, no table clear), issued
    up front on the geometry stream; it runs underneath the level-1 layers;
  * ONE host read-back per step — the row counts of levels 2..5, fetched on the geometry stream while level 1 computes — where the
    module path makes eight (tensor shapes and BatchNorm's 1/n need exact counts on the host, so a training step cannot be
    read-back free the way inference is);
  * forward of every 3x3x3 SubM layer on the staged-rows (slab) kernels; its input gradient on the SAME kernel and the SAME
    metadata with the mirrored, transposed filter (a symmetric SubM rulebook: offset k of row o reads row i <=> offset K-1-k of
    row i reads row o, so in_grad[i] = sum_k' out_grad[nbr[k', i]] @ W[K-1-k']^T); its filter gradient on the staged-rows kernel
    (csrc/spconv_wgrad_slab.h) over metadata the rulebook producers write directly in that kernel's address format;
  * strided layers on the tiled gather kernels over int32 tables built from the output side (rank-index lookups, no fill), their
    input gradient over the transposed table, their filter gradient on the gather kernel (spconv_wgrad16_kernel).

BatchNorm (+ ReLU, + identity) stays `spconv/bn.py`'s two-launch pair per direction.  16-bit compute only (autocast, or 16-bit
weights); anything else — fp32 training, rows not promised to be in linear order, a module tree that is not conv -> BN1d [-> ReLU] /
SparseBasicBlock — takes the module path."""
import os

import torch
from torch import nn

from .. import _capi
from . import bn as native_bn
from . import fused, ops

_ENABLED = os.environ.get("BEVAMD_SPCONV_FUSED_TRAIN", "1") != "0"
_PAD_ROWS = 256     # feature buffers are allocated in whole 256-row blocks (the largest staged block), handed on as [:m] views


class _Lv:
    """A fused.Level plus its exact row count on the host (None until the step's one read-back)."""

    def __init__(self, level, n=None):
        self.level = level
        self.n = n


class _Plan:
    """One forward pass: the levels, their products, the read-back.

    Issue order (round 6, from the step timeline: level 1 of a training step is bound by the HOST's launch rate — the rulebook
    launches cost ~40 us of Python each — so what the host issues first decides when the GPU starts):
      1. level 1's forward products (sorted-key index, one metadata set) and the step's filter images;
      2. the level-1 layers on the main stream — the GPU computes them while the host
      3. issues the rest of the forward's chain (every downsample, table and metadata set of levels 2..5), when the walker reaches
         the first strided layer;
      4. the read-back (the chain has finished on the GPU by the time the host has issued it), then levels 2..5, GPU-bound;
      5. what only the backward reads (filter-gradient metadata, the stem's level-1 table), behind the whole forward.
    Products are allocated from the main stream's pool and written on the geometry stream without a fork in between
    (fused.Level._fork): the top-of-pass fork orders the geometry stream behind every block freed BEFORE the pass, and the pass
    itself frees nothing — every tensor it makes is saved for the backward.  (Allocating them under the geometry stream instead,
    as fused._Chain does for inference, was measured: that pool sees new sizes every step when the row counts change, and the
    allocator falls through to hipMalloc — encoder forward 3.9 -> 26 ms on augmented inputs.)"""

    def __init__(self, enc, lv1, dtype, mods=()):
        self.enc = enc
        self.lv1 = lv1
        self.dtype = dtype
        self.pending = []      # _Lv whose count is still on the device
        self.gstream = lv1.level.gstream
        self.mods = list(mods)
        self.pos = {id(m): i for i, m in enumerate(self.mods)}
        self.layers = []       # the _Layer of mods[:len(layers)], in execution order
        self.cur = lv1         # level the next layer reads
        self.images = {}       # id(conv) -> [forward image, input-gradient image | None]
        self.late_issued = False

    def _on_geometry_stream(self):
        import contextlib

        return torch.cuda.stream(self.gstream) if self.gstream is not None else contextlib.nullcontext()

    def advance(self, upto):
        """Create the layers mods[len(layers) .. upto] and issue what their forward reads."""
        if len(self.layers) > upto:
            return
        old = fused._PREFETCHING[0]
        fused._PREFETCHING[0] = True
        try:
            while len(self.layers) <= upto and len(self.layers) < len(self.mods):
                m = self.mods[len(self.layers)]
                if m.subm:
                    L = _Layer(m, self, self.cur, self.cur)
                    L.issue()
                else:
                    L = _Layer(m, self, self.cur, None)
                    L.issue()
                    nxt = _Lv(self.cur.level.downsample(m.kernel_size, m.stride, m.padding, wait=False, want_nbr=True)[0])
                    self.pending.append(nxt)
                    L.lv_out = nxt
                    self.cur = nxt
                L.image, L.image_t = self.images.get(id(m), (None, None))
                self.layers.append(L)
        finally:
            fused._PREFETCHING[0] = old

    def layer(self, conv):
        i = self.pos[id(conv)]
        if i >= len(self.layers):
            self.advance(len(self.mods) - 1)      # the walker left level 1: everything else goes out now, in front of the read-back
        return self.layers[i]

    def resolve(self):
        """The step's ONE host sync: the row counts of every level a strided convolution produced, read on the geometry stream."""
        if not self.pending:
            return
        with self._on_geometry_stream():
            host = torch.cat([lv.level.n_dev.reshape(1) for lv in self.pending]).cpu()
        for lv, v in zip(self.pending, host.tolist()):
            lv.n = int(v)
        self.pending = []

    def issue_backward_products(self):
        """What only the backward pass reads (filter-gradient metadata, the level-1 table of the stem's filter gradient)."""
        if self.late_issued:
            return
        self.late_issued = True
        old = fused._PREFETCHING[0]
        fused._PREFETCHING[0] = True
        try:
            for L in self.layers:
                L.issue(forward=False)
        finally:
            fused._PREFETCHING[0] = old


class _Layer:
    """What one convolution of the pass reads: its module, the levels either side, the plan."""

    def __init__(self, conv, plan, lv_in, lv_out):
        # no strong reference back to the plan (plan.layers holds this object): a plan <-> layer cycle would keep a whole step's
        # activations and rulebooks alive until the cyclic collector runs — with the collector paused (bench.py's timed region) a
        # 30-step run ended in allocator retries: forward 3.9 ms for 28 steps, then 18 and 62 ms
        import weakref

        self.conv, self._plan, self.lv_in, self.lv_out = conv, weakref.ref(plan), lv_in, lv_out
        self.dtype = plan.dtype
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.K = conv.kernel_size[0] * conv.kernel_size[1] * conv.kernel_size[2]
        lvl = lv_in.level
        self.variant = fused._slab_variant_for(conv, lvl, self.cin, self.cout) if conv.subm else None
        # staged-rows filter gradient: cin == cout layers as they are; the stem (cin 5 -> 16) with its rows zero-padded to cout
        # channels (round 6: its gather-kernel filter gradient took 172 + 49 us and wanted a hash index + an int32 table of level 1
        # that nothing else reads; padded to 16 -> 16 it is the 23-us kernel of the other level-1 layers, the extra rows of dW dropped)
        self.wg_code = 0
        self.wg_cin = self.cin
        if conv.subm and lvl.linear_order and tuple(conv.kernel_size) == (3, 3, 3) and lvl.allow_slab \
                and (self.cin == self.cout or (self.cin < self.cout and self.cout == 16)):
            lib = _capi.load()
            if lib.bevamd_spconv_wgrad_slab_supported(ops._DT[plan.dtype], self.cout, self.cout) and ops.slab_grid_ok(lvl.shape, 128):
                self.wg_code = int(lib.bevamd_spconv_wgrad_slab_block_rows(self.cout))
                self.wg_cin = self.cout
        elif (not conv.subm) and self.K == 27 and tuple(conv.kernel_size) == (3, 3, 3) and lvl.linear_order and lvl.allow_slab \
                and os.environ.get("BEVAMD_SPCONV_WGRAD_SLAB_STRIDED", "1") != "0":
            # the strided 3x3x3 layers (16 -> 32, 32 -> 64, 64 -> 128): the same kernel over metadata built from the layer's table
            lib = _capi.load()
            if lib.bevamd_spconv_wgrad_slab_supported(ops._DT[plan.dtype], self.cin, self.cout):
                self.wg_code = int(lib.bevamd_spconv_wgrad_slab_block_rows(self.cin))
        self.nbr_t = None
        self.image = self.image_t = None     # forward / input-gradient filter images of this step (one batched launch: run_encoder)

    # ---- products (built on the geometry stream by issue(); these calls find them cached and wait for their events) ----
    def issue(self, forward=True):
        """forward=True: what the forward pass reads (and the level below needs); False: what only the backward pass reads."""
        conv, lvl = self.conv, self.lv_in.level
        if conv.subm:
            if forward:
                if self.variant is not None:
                    lvl.subm_slab(ops.slab_block_rows(self.cin, self.variant), wait=False)
                else:
                    lvl.subm_neighbors(conv.kernel_size, wait=False)
            else:
                if self.wg_code:
                    lvl.subm_slab(self.wg_code, wait=False)
                else:
                    lvl.subm_neighbors(conv.kernel_size, wait=False)
        elif forward:
            lvl.downsample(conv.kernel_size, conv.stride, conv.padding, wait=False, want_nbr=True)
        elif self.wg_code:
            lvl.down_slab_from_table(conv.kernel_size, conv.stride, conv.padding, self.wg_code, wait=False)

    def table(self):
        conv, lvl = self.conv, self.lv_in.level
        if conv.subm:
            return lvl.subm_neighbors(conv.kernel_size)
        return lvl.downsample(conv.kernel_size, conv.stride, conv.padding, want_nbr=True)[1]

    def table_t(self):
        """Input-stationary table of a strided layer (rows = inputs), for its input gradient."""
        if self.nbr_t is None:
            nbr = self.table()
            n_in, m = self.lv_in.n, self.lv_out.n
            t = torch.empty((self.K, max(n_in, 1)), dtype=torch.int32, device=nbr.device)
            with torch.cuda.device(nbr.device):
                rc = _capi.load().bevamd_spconv_transpose_nbr(_capi.ptr(nbr), nbr.stride(0), m, self.K, _capi.ptr(t), t.shape[1],
                                                              _capi.stream_ptr(nbr.device))
            _capi.check(rc, "spconv_transpose_nbr")
            self.nbr_t = t
        return self.nbr_t


def prepare_images(plan, dev, stem_needs_grad=False):
    """Forward and input-gradient filter images of every convolution of the pass (plan.mods, execution order; the first one is the
    stem, whose input gradient nobody asks for unless `stem_needs_grad`) from their master weights, in one launch."""
    specs, slots = [], []
    for i, m in enumerate(plan.mods):
        plan.images[id(m)] = [None, None]
        specs.append((m.weight, False, False))
        slots.append((id(m), 0))
        if i > 0 or stem_needs_grad:
            specs.append((m.weight, True, bool(m.subm)))
            slots.append((id(m), 1))
    for (key, j), img in zip(slots, ops.make_filter_images(specs, plan.dtype, dev)):
        plan.images[key][j] = img
    for L in plan.layers:
        L.image, L.image_t = plan.images[id(L.conv)]


def _rows_buffer(m, c, dtype, dev):
    full = torch.empty(((max(m, 1) + _PAD_ROWS - 1) // _PAD_ROWS * _PAD_ROWS, c), dtype=dtype, device=dev)
    return full[:m]


class _LevelConv(torch.autograd.Function):
    """out = conv(x) of one layer (no bias: the encoder's convolutions have none).  x [n_in, pitch] 16-bit, weight the fp32 (or
    16-bit) master [kx, ky, kz, cin, cout]; returns [n_out, cout] 16-bit."""

    @staticmethod
    def forward(ctx, x, weight, L):
        conv, plan = L.conv, L._plan()
        lvl_in, lvl_out = L.lv_in.level, L.lv_out.level
        image = L.image
        # the launch is bounded by the level's capacity and guarded by its device-side count, like the inference path; the buffer
        # holds the exact rows once they are known (a strided layer learns them right behind its own launch)
        if conv.subm:
            m = L.lv_out.n
            out = _rows_buffer(m, L.cout, plan.dtype, x.device)
            if L.variant is not None:
                meta = lvl_in.subm_slab(ops.slab_block_rows(L.cin, L.variant))
                ops.sparse_conv_slab(x, image, meta, lvl_out.n_cap, L.cin, L.cout, num_out_dev=lvl_out.n_dev, out=out, variant=L.variant)
            else:
                ops.sparse_conv_tiled(x, image, L.table(), lvl_out.n_cap, L.K, L.cin, L.cout, num_out_dev=lvl_out.n_dev, out=out,
                                      variant=fused._variant_for(fused._frames_equivalent(lvl_in), L.K, L.cin, L.cout))
        else:
            nbr = L.table()
            plan.resolve()
            m = L.lv_out.n
            out = _rows_buffer(m, L.cout, plan.dtype, x.device)
            ops.sparse_conv_tiled(x, image, nbr, m, L.K, L.cin, L.cout, out=out,
                                  variant=fused._variant_for(fused._frames_equivalent(lvl_in), L.K, L.cin, L.cout))
        ctx.L = L
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, grad):
        L = ctx.L
        conv, plan = L.conv, L            # (the layer carries the compute dtype; the plan may be gone by now)
        x, = ctx.saved_tensors
        lvl_in = L.lv_in.level
        n_in, m = L.lv_in.n, L.lv_out.n
        g = grad if grad.dtype == plan.dtype else grad.to(plan.dtype)
        if g.stride(1) != 1 or g.stride(0) % 8:
            g = g.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if conv.subm and L.variant is not None and L.cin == L.cout:
                # the mirrored, transposed filter over the same metadata (module docstring)
                meta = lvl_in.subm_slab(ops.slab_block_rows(L.cin, L.variant))
                dx = _rows_buffer(n_in, x.shape[1], plan.dtype, x.device)
                ops.sparse_conv_slab(g, L.image_t, meta, lvl_in.n_cap, L.cout, L.cin, num_out_dev=lvl_in.n_dev, out=dx, variant=L.variant)
            else:
                # the same mirrored image over the layer's own table (SubM), the transposed image over the transposed table (strided)
                dx = _rows_buffer(n_in, x.shape[1], plan.dtype, x.device)
                ops.sparse_conv_tiled(g, L.image_t, L.table() if conv.subm else L.table_t(), n_in, L.K, L.cout, L.cin, out=dx)
            # (columns past cin — the stem's zero padding — are not written: the padding's own backward drops them)
        dw = None
        if ctx.needs_input_grad[1]:
            if L.wg_code and x.stride(0) % 8 == 0:
                meta = lvl_in.subm_slab(L.wg_code) if conv.subm else \
                    lvl_in.down_slab_from_table(conv.kernel_size, conv.stride, conv.padding, L.wg_code)
                xw = x if x.shape[1] >= L.wg_cin else torch.nn.functional.pad(x, (0, L.wg_cin - x.shape[1]))
                dw = ops.sparse_conv_wgrad_slab(xw, g, meta, L.wg_cin, L.cout)
                if L.wg_cin != L.cin:
                    dw = dw[:, :L.cin, :].contiguous()
            else:
                lib = _capi.load()
                xs = x if x.shape[1] == L.cin and x.is_contiguous() else x[:, :L.cin].contiguous()
                gc = g if g.is_contiguous() else g.contiguous()
                nbr = L.table()
                dw = torch.empty((L.K, L.cin, L.cout), dtype=plan.dtype, device=x.device)
                with torch.cuda.device(x.device):
                    wsb = lib.bevamd_spconv_wgrad_workspace_bytes(L.K, L.cin, L.cout)
                    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
                    rc = lib.bevamd_spconv_conv_wgrad(_capi.ptr(xs), _capi.ptr(gc), ops._DT[plan.dtype], _capi.ptr(nbr), nbr.stride(0), m,
                                                      L.K, L.cin, L.cout, _capi.ptr(dw), _capi.ptr(ws), wsb, _capi.stream_ptr(x.device))
                _capi.check(rc, "spconv_conv_wgrad")
            dw = dw.view(conv.weight.shape).to(conv.weight.dtype)
        return dx, dw, None


# ---- module walkers (the shapes fused.py's inference walkers accept) -------------------------------------------------
def _check_conv_bn(conv, bn, dtype):
    """Structure of one layer, checked BEFORE anything runs (a fallback half way through a pass would update the BatchNorm buffers
    of the layers in front of it twice)."""
    from .conv import SparseConvolution

    if not isinstance(conv, SparseConvolution) or conv.conv1x1 or conv.transposed or conv.inverse or conv.ndim != 3 \
            or any(d != 1 for d in conv.dilation) or conv.bias is not None or (conv.subm and any(k % 2 == 0 for k in conv.kernel_size)):
        raise fused.Unfusable("convolution flavour not handled by the fused training path")
    if not ops.tiled_supported(dtype, conv.in_channels, conv.out_channels) or not ops.tiled_supported(dtype, conv.out_channels, conv.in_channels):
        raise fused.Unfusable(f"no tiled kernel for {conv.in_channels}->{conv.out_channels} (or its input gradient)")
    c = conv.out_channels
    ok = (type(bn) is nn.BatchNorm1d and bn.training and bn.momentum is not None and bn.track_running_stats and bn.num_features == c
          and c % 8 == 0 and c <= 256 and 256 % (c // 8) == 0
          and all(p is None or p.dtype == torch.float32 for p in (bn.weight, bn.bias, bn.running_mean, bn.running_var)))
    if not (ok and native_bn._NATIVE):
        raise fused.Unfusable("BatchNorm layer the native kernels do not serve")


def _conv_bn(conv, bn, relu, x, lv, plan, layers, residual=None):
    if layers is None:                   # the validation walk (plan = the compute dtype)
        _check_conv_bn(conv, bn, plan)
        return x, (lv if conv.subm else object())
    L = plan.layer(conv)
    z = _LevelConv.apply(x, conv.weight, L)
    # (num_batches_tracked of every layer was advanced by ONE launch at the top of the pass: run_encoder)
    y, _, _ = native_bn._BnAct.apply(z, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu)
    return y, L.lv_out


def _sequential(seq, x, lv, plan, layers):
    from ..sparse_block import SparseBasicBlock
    from .conv import SparseConvolution
    from .modules import SparseSequential

    mods = list(seq._modules.values())
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, SparseConvolution):
            if not (i + 1 < len(mods) and type(mods[i + 1]) is nn.BatchNorm1d):
                raise fused.Unfusable("convolution without a BatchNorm1d behind it")
            bn = mods[i + 1]
            i += 1
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            i += int(relu)
            x, lv = _conv_bn(m, bn, relu, x, lv, plan, layers)
        elif isinstance(m, SparseBasicBlock):
            if m.downsample is not None:
                raise fused.Unfusable("SparseBasicBlock with a downsample branch")
            y, lv_y = _conv_bn(m.conv1, m.norm1, True, x, lv, plan, layers)
            if lv_y is not lv:
                raise fused.Unfusable("strided conv inside a residual block")
            x, lv = _conv_bn(m.conv2, m.norm2, True, y, lv, plan, layers, residual=x)
        elif isinstance(m, SparseSequential):
            x, lv = _sequential(m, x, lv, plan, layers)
        else:
            raise fused.Unfusable(f"module {type(m).__name__} inside a SparseSequential")
        i += 1
    return x, lv


class _DenseBev(torch.autograd.Function):
    """[B, C*Z, X, Y] dense tensor of the last level's rows (sparse_encoder.py:126-131); backward gathers the rows' gradient."""

    @staticmethod
    def forward(ctx, x, lv):
        lvl = lv.level
        ctx.lv = lv
        ctx.c = x.shape[1]
        return fused.dense_bev(fused.FusedTensor(x, lvl))

    @staticmethod
    def backward(ctx, grad):
        lv = ctx.lv
        lvl = lv.level
        X, Y, Z = lvl.shape
        idx = lvl.indices[:lv.n].long()
        gd = grad.view(lvl.batch, ctx.c, Z, X, Y)
        return gd[idx[:, 0], :, idx[:, 3], idx[:, 1], idx[:, 2]].contiguous(), None


def unsupported_reason(enc, voxel_features, coors_order, num_voxels):
    if not _ENABLED:
        return "BEVAMD_SPCONV_FUSED_TRAIN=0"
    if not enc.training:
        return "module is in eval mode"
    if not voxel_features.is_cuda:
        return "input is not on the GPU"
    if not fused._is_linear(coors_order):
        return "rows are not promised to be in linear order (coors_order='linear')"
    if num_voxels is not None:
        return "capacity-padded inputs (the training path takes exact rows)"
    if _compute_dtype(enc) is None:
        return "fp32 training (the fused training path computes in 16 bits: autocast, or 16-bit weights)"
    return None


def _compute_dtype(enc):
    if torch.is_autocast_enabled('cuda'):
        dt = torch.get_autocast_dtype('cuda')
        return dt if dt in (torch.float16, torch.bfloat16) else None
    dt = enc.conv_input[0].weight.dtype
    return dt if dt in (torch.float16, torch.bfloat16) else None


def run_encoder(enc, voxel_features, coors, batch_size, coors_order="linear"):
    """SparseEncoder.forward in training mode on the fused kernels; raises fused.Unfusable / fused.NotThisCall like the inference
    entry when the module path has to take the call."""
    dtype = _compute_dtype(enc)
    n = voxel_features.shape[0]
    if n < 2:
        raise fused.NotThisCall("fewer than two voxels")
    if enc.__dict__.get("_bevamd_train_tree_checked") != dtype:     # module tree: structure only, nothing runs (plan = the dtype, layers = None)
        lv = object()
        for seq in (enc.conv_input, enc.encoder_layers, enc.conv_out):
            _, lv = _sequential(seq, None, lv, dtype, None)
        enc.__dict__["_bevamd_train_tree_checked"] = dtype
    cin = enc.in_channels
    if voxel_features.shape[1] != cin:
        raise fused.NotThisCall(f"voxel_features has {voxel_features.shape[1]} columns, the encoder reads {cin}")
    dev = voxel_features.device
    coors = coors.int().contiguous()
    pitch = ops.padded_channels(cin)
    with torch.no_grad():
        if voxel_features.dtype == torch.float32 and voxel_features.is_contiguous() and not voxel_features.requires_grad:
            feats = torch.empty((n, pitch), dtype=dtype, device=dev)
            with torch.cuda.device(dev):
                rc = _capi.load().bevamd_spconv_pad_cast_rows(_capi.ptr(voxel_features), n, cin, pitch, ops._dtype_code(feats),
                                                              _capi.ptr(feats), _capi.stream_ptr(dev))
            _capi.check(rc, "spconv_pad_cast_rows")
        else:
            feats = None
    if feats is None:
        feats = torch.nn.functional.pad(voxel_features.to(dtype), (0, pitch - cin))
    g = fused.geometry_stream(dev)
    main = torch.cuda.current_stream(dev)
    pool = fused._status_pool(dev)
    if g is not None:
        g.wait_stream(main)
    lvl1 = fused.Level(coors, n, None, int(batch_size), enc.sparse_shape, gstream=g, linear_order=True, status_pool=pool)
    lvl1.frames_hint = n / float(fused._ROWS_PER_FLAGSHIP_FRAME)
    mods = fused._chain_modules(enc)
    plan = _Plan(enc, _Lv(lvl1, n), dtype, mods)
    layers = True          # (the walkers' "this is not the validation walk" flag)
    try:
        strided = [i for i, m in enumerate(mods) if not m.subm]
        plan.advance((strided[0] if strided else len(mods)) - 1)          # level 1's forward products
        prepare_images(plan, dev, stem_needs_grad=voxel_features.requires_grad)
        # the device-side status words of the chain (fused._first_call_check), read BEFORE the feature pass on the first call of an
        # encoder (BEVAMD_SPCONV_CHECK=1: every call) — the kernels would stay inside their buffers either way, on a wrong rulebook.
        # A checked pass builds everything first.
        if fused._CHECK or not enc.__dict__.get("_bevamd_train_geometry_checked"):
            plan.advance(len(mods) - 1)
            plan.issue_backward_products()
            if g is not None:
                g.synchronize()
            bits = fused.geometry_status(lvl1)
            if bits:
                raise fused.NotThisCall(f"geometry status {bits:#x} (1: a staged range overflowed its 16-bit slots, 2: rows promised as "
                                        "coors_order='linear' are not in ascending linear index)")
            enc.__dict__["_bevamd_train_geometry_checked"] = True
        # ---- the feature pass
        counters = [b for b in (getattr(m, "num_batches_tracked", None) for m in enc.modules() if type(m) is nn.BatchNorm1d) if b is not None]
        if counters:
            torch._foreach_add_(counters, 1)
        x, lv = _sequential(enc.conv_input, feats, plan.lv1, plan, layers)
        x, lv = _sequential(enc.encoder_layers, x, lv, plan, layers)
        x, lv = _sequential(enc.conv_out, x, lv, plan, layers)
        out = _DenseBev.apply(x, lv)
        plan.issue_backward_products()
    finally:
        if g is not None:
            main.wait_stream(g)
    return out
