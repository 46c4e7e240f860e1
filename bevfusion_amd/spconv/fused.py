"""Sync-free inference path of the sparse 3D encoder (eval mode, 16-bit features).

What the module-by-module path (conv.py / sparse_block.py, a mirror of the reference's) costs per frame and this
path removes:
  * one host sync per strided convolution (row counts needed for tensor shapes)   -> counts stay on the device,
    buffers and launches are sized by capacity (`min(n_in * prod(ceil(k/s)), grid volume)`);
  * BatchNorm1d / ReLU / residual add as separate elementwise kernels              -> folded into the convolution
    epilogue (eval-mode BN = fp32 scale/shift per channel);
  * a filter re-layout per convolution call                                        -> filter images cached per weight;
  * hash build per rulebook                                                        -> the (bitmap, prefix) rank index
    that numbers a strided conv's outputs doubles as the lookup structure of the next layers;
  * zero-fill + scatter + permute copy of the dense tail                           -> one gather kernel writes
    [B, C*D, H, W] directly.
Nothing reads back, so the whole encoder can be captured in a HIP graph (`torch.cuda.graph`).  Rulebooks depend only on
voxel coordinates, never on features: the whole geometry chain (hash index, every level's downsample, every neighbour
table) is issued up front on a second HIP stream and runs underneath the convolutions of the earlier levels; each
convolution waits on the event of the table it reads (`BEVAMD_SPCONV_GEOM_STREAM=0` keeps everything on one stream).

Reference semantics followed: SparseEncoder.forward (models/backbones/sparse_encoder.py:100-132), SparseBasicBlock
.forward (ops/sparse_block.py:88-107), SparseConvolution.forward (ops/spconv/conv.py:118-223), BatchNorm1d eval.
"""
import ctypes
import os

import torch
from torch import nn

from .. import _capi
from . import ops

INDEX_HASH, INDEX_RANK = 0, 1


class Unfusable(Exception):
    """The module tree / state does not match what the fused path implements; the caller falls back."""


class NotThisCall(Exception):
    """This particular input is left to the module path (e.g. no voxels at all); the fused path stays enabled."""


# How many convolutions ahead of the feature pass the rulebook chain is issued (see _Chain); -1: the whole chain up front.
_CHAIN_LOOKAHEAD = int(os.environ.get("BEVAMD_SPCONV_CHAIN_LOOKAHEAD", "4"))
# hold each strided layer's products behind the convolutions issued so far (see _Chain.advance)
# (measured, off by default: 4.90 / 4.93 against 4.94 / 4.95 ms per 8-frame step, 0.98 against 0.96 ms on one frame)
_CHAIN_HOLD = os.environ.get("BEVAMD_SPCONV_CHAIN_HOLD", "0") == "1"
# skip a wait on a geometry-stream event the main stream is already ordered behind (tuning switch)
_DEDUPE_WAITS = os.environ.get("BEVAMD_SPCONV_DEDUPE_WAITS", "1") != "0"
_GEOM_STREAMS = {}
_PREFETCHING = [False]   # inside prefetch_geometry: products are built behind the fork point, nothing of this pass runs on the main stream yet

# Per-layer profile (bench.py's `roofline_spconv`): set to a list and every fused convolution appends a record with HIP events
# around its launch and, after a sync, its pair count.  None (default) = no overhead.  Only meaningful in eager passes.
LAYER_PROFILE = None
# launches per layer between the two HIP events of a profiled pass (the launch is idempotent: it reads its inputs and writes `out`).
# One launch behind a device-wide sync starts on an idle, down-clocked GPU: round 5 measured 206-223 us for a 64 -> 64 layer that takes
# 172-176 us back to back (tools/time_slab_variant.py) and 186-192 us inside the replayed graph.  bench.py profiles with 5.
LAYER_PROFILE_REPS = 1


def geometry_stream(device):
    """The side stream rulebook construction runs on (one per device, created on first use); None when disabled."""
    if os.environ.get("BEVAMD_SPCONV_GEOM_STREAM", "1") == "0":
        return None
    key = torch.device(device).index
    if key not in _GEOM_STREAMS:
        _GEOM_STREAMS[key] = torch.cuda.Stream(device=device)
    return _GEOM_STREAMS[key]


class Level:
    """An active voxel set at one resolution: coordinates (capacity-sized), live count on the device, and an index
    that maps a cell to its row (hash for arbitrary row order, rank for ascending order).

    `gstream` (optional): the stream every rulebook kernel of this level is launched on.  Products carry the event
    recorded behind them; `subm_neighbors` / `downsample` make the CURRENT stream wait on it before handing them out.
    Buffers are allocated by the caller's (current-stream) allocator; `run_encoder` joins the two streams before it
    returns, so their reuse stays ordered."""

    def __init__(self, indices, n_cap, n_dev, batch, shape, gstream=None, linear_order=False, status_pool=None, allow_slab=True,
                 sync=None):
        self.indices = indices
        # shared by the levels of one pass: next event number, highest event number the main stream has waited for
        self._sync = sync if sync is not None else {"seq": 0, "awaited": -1}
        self.allow_slab = bool(allow_slab)   # False: every layer of this chain of levels stays on the gather kernels
        self.frames_hint = None     # frames per step as the tilings count them, from the live row count of level 1 (see _frames_hint)
        self.chain = None           # _Chain issuing the rulebook products a few convolutions ahead of the feature pass
        # device status words of the products built for this chain of levels: slices of ONE zeroed tensor (a separate
        # torch.zeros(1) per product put ten 5-us fill kernels in front of the first convolution); [pool, next free]
        self._status_pool = status_pool
        self.n_cap = int(n_cap)
        self.n_dev = n_dev          # int32 device tensor [1] or None (= n_cap rows are all live)
        self.batch = int(batch)
        self.shape = [int(s) for s in shape]
        self.index_kind = None
        self.index = None
        self.index_n_cap = 0
        self.gstream = gstream
        self.ready = None           # event behind the kernels that produced indices / n_dev / index (None: already ordered)
        self._subm = {}
        self._down = {}
        self._slab = {}
        self._down_slab = {}
        self.linear_order = bool(linear_order)   # rows in ascending linear index (every set a strided convolution produced;
        #                                          level 1 when the caller vouches for it: voxelize(..., order="key"))
        self.sorted_index = None    # (keys, x-plane directory): the set's own sorted-key index (ops.sorted_index_build)
        self.sorted_status = None
        self._sorted_ev = None

    @property
    def device(self):
        return self.indices.device

    def _stream_ptr(self):
        if self.gstream is not None:
            return ctypes.c_void_p(self.gstream.cuda_stream)
        return _capi.stream_ptr(self.device)

    def _status(self):
        """A fresh zeroed int32 word for a product's status (None: the op allocates its own)."""
        pool = self._status_pool
        if pool is None or pool[1] >= pool[0].shape[0]:
            return None
        word = pool[0][pool[1]:pool[1] + 1]
        pool[1] += 1
        return word

    def _fork(self):
        """Called before a product is allocated and built OUTSIDE the up-front rulebook chain (a table a convolution asks for
        on demand, e.g. the int32 tables of a profiled pass): the geometry stream must first wait for what the main stream has
        issued so far.  The buffer comes from the main stream's allocator, which hands out a block as soon as its last tensor
        died — possibly while a convolution that read that tensor is still running — and a geometry-stream kernel writing into
        it would not be ordered behind that convolution (seen as NaN rows: sp_nbr_clear's -1 words under a running layer)."""
        if self.gstream is not None and not _PREFETCHING[0]:
            self.gstream.wait_stream(torch.cuda.current_stream(self.device))

    def _mark(self):
        """Event behind everything launched so far on the geometry stream (None on the single-stream path)."""
        if self.gstream is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self.gstream)
        ev._bevamd_seq = self._sync["seq"]     # position among the events of this pass (one geometry stream: totally ordered)
        self._sync["seq"] += 1
        return ev

    def _await(self, ev):
        """Make the current stream wait for `ev` — unless it already waited, in this pass, for this or a LATER event of the geometry
        stream (_DEDUPE_WAITS): the four layers of a level all ask for the same metadata, and every repeated wait is one more
        cross-queue edge on a node of the captured graph."""
        if ev is None:
            return
        seq = getattr(ev, "_bevamd_seq", None)
        if _DEDUPE_WAITS and seq is not None:
            if seq <= self._sync["awaited"]:
                return
            self._sync["awaited"] = seq
        torch.cuda.current_stream().wait_event(ev)

    def ensure_index(self):
        if self.index is not None:
            return
        self._fork()
        lib = _capi.load()
        dev = self.device
        with torch.cuda.device(dev):
            nbytes = lib.bevamd_spconv_hash_index_bytes(self.n_cap)
            self.index = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            rc = lib.bevamd_spconv_hash_index_build(_capi.ptr(self.indices), self.n_cap, _capi.ptr(self.n_dev), self.batch,
                                                    _capi.ints(self.shape), _capi.ptr(self.index), nbytes,
                                                    self._stream_ptr())
        _capi.check(rc, "spconv_hash_index_build")
        self.index_kind, self.index_n_cap = INDEX_HASH, self.n_cap
        self.ready = self._mark()

    def ensure_sorted(self, wait=True):
        """Sorted-key index of a set in linear order (no hash insert, nothing to clear): built once, on the geometry stream."""
        if not self.linear_order:
            raise RuntimeError("sorted-key index of a voxel set that is not in linear order")
        if self.sorted_index is None:
            self._fork()
            self.sorted_index, self.sorted_status = ops.sorted_index_build(self.indices, self.n_cap, self.n_dev, self.batch,
                                                                           self.shape, stream_ptr=self._stream_ptr(),
                                                                           status=self._status())
            self._sorted_ev = self._mark()
        if wait:
            self._await(self._sorted_ev)
        return self.sorted_index

    def use_sorted(self):
        """Neighbour search of this set goes through its sorted-key index (level 1 in key order: it has no rank index);
        BEVAMD_SPCONV_SORTED_ALL=1 extends that to the levels that do have one (tuning)."""
        if not self.linear_order or not _SORTED:
            return False
        return self.index_kind != INDEX_RANK or _SORTED_ALL

    def _neighbors(self, out_indices, m_cap, m_dev, out_shape, ksize, stride, padding, subm):
        lib = _capi.load()
        self.ensure_index()
        self._fork()
        dev = self.device
        K = ksize[0] * ksize[1] * ksize[2]
        nbr = torch.empty((K, max(m_cap, 1)), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.bevamd_spconv_neighbors(_capi.ptr(out_indices), m_cap, _capi.ptr(m_dev), self.batch, _capi.ints(self.shape),
                                             _capi.ints(out_shape), _capi.ints(ksize), _capi.ints(stride), _capi.ints(padding),
                                             int(subm), self.index_kind, _capi.ptr(self.index), self.index_n_cap,
                                             _capi.ptr(nbr), nbr.shape[1], self._stream_ptr())
        _capi.check(rc, "spconv_neighbors")
        return nbr

    def subm_neighbors(self, ksize, wait=True):
        """nbr [K, n_cap] of a submanifold convolution over this set (built once per kernel size)."""
        key = tuple(ksize)
        if key not in self._subm:
            nbr = self._neighbors(self.indices, self.n_cap, self.n_dev, self.shape, list(ksize), [1, 1, 1],
                                  [k // 2 for k in ksize], True)
            self._subm[key] = (nbr, self._mark())
        nbr, ev = self._subm[key]
        if wait:
            self._await(ev)
        return nbr

    def subm_slab(self, block_rows, wait=True):
        """Slab metadata (ops.SlabMeta) of the 3x3x3 SubM neighbour table, for `block_rows`-row blocks: built once, on the
        geometry stream, behind the table it rewrites."""
        if block_rows not in self._slab:
            self._fork()
            if self.use_sorted():
                self.ensure_sorted(wait=False)
                meta = ops.slab_build_from_sorted(self.indices, self.n_cap, self.n_dev, self.batch, self.shape, self.shape,
                                                  [1, 1, 1], [1, 1, 1], True, self.sorted_index, self.n_cap, block_rows,
                                                  stream_ptr=self._stream_ptr(), status=self._status())
            elif (3, 3, 3) in self._subm or not _SLAB_DIRECT:
                nbr = self.subm_neighbors((3, 3, 3), wait=False)
                meta = ops.slab_build(nbr, self.n_cap, self.n_dev, block_rows, stream_ptr=self._stream_ptr(), status=self._status())
            else:   # nobody asked for the int32 table: every row looks its neighbours up itself, 54 B per row written in all
                self.ensure_index()
                meta = ops.slab_build_from_index(self.indices, self.n_cap, self.n_dev, self.batch, self.shape, self.index_kind,
                                                 self.index, self.index_n_cap, block_rows, stream_ptr=self._stream_ptr(),
                                                 status=self._status())
            self._slab[block_rows] = (meta, self._mark())
        meta, ev = self._slab[block_rows]
        if wait:
            self._await(ev)
        return meta

    def down_slab(self, ksize, stride, padding, block_rows, wait=True):
        """Slab metadata of the strided 3x3x3 convolution (ksize, stride, padding) over this set, for `block_rows`-row blocks
        of OUTPUT rows: from the sorted-key index of this (input) set, behind the downsample that numbered the outputs."""
        key = (tuple(ksize), tuple(stride), tuple(padding), block_rows)
        if key not in self._down_slab:
            out, _ = self.downsample(ksize, stride, padding, wait=False, want_nbr=False)
            self.ensure_sorted(wait=False)
            self._fork()
            meta = ops.slab_build_from_sorted(out.indices, out.n_cap, out.n_dev, self.batch, self.shape, out.shape, list(stride),
                                              list(padding), False, self.sorted_index, self.n_cap, block_rows,
                                              stream_ptr=self._stream_ptr(), status=self._status())
            self._down_slab[key] = (meta, self._mark())
        meta, ev = self._down_slab[key]
        if wait:
            self._await(ev)
        return meta

    def down_slab_from_table(self, ksize, stride, padding, block_rows, wait=True):
        """Slab metadata of the strided 3x3x3 convolution from its int32 neighbour table (ops.slab_build) — what the staged-rows
        filter gradient of a training step reads (spconv/fused_train.py; a layer that trains keeps its table anyway)."""
        key = (tuple(ksize), tuple(stride), tuple(padding), block_rows, "table")
        if key not in self._down_slab:
            out, nbr = self.downsample(ksize, stride, padding, wait=False, want_nbr=True)
            self._fork()
            meta = ops.slab_build(nbr, out.n_cap, out.n_dev, block_rows, stream_ptr=self._stream_ptr(), status=self._status())
            self._down_slab[key] = (meta, self._mark())
        meta, ev = self._down_slab[key]
        if wait:
            self._await(ev)
        return meta

    def downsample(self, ksize, stride, padding, wait=True, want_nbr=True):
        """(output Level with its rank index, nbr [K, cap_out]) of a strided convolution over this set.  want_nbr=False (the
        convolution reads slab metadata instead): the int32 neighbour table is not built, nbr is None — a later call that
        does want it (profiling a slab layer) builds it from the output side."""
        key = (tuple(ksize), tuple(stride), tuple(padding))
        if key in self._down:
            out, nbr = self._down[key]
            ev = None
            if nbr is None and want_nbr:
                nbr = self._neighbors(out.indices, out.n_cap, out.n_dev, out.shape, list(ksize), list(stride), list(padding), False)
                self._down[key] = (out, nbr)
                ev = self._mark()
            if wait:
                self._await(out.ready)
                self._await(ev)
            return out, nbr
        self._fork()
        lib = _capi.load()
        dev = self.device
        out_shape = ops.get_conv_output_size(self.shape, list(ksize), list(stride), list(padding), [1, 1, 1])
        if min(out_shape) <= 0:
            raise Unfusable(f"empty output grid {out_shape}")
        bound = 1
        for k, s in zip(ksize, stride):
            bound *= (k + s - 1) // s
        volume = self.batch * out_shape[0] * out_shape[1] * out_shape[2]
        cap = max(1, min(self.n_cap * bound, volume))
        out_indices = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        num_out = torch.empty(1, dtype=torch.int32, device=dev)
        K = ksize[0] * ksize[1] * ksize[2]
        # The table of a strided layer that stays on the gather kernels: from the INPUT side inside bevamd_spconv_downsample (a
        # -1 fill of 27 * cap words, then every input scatters into the <= 8 outputs it feeds), or — when this (input) level owns a
        # rank index — from the OUTPUT side afterwards: every output row looks its 27 input cells up (8-byte loads that
        # neighbouring rows share) and writes its column of the table once, coalesced, no fill (_DOWN_NBR_FROM_OUTPUTS)
        from_outputs = want_nbr and _DOWN_NBR_FROM_OUTPUTS and self.index_kind == INDEX_RANK
        nbr = torch.empty((K, cap), dtype=torch.int32, device=dev) if want_nbr and not from_outputs else None
        # A level in linear order that owns a lookup structure (rank index; level 1 in key order: the directory of its sorted-key
        # index) finds the inputs of a tile of output cells as ONE row range: the output bitmap is built in LDS, tile by tile — two
        # launches instead of four, no 32-byte-per-word byte map (bevamd_spconv_downsample_sorted)
        src_kind = src = None
        if _DOWN_SORTED and self.linear_order and nbr is None and max(ksize) <= 3:
            if self.index_kind == INDEX_RANK:
                src_kind, src = 1, self.index
            elif self.use_sorted():
                self.ensure_sorted(wait=False)
                off = (max(self.n_cap, 1) * 4 + 255) // 256 * 256
                src_kind, src = 0, self.sorted_index[off:]
        with torch.cuda.device(dev):
            nbytes = lib.bevamd_spconv_rank_index_bytes(self.batch, _capi.ints(out_shape))
            index = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            if src is not None:
                rc = lib.bevamd_spconv_downsample_sorted(_capi.ptr(self.indices), self.n_cap, _capi.ptr(self.n_dev), self.batch,
                                                         _capi.ints(self.shape), _capi.ints(out_shape), _capi.ints(ksize),
                                                         _capi.ints(stride), _capi.ints(padding), src_kind, _capi.ptr(src),
                                                         _capi.ptr(out_indices), cap, _capi.ptr(num_out), _capi.ptr(index), nbytes,
                                                         self._stream_ptr())
            else:
                rc = lib.bevamd_spconv_downsample(_capi.ptr(self.indices), self.n_cap, _capi.ptr(self.n_dev), self.batch,
                                                  _capi.ints(self.shape), _capi.ints(out_shape), _capi.ints(ksize),
                                                  _capi.ints(stride), _capi.ints(padding), _capi.ptr(out_indices), cap,
                                                  _capi.ptr(num_out), _capi.ptr(index), nbytes, _capi.ptr(nbr), cap,
                                                  self._stream_ptr())
        _capi.check(rc, "spconv_downsample")
        out = Level(out_indices, cap, num_out, self.batch, out_shape, gstream=self.gstream, status_pool=self._status_pool,
                    allow_slab=self.allow_slab, sync=self._sync)
        out.frames_hint = self.frames_hint
        out.chain = self.chain
        out.index_kind, out.index, out.index_n_cap = INDEX_RANK, index, cap
        out.linear_order = True
        if from_outputs:
            nbr = self._neighbors(out_indices, cap, num_out, out_shape, list(ksize), list(stride), list(padding), False)
        out.ready = self._mark()    # behind the downsample: indices, count, rank index and this conv's nbr are final
        self._down[key] = (out, nbr)
        if wait:
            self._await(out.ready)
        return out, nbr


class FusedTensor:
    def __init__(self, features, level):
        self.features = features   # [level.n_cap, C] 16-bit; rows >= the live count are undefined
        self.level = level


# ---- per-module caches ------------------------------------------------------------------------------------
def _version_key(*tensors):
    return tuple((t.data_ptr(), t._version, t.dtype, t.device) if t is not None else None for t in tensors)


def folded(conv, bn, dtype):
    """(filter image, bias, bn_scale, bn_shift) of `conv` [+ eval-mode BatchNorm1d], cached on the conv module."""
    key = (dtype,) + _version_key(conv.weight, conv.bias, *(() if bn is None else (bn.weight, bn.bias, bn.running_mean,
                                                                                   bn.running_var)))
    cache = conv.__dict__.get("_bevamd_fused")
    if cache is not None and cache[0] == key:
        return cache[1]
    w = conv.weight.detach()
    if w.dtype != dtype:
        w = w.to(dtype)
    image = ops.make_filter_image(w)
    bias = None if conv.bias is None else conv.bias.detach().to(dtype).contiguous()
    scale = shift = None
    if bn is not None:
        if bn.running_mean is None or bn.running_var is None:
            raise Unfusable("BatchNorm without running statistics")
        inv = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
        g = bn.weight.detach().float() if bn.weight is not None else torch.ones_like(inv)
        b = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(inv)
        scale = (g * inv).contiguous()
        shift = (b - bn.running_mean.detach().float() * scale).contiguous()
    val = (image, bias, scale, shift)
    conv.__dict__["_bevamd_fused"] = (key, val)
    return val


# ---- module walkers ---------------------------------------------------------------------------------------
def _conv(conv, x, bn=None, relu=False, residual=None):
    from .conv import SparseConvolution

    assert isinstance(conv, SparseConvolution)
    if conv.conv1x1 or conv.transposed or conv.inverse or any(d != 1 for d in conv.dilation) or conv.ndim != 3:
        raise Unfusable("convolution flavour not handled by the fused path")
    if bn is not None and (bn.training or not isinstance(bn, nn.BatchNorm1d)):
        raise Unfusable("BatchNorm must be BatchNorm1d in eval mode")
    dtype = x.features.dtype
    cin, cout = conv.in_channels, conv.out_channels
    if not ops.tiled_supported(dtype, cin, cout):
        raise Unfusable(f"no tiled kernel for {cin}->{cout} {dtype}")
    if x.features.shape[1] < ops.padded_channels(cin):
        raise Unfusable("feature pitch smaller than the padded channel count")
    image, bias, scale, shift = folded(conv, bn, dtype)
    lvl = x.level
    if lvl.chain is not None:
        lvl.chain.before(conv)
    slab_variant = _slab_variant_for(conv, lvl, cin, cout)
    # a strided 3x3x3 layer that stays on the gather kernels can still read slot metadata (sorted-key search) instead of a table
    slots_meta = slab_variant is None and _gather_reads_slots(conv, lvl)
    want_nbr = (slab_variant is None and not slots_meta) or LAYER_PROFILE is not None
    if conv.subm:
        # the slab kernels read their own metadata: the int32 neighbour table is only built for them when profiling (pair counts)
        nbr = lvl.subm_neighbors(conv.kernel_size) if want_nbr else None
        out_lvl = lvl
    else:
        out_lvl, nbr = lvl.downsample(conv.kernel_size, conv.stride, conv.padding, want_nbr=want_nbr)
    K = conv.kernel_size[0] * conv.kernel_size[1] * conv.kernel_size[2]
    out = torch.empty((out_lvl.n_cap, cout), dtype=dtype, device=x.features.device)
    rec = None
    if LAYER_PROFILE is not None:
        torch.cuda.synchronize()          # the rulebook chain (geometry stream) is out of the way: the events bracket the kernel
        rec = dict(cin=cin, cout=cout, K=K, subm=bool(conv.subm),
                   kernel="slab" if slab_variant is not None else "gather+slots" if slots_meta else "gather",
                   variant=slab_variant if slab_variant is not None else _variant_for(_frames_equivalent(lvl), K, cin, cout),
                   start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True),
                   n_in=lvl.n_dev, n_out=out_lvl.n_dev, n_in_cap=lvl.n_cap, n_out_cap=out_lvl.n_cap, nbr=nbr)
        rec["start"].record()
    for _ in range(LAYER_PROFILE_REPS if rec is not None else 1):
        if slab_variant is not None:
            rows = ops.slab_block_rows(cin, slab_variant)
            meta = lvl.subm_slab(rows) if conv.subm else lvl.down_slab(conv.kernel_size, conv.stride, conv.padding, rows)
            ops.sparse_conv_slab(x.features, image, meta, out_lvl.n_cap, cin, cout, bias=bias, bn_scale=scale, bn_shift=shift,
                                 residual=residual, relu=relu, num_out_dev=out_lvl.n_dev, out=out, variant=slab_variant)
        elif slots_meta:
            meta = lvl.down_slab(conv.kernel_size, conv.stride, conv.padding, _GATHER_SLOT_ROWS)
            ops.sparse_conv_tiled_slots(x.features, image, meta, out_lvl.n_cap, cin, cout, bias=bias, bn_scale=scale, bn_shift=shift,
                                        residual=residual, relu=relu, num_out_dev=out_lvl.n_dev, out=out,
                                        variant=_variant_for(_frames_equivalent(lvl), K, cin, cout))
        else:
            ops.sparse_conv_tiled(x.features, image, nbr, out_lvl.n_cap, K, cin, cout, bias=bias, bn_scale=scale, bn_shift=shift,
                                  residual=residual, relu=relu, num_out_dev=out_lvl.n_dev, out=out,
                                  variant=_variant_for(_frames_equivalent(lvl), K, cin, cout))
    if rec is not None:
        rec["end"].record()
        rec["reps"] = LAYER_PROFILE_REPS
        LAYER_PROFILE.append(rec)
    return FusedTensor(out, out_lvl)


def summarize_layer_profile(records, elem_bytes=2):
    """Per-layer figures of a profiled eager pass: FLOP = 2 * pairs * Cin * Cout, ideal bytes =
    N_in*Cin*s + pairs*8 + K*Cin*Cout*s + N_out*Cout*s, microseconds from the HIP events, fractions of the dense MFMA peak
    (16-bit: 2.5 PFLOP/s) and of 8 TB/s."""
    torch.cuda.synchronize()
    out = []
    for r in records:
        n_in = int(r["n_in"].item()) if r["n_in"] is not None else r["n_in_cap"]
        n_out = int(r["n_out"].item()) if r["n_out"] is not None else r["n_out_cap"]
        pairs = int((r["nbr"][:, :n_out] >= 0).sum().item())
        us = r["start"].elapsed_time(r["end"]) * 1e3 / max(int(r.get("reps", 1)), 1)
        flop = 2.0 * pairs * r["cin"] * r["cout"]
        ideal = n_in * r["cin"] * elem_bytes + pairs * 8 + r["K"] * r["cin"] * r["cout"] * elem_bytes + n_out * r["cout"] * elem_bytes
        out.append(dict(layer=f"{'subm' if r['subm'] else 'conv'} {r['cin']}->{r['cout']} K={r['K']}", kernel=r["kernel"],
                        variant=r["variant"], rows_in=n_in, rows_out=n_out, pairs=pairs, us=us, gflop=flop / 1e9,
                        tflops=flop / (us * 1e-6) / 1e12, frac_mfma_peak=flop / (us * 1e-6) / 2.5e15,
                        # real pairs / issued taps of the dense-tap output-stationary kernels (every row x every kernel offset)
                        useful_mfma_fraction=pairs / float(max(r["K"] * n_out, 1)), ideal_mb=ideal / 1e6,
                        frac_hbm_peak_on_ideal_bytes=ideal / (us * 1e-6) / 8e12))
    return out


# Tiling per layer shape when a step carries several frames.  The library's own choice (variant 0) is tuned on one frame's row
# counts, where the 128-channel layers have fewer row blocks than the GPU has CUs and want the smallest tile; the live row count is
# unknown to the host here (capacity launches), so the frame count stands in for it.  tools/sweep_spconv.py --frames 8:
# 128->128 228 -> 194 us, 64->128 128 -> 102 us, 32->64 153 -> 145 us, 32->32 274 -> 266 us, 64->64 251 -> 246 us
# (A/B on one box, 8 frames: LiDAR branch 6.11-6.18 -> 5.91-5.94 ms; entries for the 5->16 and 16->32 layers, 10-16 % faster
# in isolation, changed nothing inside the graph and were left out).
_BATCHED_VARIANTS = {(32, 32): 2213, (32, 64): 2211, (64, 64): 2221, (64, 128): 2211, (128, 128): 2221}


# Slab (staged-rows) kernels for the 3x3x3 SubM layers of the sorted levels (csrc/spconv_slab*.h): BEVAMD_SPCONV_SLAB=0 keeps
# the gather kernels; BEVAMD_SPCONV_SLAB_VARIANTS="32:322133,64:642232" overrides the per-width variant (tuning).  Codes:
# xxxxxx = filter staged in LDS (spconv_slab.h), 1xxxxxx = filter fragments in registers (spconv_slab_regw.h), 2xxxxxx = the same,
# persistent (spconv_slab_persist.h).  Measured (tools/sweep_spconv.py --slab, profiles/r02_slab_sweep_b{1,8}_v3.txt), 8 frames,
# 32 / 64 / 128 channels: 186 / 183 / 151 us against 290 / 242 / 193 us of the gather kernels (first cut: 211 / 198 / 164); one
# frame: 26.7 / 30.8 us against 41 / 35.8 us, 128 channels 38.5-40 against 38.8 us (188 blocks on 256 CUs) -> those layers switch
# over from 4 frames per step.  The persistent 32-channel kernel was only measured at 8 frames: smaller batches keep its one-block twin.
# Narrow layers (cin padded to 8 | 16: csrc/spconv_slab_small.h) on a level in linear order — level 1 when the voxelizer wrote
# its rows in key order: 3000256 = 256-row blocks (the SubM layers share one metadata set); the strided 16 -> 32 convolution:
# 3100128 = 128-row blocks, BOTH output tiles in one wave (round 5: these kernels are bound by the LDS pipe, and with the two tiles
# on two waves — 3000128, 8-wave workgroups — every operand fragment was read from LDS twice: LiDAR branch alone 3.54 -> 3.45 ms).
_CHECK = os.environ.get("BEVAMD_SPCONV_CHECK", "0") == "1"   # read the geometry status words back after every fused forward (one sync)
_SLAB_NARROW_SUBM = 3000256
_SLAB_NARROW_STRIDED = 3100128
_SORTED = os.environ.get("BEVAMD_SPCONV_SORTED", "1") != "0"          # sorted-key neighbour search on levels without a rank index
_SORTED_ALL = os.environ.get("BEVAMD_SPCONV_SORTED_ALL", "0") == "1"  # ... and on those that have one (tuning)
# 32 channels from 4 frames: the filter-stationary kernel (spconv_slab_fstat.h; 64-row blocks, baked slots): 159 us isolated against 190
# of the persistent register-ring kernel 2324410, 5.58-5.60 against 5.77-5.78 ms per 8-frame step on the same box (its results
# agree with the other kernels' to fp32 rounding, not bit for bit: v_mfma_32x32x16 sums 16 channels per step)
# Round 5: 64 channels on the register-filter kernel with BAKED slot metadata (1644228: the fragment address is one v_xad_u32;
# 189-193 -> 172-174 us per layer at 8 frames, 29.9 -> 28.2 at one; the same flag buys nothing at 128 channels: 154-157 either way);
# 128 channels below 4 frames: 64-row blocks (1642220: one frame's level 4 is 188 blocks of 128 rows on 256 CUs; 39.5 -> 34.0 us,
# gather kernel 38.8) — the kernel reads the filter-stationary kernels' baked 64-row metadata (EXPERIMENTS.md C.3: the "epilogue
# bug" of B.17 was that metadata format read as raw slots).
# Round 6: 32 channels on the filter-stationary WAVE-PAIR kernel (spconv_slab_fstat2.h, 4100128: two waves per SIMD split the input
# channels, same baked metadata as 4000112).  One box, 8 frames, the encoder's epilogues: 139.7 / 149.8 / 154.3 us (BN + ReLU / + residual /
# + device-side row count) against 167.0 / 171.9 / 173.6 of the one-wave kernel; inside the HIP graph 141-145 against 161-170 us per
# layer.  The 8-frame step does not move (4.32-4.36 ms either way, three interleaved pairs of 200 steps: those layers run beside bev_pool
# and the next batch's rulebook chain, and the three together are bound by what they share), the 4-frame step does: 2.36-2.37 against
# 2.40-2.41 ms.  Two waves per SIMD leave room for a neighbour: with a rulebook kernel of the next levels beside it (the encoder's own
# geometry stream under --overlap none) a layer takes 250 us where the one-wave kernel, whose 469-register waves fill the SIMDs, took
# 170 and let the rulebook kernel wait instead — the LiDAR branch alone is 30 us slower for it (3.46 against 3.43 ms), every shared
# schedule (ahead, lidar) is not (EXPERIMENTS D.12).
_SLAB_DEFAULT = {32: 4100128, 64: 1644228, 128: 1644220}
_SLAB_DEFAULT_SMALL_BATCH = {32: 1322410, 128: 1642220}   # below _SLAB_SMALL_BATCH_BELOW frames per step
# 32 channels: the wave-pair kernel from TWO frames on (one box, encoder epilogues + device-side row count, 1 / 2 / 3 frames:
# 1322410 29.6 / 53.2 / 74.2 us, 4000112 30.8 / 50.3 / 67.2, 4100128 29.2 / 43.5 / 58.8); 128 channels: 64-row blocks below 2.5 frames
# (same box, 1 / 2 / 3 / 4 frames: 1642220 34.4 / 54.7 / 85.5 / 102.2 us, 1644220 40.4 / 56.6 / 79.9 / 93.4; 3.5 until round 6)
_SLAB_SMALL_BATCH_BELOW = {32: 1.5, 128: 2.5}
_SLAB_MIN_BATCH = {}
# The same decisions in LIVE ROWS (8 sparse frames are not 8 capped ones).  The tilings were measured on
# capped flagship frames (160 k voxels each), so "frames" = live level-1 rows / 160 k.  The host does not know the row count on the
# sync-free path: the encoder's FIRST eager call at a batch size reads it back once (_frames_hint), every later call at that batch
# size — graph captures included — uses the same figure, so that a replayed graph and its eager warm-up run the same kernels.
_ROWS_PER_FLAGSHIP_FRAME = 160000


def _frames_hint(enc, batch_size, n_rows, num_voxels):
    """Frames-equivalent of this encoder's inputs at `batch_size`, measured once (one 4-byte read-back outside graph capture)."""
    hints = enc.__dict__.setdefault("_bevamd_frames_hint", {})
    if batch_size not in hints:
        if num_voxels is None:
            hints[batch_size] = n_rows / float(_ROWS_PER_FLAGSHIP_FRAME)
        elif torch.cuda.is_current_stream_capturing():
            return None
        else:
            live = int(num_voxels.reshape(-1)[0])
            if live < 0:   # the voxelizer's single-pass kernels flag a stalled look-back word as total = -1 (csrc/voxelize.hip): flagged, never silent
                raise RuntimeError("SparseEncoder: the voxelizer reported a failed pass (num_voxels < 0: a single-pass look-back word "
                                   "stalled); rerun with BEVAMD_SINGLE_PASS=0")
            hints[batch_size] = min(live, n_rows) / float(_ROWS_PER_FLAGSHIP_FRAME)
    return hints[batch_size]


def _frames_equivalent(lvl):
    """Frames per step as the tilings understand them: from the live row count when it is known, else the batch size."""
    return float(lvl.batch) if lvl.frames_hint is None else lvl.frames_hint
# BEVAMD_SPCONV_SLAB_DIRECT=0: build the slab metadata from the int32 neighbour table instead of straight from the rank index
_SLAB_DIRECT = os.environ.get("BEVAMD_SPCONV_SLAB_DIRECT", "1") != "0"


def _slab_overrides():
    spec = os.environ.get("BEVAMD_SPCONV_SLAB_VARIANTS", "")
    out = {}
    for item in filter(None, spec.split(",")):
        c, v = item.split(":")
        out[int(c)] = int(v)
    return out


def _narrow_variant_for(conv, lvl, cin, cout):
    """Variant of the narrow-row slab kernels for this layer, None if it does not apply (3x3x3, cin <= 16, cout 16 | 32, input
    level in linear order with a sorted-key index)."""
    if os.environ.get("BEVAMD_SPCONV_SLAB_NARROW", "1") == "0" or not lvl.use_sorted():
        return None
    if not (cout in (16, 32) and (cout == 16 or cin > 8)):
        return None
    overrides = _slab_overrides()
    if conv.subm:
        variant = overrides.get(16, _SLAB_NARROW_SUBM)
        rows = ops.slab_block_rows(cin, variant)
        return variant if rows and ops.slab_grid_ok(lvl.shape, rows) else None
    variant = overrides.get(-16, _SLAB_NARROW_STRIDED)
    if not ops.slab_block_rows(cin, variant) and -16 not in overrides:
        variant = 3000128          # cin <= 8 -> 16: the both-tiles-in-one-wave shape (3100128) is built for 16 -> 32 only
    return variant if ops.slab_block_rows(cin, variant) else None


# neighbour table of the strided gather layers from the output side (rank-index lookups, no fill); 0 = from the input side
_DOWN_NBR_FROM_OUTPUTS = os.environ.get("BEVAMD_SPCONV_DOWN_NBR", "outputs") != "inputs"
# output set of a strided layer over a level in linear order: bitmap tiles in LDS from contiguous input row ranges (0: byte map)
_DOWN_SORTED = os.environ.get("BEVAMD_SPCONV_DOWN_SORTED", "1") != "0"
_STATUS_WORDS = 64
_STATUS_POOL = os.environ.get("BEVAMD_SPCONV_STATUS_POOL", "1") != "0"
_STATUS_POOLS = {}


def _status_pool(dev):
    """[zeroed int32 words, next free]: the status words of one pass's products.  ONE tensor per device, zeroed when it is created
    and whenever a reader finds a bit set (geometry_status) — not per pass (that was a fill kernel in front of every encoder
    pass): the kernels only ever OR bits into a word on an error, nobody looks at the words of a pass that is not checked, and a
    checked pass clears what it reads.  The first pass on a device that happens to be a graph capture gets a tensor of its own
    (allocated and zeroed inside the capture, not kept)."""
    if not _STATUS_POOL:
        return None
    key = torch.device(dev).index
    if key not in _STATUS_POOLS:
        words = torch.zeros(_STATUS_WORDS, dtype=torch.int32, device=dev)
        if torch.cuda.is_current_stream_capturing():
            return [words, 0]
        _STATUS_POOLS[key] = words
    return [_STATUS_POOLS[key], 0]
_PAD_CAST = os.environ.get("BEVAMD_SPCONV_PAD_CAST", "1") != "0"
_GATHER_SLOT_ROWS = 128
# measured at 8 frames and NOT the default: LiDAR branch 3.94 ms with it against 3.83 ms with the int32 tables — the 2-byte slot
# decode makes the gather kernels 6-13 % slower (196 / 130 us against 185 / 115) and the extra sorted-key searches (128 + 60 us)
# run under the level-2 layers, which is worth more than the 0.32 ms of table kernels they replace
_GATHER_SLOTS = os.environ.get("BEVAMD_SPCONV_GATHER_SLOTS", "0") == "1"


def _gather_reads_slots(conv, lvl):
    """Strided 3x3x3 layers left on the gather kernels (32 -> 64, 64 -> 128: their input ranges are too long to stage) read slot
    metadata built by sorted-key search from the input level instead of an int32 table that has to be cleared and scattered."""
    return (_GATHER_SLOTS and _SORTED and lvl.allow_slab and not conv.subm and tuple(conv.kernel_size) == (3, 3, 3)
            and lvl.linear_order)


def _slab_variant_for(conv, lvl, cin, cout):
    if os.environ.get("BEVAMD_SPCONV_SLAB", "1") == "0" or not lvl.allow_slab:
        return None
    if tuple(conv.kernel_size) != (3, 3, 3) or not lvl.linear_order:
        return None
    if ops.padded_channels(cin) <= 16:
        return _narrow_variant_for(conv, lvl, cin, cout)
    if not (conv.subm and cin == cout):
        return None
    frames = _frames_equivalent(lvl)
    if cin not in _SLAB_DEFAULT or frames < _SLAB_MIN_BATCH.get(cin, 1) - 0.5:
        return None
    default = _SLAB_DEFAULT_SMALL_BATCH[cin] if frames < _SLAB_SMALL_BATCH_BELOW.get(cin, 0.0) else _SLAB_DEFAULT[cin]
    variant = _slab_overrides().get(cin, default)
    rows = ops.slab_block_rows(cin, variant)
    if rows == 0 or not ops.slab_grid_ok(lvl.shape, rows):
        return None
    return variant


def _variant_for(batch, K, cin, cout):
    """`batch`: frames per step, or the frames-equivalent of the level's live rows (_frames_equivalent)."""
    if batch >= 5.5 and K == 27:   # measured neutral at 4 frames (3.15 vs 3.17 ms per step), +4 % at 8
        return _BATCHED_VARIANTS.get((ops.padded_channels(cin), cout), 0)
    return 0


def _sequential(seq, x):
    from ..sparse_block import SparseBasicBlock
    from .conv import SparseConvolution
    from .modules import SparseSequential

    mods = list(seq._modules.values())
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, SparseConvolution):
            bn, relu = None, False
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d):
                bn = mods[i + 1]
                i += 1
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                relu = True
                i += 1
            x = _conv(m, x, bn, relu)
        elif isinstance(m, SparseBasicBlock):
            x = _basic_block(m, x)
        elif isinstance(m, SparseSequential):
            x = _sequential(m, x)
        else:
            raise Unfusable(f"module {type(m).__name__} inside a SparseSequential")
        i += 1
    return x


def _basic_block(block, x):
    """sparse_block.py:88-107: conv1-bn1-relu, conv2-bn2, + identity, relu."""
    if block.downsample is not None:
        raise Unfusable("SparseBasicBlock with a downsample branch")
    if not isinstance(block.norm1, nn.BatchNorm1d) or not isinstance(block.norm2, nn.BatchNorm1d):
        raise Unfusable("SparseBasicBlock norm is not BatchNorm1d")
    y = _conv(block.conv1, x, block.norm1, relu=True)
    if y.level is not x.level:
        raise Unfusable("strided conv inside a residual block")
    return _conv(block.conv2, y, block.norm2, relu=True, residual=x.features)


def dense_bev(x):
    """[B, C*Z, X, Y] dense tensor of a FusedTensor (sparse_encoder.py:126-131)."""
    lib = _capi.load()
    lvl = x.level
    lvl.ensure_index()
    lvl._await(lvl.ready)
    C = x.features.shape[1]
    X, Y, Z = lvl.shape
    out = torch.empty((lvl.batch, C * Z, X, Y), dtype=x.features.dtype, device=x.features.device)
    with torch.cuda.device(out.device):
        rc = lib.bevamd_spconv_dense_bev(_capi.ptr(x.features), x.features.element_size(), x.features.stride(0), C,
                                         lvl.index_kind, _capi.ptr(lvl.index), lvl.index_n_cap, lvl.batch,
                                         _capi.ints(lvl.shape), _capi.ptr(out), _capi.stream_ptr(out.device))
    _capi.check(rc, "spconv_dense_bev")
    return out


def unsupported_reason(enc, voxel_features):
    """None when this call can take the fused path, else a one-line reason (logged once by SparseEncoder)."""
    if enc.training:
        return "module is in training mode"
    if torch.is_grad_enabled():
        return "autograd is enabled (wrap inference in torch.no_grad())"
    if not voxel_features.is_cuda:
        return "input is not on the GPU"
    dtype = enc.conv_input[0].weight.dtype
    if dtype not in (torch.float16, torch.bfloat16):
        return f"weights are {dtype} (the fused path runs fp16 / bf16 weights: .half() or .bfloat16() the encoder)"
    return None


def encoder_supported(enc, voxel_features):
    return unsupported_reason(enc, voxel_features) is None


@torch.no_grad()
def prepare_geometry(enc, coors, batch_size, num_voxels=None, coors_order=None, allow_slab=True):
    """Everything of `run_encoder` that depends on voxel COORDINATES only — hash index, every level's active set, neighbour
    tables, slab metadata — issued on the current stream (and finished there: the geometry stream, if any, is joined back).
    Returns the level-1 `Level` holding the products; pass it to `SparseEncoder.forward(..., geometry=level)` /
    `run_encoder(..., geometry=level)`.  Lets a caller run the rulebook chain (latency-bound integer kernels) underneath
    something else — bench.py runs it, with the voxelizer, beside the HBM-bound camera branch — instead of underneath the
    encoder's own convolutions."""
    coors = coors.int().contiguous()
    n = coors.shape[0]
    if n == 0:
        raise NotThisCall("empty input")
    if num_voxels is not None:
        num_voxels = num_voxels.reshape(-1)[:1].int().contiguous()
    dev = coors.device
    g = geometry_stream(dev)
    main = torch.cuda.current_stream(dev)
    pool = _status_pool(dev)
    if g is not None:
        g.wait_stream(main)
    lvl = Level(coors, n, num_voxels, batch_size, enc.sparse_shape, gstream=g, linear_order=_is_linear(coors_order), status_pool=pool,
                allow_slab=allow_slab)
    lvl.frames_hint = _frames_hint(enc, int(batch_size), n, num_voxels)
    try:
        prefetch_geometry(enc, lvl)
    finally:
        if g is not None:
            main.wait_stream(g)
    # the products are complete on `main` from here on: later consumers need no per-product event
    _drop_events(lvl)
    redo = _first_call_check(enc, lvl)
    if redo is not None:
        lvl = prepare_geometry(enc, coors, batch_size, num_voxels=num_voxels, **dict(dict(coors_order=coors_order, allow_slab=allow_slab), **redo))
        enc.__dict__["_bevamd_geometry_checked"] = False   # the clean re-run does not vouch for the caller's next input
    return lvl


def _drop_events(lvl):
    seen = set()
    while lvl is not None and id(lvl) not in seen:
        seen.add(id(lvl))
        lvl.ready = None
        lvl.gstream = None
        lvl._subm = {k: (v[0], None) for k, v in lvl._subm.items()}
        lvl._slab = {k: (v[0], None) for k, v in lvl._slab.items()}
        lvl._down_slab = {k: (v[0], None) for k, v in lvl._down_slab.items()}
        lvl._sorted_ev = None
        nxt = None
        for out, _ in lvl._down.values():
            nxt = out
        lvl = nxt


def _first_call_check(enc, lvl):
    """The device-side status words of a rulebook chain (bit 0: a slab range did not fit its 16-bit slots, neighbours were
    dropped; bit 1: rows promised as coors_order="linear" were not in ascending linear index / inside the grid) are invisible to
    a sync-free caller.  Kernels stay inside their buffers either way, but the convolutions would run on a wrong rulebook.
    The FIRST eager call of an encoder therefore reads them back (one host sync, never under graph capture; later calls trust
    the data, BEVAMD_SPCONV_CHECK=1 checks every call).  Returns None (all clear) or the keyword overrides the caller re-runs
    with: the order-free route (hash index + gather kernels at level 1) for a broken promise, gather kernels throughout for an
    overflowing range (dense synthetic grids; LiDAR sweeps are two orders of magnitude below the limit)."""
    if enc.__dict__.get("_bevamd_geometry_checked") or torch.cuda.is_current_stream_capturing():
        return None
    bits = geometry_status(lvl)
    if not bits:
        enc.__dict__["_bevamd_geometry_checked"] = True   # stays unset after a failure: keep checking until a call is clean
        return None
    import warnings

    redo = {}
    if bits & 2:
        warnings.warn("SparseEncoder: coors_order='linear' was promised but the voxel rows are not in ascending linear index "
                      "(or leave the grid); falling back to the order-free rulebook route for this call", RuntimeWarning)
        redo["coors_order"] = None
    if bits & 1:
        warnings.warn("SparseEncoder: a staged-rows range exceeded its 16-bit slots (an input plane denser than the slab kernels "
                      "admit); falling back to the gather kernels for this call", RuntimeWarning)
        redo["allow_slab"] = False
    return redo


def _is_linear(coors_order):
    if coors_order in (None, "first", "any"):
        return False
    if coors_order in ("linear", "key"):
        return True
    raise ValueError(f"coors_order must be None or 'linear', got {coors_order!r}")


def geometry_status(lvl):
    """OR of the device status words of every structure built for `lvl` and the levels below it (one host sync; tests and
    BEVAMD_SPCONV_CHECK=1): bit 0 = a slab range overflowed its 16-bit slots, bit 1 = rows promised to be in linear order
    were not."""
    bits, seen = 0, set()
    while lvl is not None and id(lvl) not in seen:
        seen.add(id(lvl))
        words = [m.status for m, _ in list(lvl._slab.values()) + list(lvl._down_slab.values())]
        if lvl.sorted_status is not None:
            words.append(lvl.sorted_status)
        for wd in words:
            if wd is not None:
                v = int(wd.item())
                if v:
                    bits |= v
                    wd.zero_()      # the words live in a per-device pool that is not re-zeroed per pass (_status_pool)
        nxt = None
        for out, _ in lvl._down.values():
            nxt = out
        lvl = nxt
    return bits


def run_encoder(enc, voxel_features, coors, batch_size, num_voxels=None, geometry=None, coors_order=None, allow_slab=True):
    """SparseEncoder.forward on the fused path.  voxel_features [N, C_in] (any float dtype), coors [N, 4] int32
    (batch, x, y, z); `num_voxels` (optional int32 device tensor [1]): live row count when the inputs are
    capacity-padded buffers straight from the voxelizer (`voxelize_batch(..., sync=False)`); `geometry`: the Level returned
    by `prepare_geometry` for these coordinates (the rulebook chain is then already built and ordered before this call)."""
    dtype = enc.conv_input[0].weight.dtype
    n = voxel_features.shape[0]
    if n == 0:
        raise NotThisCall("empty input")
    cin = enc.in_channels if hasattr(enc, "in_channels") else voxel_features.shape[1]
    pitch = ops.padded_channels(cin)
    if voxel_features.dtype == dtype and voxel_features.shape[1] == pitch and voxel_features.is_contiguous() and pitch != cin:
        feats = voxel_features      # the voxelizer already wrote the padded 16-bit rows (voxelize_batch_device(encoder_rows=...))
    elif voxel_features.shape[1] != cin:
        raise RuntimeError(f"run_encoder: voxel_features has {voxel_features.shape[1]} columns, the encoder reads {cin}")
    elif _PAD_CAST and voxel_features.dtype == torch.float32 and voxel_features.is_contiguous():
        feats = torch.empty((n, pitch), dtype=dtype, device=voxel_features.device)
        with torch.cuda.device(voxel_features.device):
            rc = _capi.load().bevamd_spconv_pad_cast_rows(_capi.ptr(voxel_features), n, cin, pitch, ops._dtype_code(feats),
                                                          _capi.ptr(feats), _capi.stream_ptr(voxel_features.device))
        _capi.check(rc, "spconv_pad_cast_rows")
    else:
        feats = torch.zeros((n, pitch), dtype=dtype, device=voxel_features.device)
        feats[:, :cin] = voxel_features
    if geometry is not None:
        if geometry.n_cap != n or geometry.batch != int(batch_size):
            raise RuntimeError("run_encoder: `geometry` was prepared for other inputs")
        x = FusedTensor(feats, geometry)
        x = _sequential(enc.conv_input, x)
        x = _sequential(enc.encoder_layers, x)
        x = _sequential(enc.conv_out, x)
        return dense_bev(x)
    coors = coors.int().contiguous()
    if num_voxels is not None:
        num_voxels = num_voxels.reshape(-1)[:1].int().contiguous()
    dev = voxel_features.device
    g = geometry_stream(dev)
    main = torch.cuda.current_stream(dev)
    pool = _status_pool(dev)
    if g is not None:
        g.wait_stream(main)       # fork: coordinates / count are final, recycled buffers are quiescent
    lvl = Level(coors, n, num_voxels, batch_size, enc.sparse_shape, gstream=g, linear_order=_is_linear(coors_order), status_pool=pool,
                allow_slab=allow_slab)
    lvl.frames_hint = _frames_hint(enc, int(batch_size), n, num_voxels)
    redo = None
    try:
        if g is not None:
            if _CHAIN_LOOKAHEAD < 0:
                prefetch_geometry(enc, lvl)
            else:
                lvl.chain = _Chain(enc, lvl, _CHAIN_LOOKAHEAD)
        x = FusedTensor(feats, lvl)
        x = _sequential(enc.conv_input, x)
        x = _sequential(enc.encoder_layers, x)
        x = _sequential(enc.conv_out, x)
        out = dense_bev(x)
        redo = _first_call_check(enc, lvl)
        if redo is None and _CHECK:
            bits = geometry_status(lvl)
            if bits:
                raise RuntimeError(f"SparseEncoder fused path: geometry status {bits:#x} (1: slab range overflow, 2: coordinates "
                                   "passed with coors_order='linear' are not in ascending linear index)")
    finally:
        if g is not None:
            main.wait_stream(g)   # join: whatever follows on this stream is ordered behind the geometry kernels
    if redo is not None:
        out = run_encoder(enc, voxel_features, coors, batch_size, num_voxels=num_voxels,
                          **dict(dict(coors_order=coors_order, allow_slab=allow_slab), **redo))
        enc.__dict__["_bevamd_geometry_checked"] = False   # the clean re-run does not vouch for the caller's next input
    return out


def prefetch_geometry(enc, lvl):
    """Issue the whole rulebook chain on the geometry stream, in module (= execution) order, without waiting for any of
    it: hash index, SubM neighbour tables and downsamples of every level.  The feature pass then finds each product in
    its Level's cache and waits on its event only.  A tree whose module order differs from its execution order merely
    builds some tables later, on demand."""
    _PREFETCHING[0] = True
    try:
        for _ in _geometry_steps(enc, lvl):
            pass
    finally:
        _PREFETCHING[0] = False


def _chain_modules(enc):
    """The convolutions whose rulebook products the chain builds, in module order."""
    from .conv import SparseConvolution

    return [m for m in enc.modules()
            if isinstance(m, SparseConvolution) and not (m.conv1x1 or m.transposed or m.inverse or m.ndim != 3)
            and not any(d != 1 for d in m.dilation)]


def _geometry_steps(enc, lvl):
    """Generator: one step per convolution of `_chain_modules`, issuing (wait=False) whatever that layer reads."""
    cur = lvl
    for m in _chain_modules(enc):
        v = _slab_variant_for(m, cur, m.in_channels, m.out_channels)
        if m.subm:
            if v is not None:
                cur.subm_slab(ops.slab_block_rows(m.in_channels, v), wait=False)
            else:
                cur.subm_neighbors(m.kernel_size, wait=False)
        else:
            slots_meta = v is None and _gather_reads_slots(m, cur)
            nxt, _ = cur.downsample(m.kernel_size, m.stride, m.padding, wait=False,
                                    want_nbr=(v is None and not slots_meta) or LAYER_PROFILE is not None)
            if v is not None:
                cur.down_slab(m.kernel_size, m.stride, m.padding, ops.slab_block_rows(m.in_channels, v), wait=False)
            elif slots_meta:
                cur.down_slab(m.kernel_size, m.stride, m.padding, _GATHER_SLOT_ROWS, wait=False)
            cur = nxt
        yield m


class _Chain:
    """The rulebook chain of one encoder pass, issued on the geometry stream `lookahead` convolutions ahead of the feature pass
    instead of all at once in front of it.  On the GPU nothing changes at 8 frames (the host is far ahead either way); what
    changes is the order the nodes of a captured graph are CREATED in, and a replayed HIP graph enqueues its nodes in that
    order: with the whole chain issued first the first convolution of a single-frame step was enqueued behind ~45 rulebook
    nodes and started 190 us after its inputs were ready (tools/ubench/graph_order.py: a branch created after a 50-node chain
    starts 380 us late, created right after its fork node 5 us late).

    Products are allocated under the geometry stream (its pool): a buffer of the main stream's pool may be a block a
    still-running convolution reads (see Level._fork), and once convolutions have been issued the chain can no longer rely on
    the fork at the top of run_encoder for that.  They stay alive in their Level's caches until the join at the end of the pass."""

    def __init__(self, enc, lvl, lookahead):
        mods = _chain_modules(enc)
        self.strided = [not m.subm for m in mods]
        self.main = torch.cuda.current_stream(lvl.device)
        self.pos = {id(m): i for i, m in enumerate(mods)}
        self.steps = _geometry_steps(enc, lvl)
        self.issued = 0
        self.total = len(mods)
        self.lookahead = int(lookahead)
        self.gstream = lvl.gstream
        self.advance(self.lookahead)

    def advance(self, upto):
        """Issue the products of modules [issued, upto]."""
        if self.issued > upto or self.issued >= self.total:
            return
        _PREFETCHING[0] = True
        try:
            with torch.cuda.stream(self.gstream):
                while self.issued <= upto and self.issued < self.total:
                    if _CHAIN_HOLD and self.issued > 0 and self.strided[self.issued]:
                        # The products of a strided layer (the next level's active set and everything built on it) start behind the
                        # convolutions issued so far: they are not needed before the NEXT level's layers, and kept out from under
                        # the layers they used to run beside — the strided layer of the level before (16 -> 32: 248 us beside the
                        # level-3 set construction, 133 us alone); they land beside the first wide layers of the next level instead.
                        ev = torch.cuda.Event()
                        ev.record(self.main)
                        self.gstream.wait_event(ev)
                    next(self.steps)
                    self.issued += 1
        finally:
            _PREFETCHING[0] = False

    def before(self, conv):
        i = self.pos.get(id(conv))
        if i is not None:
            self.advance(i + self.lookahead)
