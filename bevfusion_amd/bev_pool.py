"""bev_pool — host-side mirror of `mmdet3d/ops/bev_pool/bev_pool.py` over the HIP C ABI.

Reference interface reproduced here (same names, argument order, shapes):
  * `bev_pool_ext.bev_pool_forward / bev_pool_backward`  (bev_pool_cpu.cpp:22-28,60-66, :89-94)
  * `QuickCumsumCuda`                                     (bev_pool.py:37-80)
  * `bev_pool(feats, coords, B, D, H, W) -> [B, C, D, H, W]`  (bev_pool.py:83-97)

MI355X-native additions: `BevPoolPlan` (the rank/sort/interval precompute as a cached,
sync-free device object) and the indexed kernels that read features through the sort
permutation instead of materialising `feats[indices]`.

Host tensors: `bev_pool()` runs the reference's device-agnostic `QuickCumsum` formulation (bev_pool.py:8-34) in torch —
BASELINE configs[0], plumbing on a box without a GPU.  It is not a fallback: GPU tensors go through the HIP library or raise.
"""
import os

import torch

from . import _capi

__all__ = ["bev_pool", "bev_pool_ext", "QuickCumsum", "QuickCumsumCuda", "BevPoolPlan"]


_FUSED_SCHEDULE = os.environ.get("BEVAMD_FUSED_POOL_SCHEDULE", "1") != "0"   # 0: frame-major walk over all XCDs
# fused pooling formulation: "columns" (default; csrc/bev_pool_fused_cols.hip: one partial row per run of an image column, then a
# per-cell reduce) falls back to the cell-centric kernels when the shape or the plan does not suit it; "cells" forces those
_FUSED_MODE = os.environ.get("BEVAMD_FUSED_POOL_MODE", "columns")
# a column plan is used while it has at most this fraction of the kept points as runs (flagship rig: 1 run per 28 points;
# a camera rolled by 90 degrees: 1 per point -> the cell-centric kernel does the same work without the second pass)
_FUSED_COLUMNS_MAX_RUN_FRACTION = float(os.environ.get("BEVAMD_FUSED_COLUMNS_MAX_RUN_FRACTION", "0.25"))
_BWD_POINTS = os.environ.get("BEVAMD_BEV_POOL_BWD_POINTS", "1") != "0"   # 0: the row-parallel backward (sorted-row order)


def _require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: the HIP extension has no CPU path")


class _BevPoolExt:
    """Drop-in for the reference's pybind module `bev_pool_ext`."""

    @staticmethod
    def bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, b, d, h, w):
        _require_cuda(x, "x")
        lib = _capi.load()
        b, d, h, w = int(b), int(d), int(h), int(w)
        n, c = x.shape
        out = torch.empty((b, d, h, w, c), dtype=torch.float32, device=x.device)
        geom_feats = geom_feats.contiguous()
        interval_lengths = interval_lengths.contiguous()
        interval_starts = interval_starts.contiguous()
        if geom_feats.dtype != torch.int32 or interval_lengths.dtype != torch.int32 or interval_starts.dtype != torch.int32:
            raise RuntimeError("geom_feats / interval_lengths / interval_starts must be int32")
        x = x.contiguous()
        with torch.cuda.device(x.device):
            if x.dtype == torch.float32:
                fn = lib.bevamd_bev_pool_forward
            elif x.dtype == torch.bfloat16:
                fn = lib.bevamd_bev_pool_forward_bf16
            else:
                raise RuntimeError(f"bev_pool_forward: unsupported dtype {x.dtype}")
            rc = fn(_capi.ptr(x), _capi.ptr(geom_feats), _capi.ptr(interval_lengths), _capi.ptr(interval_starts),
                    _capi.ptr(out), n, c, interval_lengths.shape[0], b, d, h, w, _capi.stream_ptr(x.device))
        _capi.check(rc, "bev_pool_forward")
        return out

    @staticmethod
    def bev_pool_backward(out_grad, geom_feats, interval_lengths, interval_starts, b, d, h, w):
        _require_cuda(out_grad, "out_grad")
        lib = _capi.load()
        b, d, h, w = int(b), int(d), int(h), int(w)
        out_grad = out_grad.contiguous().float()
        n = geom_feats.shape[0]
        c = out_grad.shape[4]
        x_grad = torch.empty((n, c), dtype=torch.float32, device=out_grad.device)
        with torch.cuda.device(out_grad.device):
            rc = lib.bevamd_bev_pool_backward(
                _capi.ptr(out_grad), _capi.ptr(geom_feats.contiguous()), _capi.ptr(interval_lengths.contiguous()),
                _capi.ptr(interval_starts.contiguous()), _capi.ptr(x_grad), n, c, interval_lengths.shape[0],
                b, d, h, w, 0, _capi.stream_ptr(out_grad.device))
        _capi.check(rc, "bev_pool_backward")
        return x_grad


bev_pool_ext = _BevPoolExt()


class QuickCumsum(torch.autograd.Function):
    """The reference's device-agnostic formulation (bev_pool.py:8-34), the only form of the op it can run without its
    extension: rows sorted by rank in, one row per rank out — prefix sum over the rows, keep the LAST row of every run of equal
    ranks, difference of consecutive kept prefix sums.  Pure torch; this is what `bev_pool()` runs for HOST tensors (BASELINE
    configs[0]: plumbing on a box without a GPU).  GPU tensors never come here: they go through the HIP library or fail.

    One deviation, on purpose: the prefix sum is carried in float64 (a 300 k-row fp32 prefix sum differenced back loses ~1e-3
    of a cell's value) and the interval sums are rounded to the input dtype at the end."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks):
        last_of_run = torch.ones(x.shape[0], device=x.device, dtype=torch.bool)
        last_of_run[:-1] = ranks[1:] != ranks[:-1]
        prefix = x.to(torch.float64).cumsum(0)[last_of_run]
        sums = prefix.clone()
        sums[1:] -= prefix[:-1]
        geom_out = geom_feats[last_of_run]
        ctx.save_for_backward(last_of_run)
        ctx.mark_non_differentiable(geom_out)
        return sums.to(x.dtype), geom_out

    @staticmethod
    def backward(ctx, grad_sums, grad_geom):
        (last_of_run,) = ctx.saved_tensors
        # row i belongs to the interval that closes at the first kept row >= i: count the kept rows strictly before i
        interval_of_row = torch.cumsum(last_of_run, 0) - last_of_run.to(torch.int64)
        return grad_sums[interval_of_row], None, None


def _bev_pool_host(feats, coords, B, D, H, W):
    """`bev_pool()` for host tensors: rank -> argsort -> QuickCumsum -> dense scatter (bev_pool.py:83-97 with the device-agnostic
    reduction in place of the extension call).  -> [B, C, D, H, W], differentiable w.r.t. feats."""
    coords = coords.long()
    ranks = coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B) + coords[:, 2] * B + coords[:, 3]
    order = ranks.argsort(stable=True)
    feats, coords, ranks = feats[order], coords[order], ranks[order]
    out = feats.new_zeros((B, D, H, W, feats.shape[1]))
    if feats.shape[0]:
        sums, cells = QuickCumsum.apply(feats, coords, ranks)
        out = out.index_put((cells[:, 3], cells[:, 2], cells[:, 0], cells[:, 1]), sums)
    return out.permute(0, 4, 1, 2, 3).contiguous()


class QuickCumsumCuda(torch.autograd.Function):
    """Same contract as the reference class (bev_pool.py:37-80): sorted feats/coords/ranks in,
    dense [B, D, H, W, C] out.  Interval boundaries are found on the host side of the op exactly
    as the reference does (torch ops; one D2H sync in torch.where)."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks, B, D, H, W):
        kept = torch.ones(x.shape[0], device=x.device, dtype=torch.bool)
        kept[1:] = ranks[1:] != ranks[:-1]
        interval_starts = torch.where(kept)[0].int()
        interval_lengths = torch.zeros_like(interval_starts)
        interval_lengths[:-1] = interval_starts[1:] - interval_starts[:-1]
        interval_lengths[-1] = x.shape[0] - interval_starts[-1]
        geom_feats = geom_feats.int()
        out = bev_pool_ext.bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, B, D, H, W)
        ctx.save_for_backward(interval_starts, interval_lengths, geom_feats)
        ctx.saved_shapes = B, D, H, W
        return out

    @staticmethod
    def backward(ctx, out_grad):
        interval_starts, interval_lengths, geom_feats = ctx.saved_tensors
        B, D, H, W = ctx.saved_shapes
        out_grad = out_grad.contiguous()
        x_grad = bev_pool_ext.bev_pool_backward(out_grad, geom_feats, interval_lengths, interval_starts, B, D, H, W)
        return x_grad, None, None, None, None, None, None


class BevPoolPlan:
    """Device-resident result of the bev_pool precompute (rank -> stable sort -> cell CSR).

    Built once per camera calibration and reused every frame at inference; building it never
    synchronises with the host.  Holds `order` (sorted row -> input row), `ranks_sorted` and the
    CSR `cell_start` over rank-ordered cells; the reference-shaped interval arrays are optional
    (`want_intervals`) and exist for API parity / tests.
    """

    def __init__(self, n, B, D, H, W, device, want_intervals, want_geom):
        self.n, self.B, self.D, self.H, self.W = int(n), int(B), int(D), int(H), int(W)
        self.device = device
        self.ncells = self.B * self.D * self.H * self.W
        n_alloc = max(self.n, 1)
        self.ranks_sorted = torch.empty(n_alloc, dtype=torch.int32, device=device)
        self.order = torch.empty(n_alloc, dtype=torch.int32, device=device)
        self.cell_start = torch.empty(self.ncells + 2, dtype=torch.int32, device=device)
        self.interval_starts = self.interval_lengths = self.n_intervals_dev = self.geom_sorted = None
        self._cell_of_point = None   # rank per frustum point in point order (built on first fused backward)
        self._fused_sched = {}       # (depth_bins, fh, fw) -> (perm, xcd_start): camera-sector walk of the fused pooling
        self._fused_cols = {}        # (depth_bins, fh, fw) -> _ColumnPlan: column formulation of the fused pooling
        if want_intervals:
            cap = max(min(self.n, self.ncells), 1)
            self.interval_starts = torch.empty(cap, dtype=torch.int32, device=device)
            self.interval_lengths = torch.empty(cap, dtype=torch.int32, device=device)
            self.n_intervals_dev = torch.zeros(1, dtype=torch.int32, device=device)
        if want_geom:
            self.geom_sorted = torch.empty((n_alloc, 4), dtype=torch.int32, device=device)

    def _workspace(self, lib):
        nbytes = lib.bevamd_bev_pool_prepare_workspace_bytes(self.n, self.B, self.D, self.H, self.W)
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device), nbytes

    # -- builders ---------------------------------------------------------------------------
    @classmethod
    def from_coords(cls, coords, B, D, H, W, want_intervals=False, want_geom=False):
        """coords: [N, 4] (x, y, z, b), int32 or int64 — the tensor `bev_pool()` receives."""
        _require_cuda(coords, "coords")
        lib = _capi.load()
        if coords.dtype not in (torch.int32, torch.int64):
            coords = coords.long()
        coords = coords.contiguous()
        plan = cls(coords.shape[0], B, D, H, W, coords.device, want_intervals, want_geom)
        with torch.cuda.device(coords.device):
            ws, ws_bytes = plan._workspace(lib)
            rc = lib.bevamd_bev_pool_prepare(
                _capi.ptr(coords), int(coords.dtype == torch.int64), plan.n, plan.B, plan.D, plan.H, plan.W,
                _capi.ptr(plan.ranks_sorted), _capi.ptr(plan.order), _capi.ptr(plan.cell_start),
                _capi.ptr(plan.interval_starts), _capi.ptr(plan.interval_lengths), _capi.ptr(plan.n_intervals_dev),
                _capi.ptr(plan.geom_sorted), _capi.ptr(ws), ws_bytes, _capi.stream_ptr(coords.device))
        _capi.check(rc, "bev_pool_prepare")
        return plan

    @classmethod
    def from_geometry(cls, geom_xyz, batch, bx_minus_half_dx, dx, nx, want_intervals=False, want_geom=False):
        """geom_xyz: [N', 3] fp32 frustum points in the lidar frame (batch-major), unfiltered.
        Implements vtransforms/base.py:149-169 (truncation, batch index, range mask) + the
        prologue of bev_pool.py:83-93 in one device pipeline."""
        _require_cuda(geom_xyz, "geom_xyz")
        lib = _capi.load()
        geom_xyz = geom_xyz.contiguous().float()
        H, W, D = int(nx[0]), int(nx[1]), int(nx[2])
        plan = cls(geom_xyz.shape[0], batch, D, H, W, geom_xyz.device, want_intervals, want_geom)
        o = _capi.float3(bx_minus_half_dx)
        s = _capi.float3(dx)
        with torch.cuda.device(geom_xyz.device):
            ws, ws_bytes = plan._workspace(lib)
            rc = lib.bevamd_bev_pool_prepare_from_geom(
                _capi.ptr(geom_xyz), plan.n, plan.B, plan.D, plan.H, plan.W, o, s,
                _capi.ptr(plan.ranks_sorted), _capi.ptr(plan.order), _capi.ptr(plan.cell_start),
                _capi.ptr(plan.interval_starts), _capi.ptr(plan.interval_lengths), _capi.ptr(plan.n_intervals_dev),
                _capi.ptr(plan.geom_sorted), _capi.ptr(ws), ws_bytes, _capi.stream_ptr(geom_xyz.device))
        _capi.check(rc, "bev_pool_prepare_from_geom")
        return plan

    # -- host-visible scalars (these DO synchronise; for tests and the reference-shaped API) ---
    def n_intervals(self):
        if self.n_intervals_dev is None:
            raise RuntimeError("plan was built without want_intervals=True")
        return int(self.n_intervals_dev.item())

    def n_kept(self):
        return int(self.cell_start[self.ncells].item())

    # -- kernels ----------------------------------------------------------------------------
    def forward(self, feats):
        """feats: [N, C] fp32 or bf16, UNSORTED (row i belongs to coords[i]).  -> [B, D, H, W, C] fp32."""
        return _PlannedBevPool.apply(feats, self)

    def launch_forward(self, feats, out=None):
        lib = _capi.load()
        feats = feats.contiguous()
        if feats.shape[0] != self.n:
            raise RuntimeError(f"feats has {feats.shape[0]} rows, plan was built for {self.n}")
        if feats.dtype == torch.float32:
            is_bf16 = 0
        elif feats.dtype == torch.bfloat16:
            is_bf16 = 1
        else:
            raise RuntimeError(f"bev_pool: unsupported feature dtype {feats.dtype}")
        c = feats.shape[1]
        if out is None:
            out = torch.empty((self.B, self.D, self.H, self.W, c), dtype=torch.float32, device=feats.device)
        with torch.cuda.device(feats.device):
            rc = lib.bevamd_bev_pool_forward_cells(
                _capi.ptr(feats), is_bf16, _capi.ptr(self.order), _capi.ptr(self.cell_start), _capi.ptr(out),
                self.n, c, self.B, self.D, self.H, self.W, _capi.stream_ptr(feats.device))
        _capi.check(rc, "bev_pool_forward_cells")
        return out

    def launch_fused(self, depth, ctx, depth_bins, fh, fw, out=None, mode=None):
        """Fused depth (x) context -> BEV: out[cell] = sum_p depth[p] * ctx[pixel(p)] without the [N', C] volume.  depth: fp32,
        `self.n` elements in frustum-point order ([cams, D, fH, fW] flattened); ctx [cams*fH*fW, C] channels-last fp32 / bf16.
        Forward only.  mode: None = BEVAMD_FUSED_POOL_MODE ("columns": csrc/bev_pool_fused_cols.hip when the shape and the plan
        suit it, else the cell-centric kernels of csrc/bev_pool_fused.hip), "cells" = cell-centric, "columns!" = the column
        formulation or an error (tests).  Calls on ONE plan share the partial-row scratch: issue them on one stream."""
        lib = _capi.load()
        depth = depth.contiguous()
        ctx = ctx.contiguous()
        if depth.dtype != torch.float32 or depth.numel() != self.n:
            raise RuntimeError(f"depth must be fp32 with {self.n} elements, got {depth.dtype} x {depth.numel()}")
        if ctx.dtype == torch.float32:
            is_bf16 = 0
        elif ctx.dtype == torch.bfloat16:
            is_bf16 = 1
        else:
            raise RuntimeError(f"bev_pool fused: unsupported context dtype {ctx.dtype}")
        if not depth.is_cuda or not ctx.is_cuda:
            raise RuntimeError("bev_pool fused: inputs must be GPU tensors (the HIP extension has no CPU path)")
        c = ctx.shape[-1]
        if ctx.numel() // c * depth_bins != self.n:
            raise RuntimeError("ctx rows x depth_bins must equal the number of frustum points of the plan")
        if out is None:
            out = torch.empty((self.B, self.D, self.H, self.W, c), dtype=torch.float32, device=ctx.device)
        capturing = torch.cuda.is_current_stream_capturing()
        mode = mode or _FUSED_MODE
        cols = None
        if mode in ("columns", "columns!") and self.n > 0:
            cols = self.fused_columns(depth_bins, fh, fw, c, build=not capturing, force=mode == "columns!")
        if mode == "columns!" and cols is None:
            raise RuntimeError("bev_pool fused: the column formulation does not support this shape (c % 4, fh <= 32, fw % 4) "
                               "or its plan is not built yet (graph capture)")
        if cols is not None:
            with torch.cuda.device(ctx.device):
                partial = cols.partial_rows(c)
                rc = lib.bevamd_bev_pool_fused_forward_columns(
                    _capi.ptr(depth), _capi.ptr(ctx), is_bf16, _capi.ptr(cols.keep), _capi.ptr(cols.end), _capi.ptr(cols.run_first),
                    _capi.ptr(cols.slot_of_run), _capi.ptr(cols.prow_start), _capi.ptr(partial), _capi.ptr(out), self.n, cols.nruns,
                    c, int(depth_bins), int(fh), int(fw), self.B, self.D, self.H, self.W, _capi.stream_ptr(ctx.device))
            _capi.check(rc, "bev_pool_fused_forward_columns")
            return out
        # cell-centric kernels: the camera-sector schedule needs whole frames of depth_bins*fh*fw points per camera and is built
        # with a sort — not while a graph is being captured (the sort would be replayed and its buffers would live in the graph's
        # pool): the unscheduled kernel computes the same bits
        per_frame_ok = self.n % (int(depth_bins) * int(fh) * int(fw) * self.B) == 0
        key = (int(depth_bins), int(fh), int(fw))
        sched = None
        if _FUSED_SCHEDULE and self.n > 0 and per_frame_ok and (not capturing or key in self._fused_sched):
            sched = self.fused_schedule(depth_bins, fh, fw)
        with torch.cuda.device(ctx.device):
            if sched is not None:
                rc = lib.bevamd_bev_pool_fused_forward_scheduled(
                    _capi.ptr(depth), _capi.ptr(ctx), is_bf16, _capi.ptr(self.order), _capi.ptr(self.cell_start),
                    _capi.ptr(sched[0]), _capi.ptr(sched[1]), _capi.ptr(out), self.n, c, int(depth_bins), int(fh), int(fw),
                    self.B, self.D, self.H, self.W, _capi.stream_ptr(ctx.device))
            else:
                rc = lib.bevamd_bev_pool_fused_forward(
                    _capi.ptr(depth), _capi.ptr(ctx), is_bf16, _capi.ptr(self.order), _capi.ptr(self.cell_start), _capi.ptr(out),
                    self.n, c, int(depth_bins), int(fh), int(fw), self.B, self.D, self.H, self.W, _capi.stream_ptr(ctx.device))
        _capi.check(rc, "bev_pool_fused_forward")
        return out

    def fused_schedule(self, depth_bins, fh, fw):
        """(perm, xcd_start) of the fused pooling's camera-sector walk for this plan (csrc/bev_pool_fused.hip): built once per
        (plan, frustum shape) on the device — one radix sort of the cells — and cached; static per calibration like the plan."""
        key = (int(depth_bins), int(fh), int(fw))
        if key not in self._fused_sched:
            lib = _capi.load()
            ncells = self.B * self.D * self.H * self.W
            perm = torch.empty(ncells, dtype=torch.int32, device=self.device)
            cuts = torch.empty(9, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                wsb = lib.bevamd_bev_pool_fused_schedule_workspace_bytes(ncells)
                ws = torch.empty(max(int(wsb), 1), dtype=torch.uint8, device=self.device)
                rc = lib.bevamd_bev_pool_fused_schedule(_capi.ptr(self.order), _capi.ptr(self.cell_start), self.n, key[0], key[1],
                                                        key[2], self.B, self.D, self.H, self.W, _capi.ptr(perm), _capi.ptr(cuts),
                                                        _capi.ptr(ws), wsb, _capi.stream_ptr(self.device))
            _capi.check(rc, "bev_pool_fused_schedule")
            self._fused_sched[key] = (perm, cuts)
        return self._fused_sched[key]

    def fused_columns(self, depth_bins, fh, fw, c=80, build=True, force=False):
        """The column plan of the fused pooling for this plan and frustum shape (`_ColumnPlan`), or None when the shape is not
        supported or the plan is long-tailed (about as many runs as points).  Built once on the device and cached — static per
        calibration like the plan itself; building reads ONE uint32 back (the run count sizes the buffers), so it is never done
        under graph capture (`build=False` returns what is cached): call `prepare_fused()` before capturing."""
        key = (int(depth_bins), int(fh), int(fw))
        lib = _capi.load()
        if not (self.n > 0 and self.n % (key[0] * key[1] * key[2]) == 0 and lib.bevamd_bev_pool_fused_columns_supported(int(c), *key)):
            return None
        if key not in self._fused_cols:
            if not build:
                return None
            self._fused_cols[key] = _ColumnPlan.build(self, *key)
        cols = self._fused_cols[key]
        # `force` (tests): also for long-tailed plans — about as many runs as points, where the second pass buys nothing
        if not force and cols.nruns > _FUSED_COLUMNS_MAX_RUN_FRACTION * max(cols.n_kept, 1):
            return None
        return cols

    def prepare_fused(self, depth_bins, fh, fw, c=80):
        """Build whatever `launch_fused` would build lazily for this frustum shape (column plan, or the camera-sector schedule of the
        cell-centric kernel) — call once before capturing `launch_fused` into a HIP graph."""
        if _FUSED_MODE == "columns" and self.fused_columns(depth_bins, fh, fw, c) is not None:
            return
        if _FUSED_SCHEDULE and self.n > 0 and self.n % (int(depth_bins) * int(fh) * int(fw) * self.B) == 0:
            self.fused_schedule(depth_bins, fh, fw)

    def cell_of_point(self):
        if self._cell_of_point is None:
            lib = _capi.load()
            cop = torch.empty(max(self.n, 1), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                rc = lib.bevamd_bev_pool_cell_of_point(_capi.ptr(self.order), _capi.ptr(self.ranks_sorted), self.n,
                                                       _capi.ptr(cop), _capi.stream_ptr(self.device))
            _capi.check(rc, "bev_pool_cell_of_point")
            self._cell_of_point = cop
        return self._cell_of_point

    def launch_fused_backward(self, out_grad, depth, ctx, depth_bins, fh, fw):
        """(d_depth like depth, d_ctx like ctx) of `launch_fused` for fp32 context."""
        lib = _capi.load()
        out_grad = out_grad.contiguous().float()
        depth, ctx = depth.contiguous(), ctx.contiguous()
        if ctx.dtype != torch.float32:
            raise RuntimeError("bev_pool fused backward: fp32 context only")
        c = ctx.shape[-1]
        d_ctx = torch.empty_like(ctx)
        cop = self.cell_of_point()
        cols = None
        if (_FUSED_MODE == "columns" and self.n > 0
                and lib.bevamd_bev_pool_fused_backward_columns_supported(int(c), int(depth_bins), int(fh), int(fw))):
            cols = self.fused_columns(depth_bins, fh, fw, c, build=not torch.cuda.is_current_stream_capturing())
        if cols is not None:
            # column formulation (csrc/bev_pool_fused_cols.hip): the depth gradient comes back with an image column's values
            # contiguous, [cams, fw, depth_bins, fh], and is handed on as a permuted view of that buffer
            cams = self.n // (int(depth_bins) * int(fh) * int(fw))
            d_depth_t = torch.empty((cams, int(fw), int(depth_bins), int(fh)), dtype=torch.float32, device=ctx.device)
            with torch.cuda.device(ctx.device):
                rc = lib.bevamd_bev_pool_fused_backward_columns(
                    _capi.ptr(out_grad), _capi.ptr(depth), _capi.ptr(ctx), _capi.ptr(cols.keep), _capi.ptr(cols.end), _capi.ptr(cop),
                    _capi.ptr(d_depth_t), _capi.ptr(d_ctx), self.n, c, int(depth_bins), int(fh), int(fw), self.B, self.D, self.H,
                    self.W, _capi.stream_ptr(ctx.device))
            _capi.check(rc, "bev_pool_fused_backward_columns")
            return d_depth_t.permute(0, 2, 3, 1).reshape(depth.shape), d_ctx   # a view unless `depth` was flat (then one copy)
        d_depth = torch.empty_like(depth)
        with torch.cuda.device(ctx.device):
            rc = lib.bevamd_bev_pool_fused_backward(
                _capi.ptr(out_grad), _capi.ptr(depth), _capi.ptr(ctx), _capi.ptr(cop), _capi.ptr(d_depth), _capi.ptr(d_ctx),
                self.n, c, int(depth_bins), int(fh), int(fw), self.B, self.D, self.H, self.W, _capi.stream_ptr(ctx.device))
        _capi.check(rc, "bev_pool_fused_backward")
        return d_depth, d_ctx

    def fused(self, depth, ctx, depth_bins, fh, fw):
        """Differentiable fused depth (x) context -> BEV: [B, D, H, W, C] fp32 (autograd through `launch_fused_backward`)."""
        return _FusedPool.apply(depth, ctx, self, int(depth_bins), int(fh), int(fw))

    def launch_backward(self, out_grad, c):
        lib = _capi.load()
        out_grad = out_grad.contiguous().float()
        x_grad = torch.empty((self.n, c), dtype=torch.float32, device=out_grad.device)
        if c % 4 == 0 and _BWD_POINTS:
            # point-order walk: a streaming write of x_grad, the gather is on the (cached) cell gradients
            cop = self.cell_of_point()
            with torch.cuda.device(out_grad.device):
                rc = lib.bevamd_bev_pool_backward_points(_capi.ptr(out_grad), _capi.ptr(cop), _capi.ptr(x_grad), self.n, c,
                                                         self.B, self.D, self.H, self.W, _capi.stream_ptr(out_grad.device))
            _capi.check(rc, "bev_pool_backward_points")
            return x_grad
        with torch.cuda.device(out_grad.device):
            rc = lib.bevamd_bev_pool_backward_rows(
                _capi.ptr(out_grad), _capi.ptr(self.order), _capi.ptr(self.ranks_sorted), _capi.ptr(x_grad),
                self.n, c, self.B, self.D, self.H, self.W, _capi.stream_ptr(out_grad.device))
        _capi.check(rc, "bev_pool_backward_rows")
        return x_grad


class _ColumnPlan:
    """Column formulation of the fused pooling (csrc/bev_pool_fused_cols.hip), static per (plan, frustum shape): row masks per
    image column, the slot of every run in the (frame, cell)-sorted partial-row buffer, the CSR of that buffer over frame-major
    cells, and the buffer itself (scratch, reused by every call)."""

    def __init__(self):
        self.keep = self.end = self.run_first = self.slot_of_run = self.prow_start = None
        self.nruns = self.n_kept = 0
        self._partial = None

    @classmethod
    def build(cls, plan, depth_bins, fh, fw):
        lib = _capi.load()
        dev = plan.device
        self = cls()
        ncols = plan.n // fh
        cop = plan.cell_of_point()
        self.keep = torch.empty(ncols, dtype=torch.int32, device=dev)
        self.end = torch.empty(ncols, dtype=torch.int32, device=dev)
        self.run_first = torch.empty(ncols, dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int32, device=dev)
        self.prow_start = torch.empty(plan.ncells + 1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            wsb = lib.bevamd_bev_pool_fused_columns_workspace_bytes(ncols, 0)
            ws = torch.empty(max(int(wsb), 1), dtype=torch.uint8, device=dev)
            rc = lib.bevamd_bev_pool_fused_columns_count(_capi.ptr(cop), plan.n, depth_bins, fh, fw, plan.B, plan.D, plan.H, plan.W,
                                                         _capi.ptr(self.keep), _capi.ptr(self.end), _capi.ptr(self.run_first),
                                                         _capi.ptr(total), _capi.ptr(ws), wsb, _capi.stream_ptr(dev))
            _capi.check(rc, "bev_pool_fused_columns_count")
            self.nruns = int(total.item())                      # the one read-back of the plan (sizes slot_of_run / the partial rows)
            self.n_kept = plan.n_kept()
            self.slot_of_run = torch.empty(max(self.nruns, 1), dtype=torch.int32, device=dev)
            wsb = lib.bevamd_bev_pool_fused_columns_workspace_bytes(ncols, self.nruns)
            ws = torch.empty(max(int(wsb), 1), dtype=torch.uint8, device=dev)
            rc = lib.bevamd_bev_pool_fused_columns_build(_capi.ptr(cop), _capi.ptr(self.end), _capi.ptr(self.run_first), plan.n,
                                                         self.nruns, depth_bins, fh, fw, plan.B, plan.D, plan.H, plan.W,
                                                         _capi.ptr(self.slot_of_run), _capi.ptr(self.prow_start), _capi.ptr(ws), wsb,
                                                         _capi.stream_ptr(dev))
            _capi.check(rc, "bev_pool_fused_columns_build")
        return self

    def partial_rows(self, c):
        if self._partial is None or self._partial.shape[1] != c:
            self._partial = torch.empty((max(self.nruns, 1), c), dtype=torch.float32, device=self.keep.device)
        return self._partial


class _PlannedBevPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, plan):
        ctx.plan = plan
        ctx.c = feats.shape[1]
        ctx.in_dtype = feats.dtype
        return plan.launch_forward(feats)

    @staticmethod
    def backward(ctx, out_grad):
        g = ctx.plan.launch_backward(out_grad, ctx.c)
        return g.to(ctx.in_dtype), None


class _FusedPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx_, depth, ctx, plan, depth_bins, fh, fw):
        depth32, ctx32 = depth.float().contiguous(), ctx.contiguous()
        ctx_.plan, ctx_.dims, ctx_.dtypes = plan, (depth_bins, fh, fw), (depth.dtype, ctx.dtype)
        ctx_.save_for_backward(depth32, ctx32)
        return plan.launch_fused(depth32.reshape(-1), ctx32, depth_bins, fh, fw)

    @staticmethod
    def backward(ctx_, out_grad):
        depth32, ctx32 = ctx_.saved_tensors
        d_depth, d_ctx = ctx_.plan.launch_fused_backward(out_grad, depth32, ctx32.float(), *ctx_.dims)
        return d_depth.to(ctx_.dtypes[0]), d_ctx.to(ctx_.dtypes[1]), None, None, None, None


def bev_pool(feats, coords, B, D, H, W, plan=None, channels_last_view=False):
    """Drop-in for `mmdet3d.ops.bev_pool(feats, coords, B, D, H, W)` (bev_pool.py:83-97).

    feats [N, C]; coords [N, 4] integer (x, y, z, b) -> [B, C, D, H, W].
    `plan` lets a caller reuse the precompute across frames (static calibration);
    `channels_last_view=True` skips the final permute copy and returns the [B,D,H,W,C]
    buffer viewed as [B,C,D,H,W] (same values, channels-last strides).
    """
    assert feats.shape[0] == coords.shape[0]
    B, D, H, W = int(B), int(D), int(H), int(W)
    if not feats.is_cuda and not coords.is_cuda:
        # host tensors: the reference's device-agnostic QuickCumsum pipeline in torch (BASELINE configs[0]); never a fallback for
        # GPU tensors — those need the HIP library
        if plan is not None:
            raise RuntimeError("bev_pool: a BevPoolPlan is a GPU object; host tensors take the torch QuickCumsum route")
        return _bev_pool_host(feats, coords, B, D, H, W)
    if plan is None:
        plan = BevPoolPlan.from_coords(coords, B, D, H, W)
    x = plan.forward(feats)
    x = x.permute(0, 4, 1, 2, 3)
    if not channels_last_view:
        x = x.contiguous()
    return x
