"""spconv ops — host-side mirror of `mmdet3d/ops/spconv/ops.py` + the pybind module `sparse_conv_ext`
(spconv/src/all.cc:21-51) over the HIP C ABI.

Two layers:
  * `Rulebook` + `build_rulebook` / `sparse_conv` / `sparse_conv_backward`: the native, output-stationary
    path the modules use (one launch per convolution, rulebooks cached, no per-offset GEMM loop);
  * `sparse_conv_ext.*` and the reference-named helpers (`get_indice_pairs`, `indice_conv`,
    `indice_conv_backward`, ...) with the reference's exact signatures and array shapes, for drop-in use.
"""
import os

import torch

from .. import _capi

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _dtype_code(t):
    if t.dtype not in _DT:
        raise RuntimeError(f"spconv: unsupported dtype {t.dtype} (fp32 / fp16 / bf16)")
    return _DT[t.dtype]


def _require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: the HIP extension has no CPU path")


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    """ops.py:20-31."""
    ndim = len(input_size)
    output_size = []
    for i in range(ndim):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        if kernel_size[i] == -1:
            output_size.append(1)
        else:
            output_size.append(size)
    return output_size


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    """ops.py:34-42."""
    ndim = len(input_size)
    output_size = []
    for i in range(ndim):
        if kernel_size[i] == -1:
            raise ValueError("deconv don't support kernel_size < 0")
        size = (input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i]
        output_size.append(size)
    return output_size


def _as_list(v, ndim):
    if not isinstance(v, (list, tuple)):
        return [int(v)] * ndim
    return [int(t) for t in v]


# --------------------------------------------------------------------------------------------
# native rulebook
# --------------------------------------------------------------------------------------------
class Rulebook:
    """Output-stationary rulebook of one sparse convolution.

    nbr [K, stride] int32: nbr[k, o] = input row feeding output row o through kernel offset k, or -1.
    `indice_pairs()` materialises the reference-shaped arrays on demand."""

    def __init__(self, out_indices, nbr, num_out, num_in, kernel_volume, subm, out_spatial_shape, symmetric=None):
        # symmetric: offset k of row o reads row i  <=>  offset K-1-k of row i reads row o.  True for a SubM
        # rulebook with odd kernel sizes and no dilation (padding k/2 centres the window only then).
        self.symmetric = bool(subm) if symmetric is None else bool(symmetric)
        self.out_indices = out_indices
        self.nbr = nbr
        self.num_out = int(num_out)
        self.num_in = int(num_in)
        self.kernel_volume = int(kernel_volume)
        self.subm = bool(subm)
        self.out_spatial_shape = list(out_spatial_shape)
        self._nbr_t = None
        self._pairs = None
        self._conv_tables = None
        self._slab128 = None

    def slab_meta_wgrad(self, cin):
        """Slab metadata (128-row blocks, the filter gradient's address format for `cin` channels: csrc/spconv_slab_meta.h FMT_WG64 /
        FMT_WG32) of a symmetric 3x3x3 SubM rulebook whose rows turn out to be in (near-)linear order — what the staged-rows filter
        gradient reads — or None.  Built from the neighbour table on first use and shared by the convolutions of a level (they share
        the rulebook through `indice_key`).  Whether the rows are ordered is MEASURED, not promised: the mean staged range of a block
        must stay below 4 blocks (a level in linear order stages little more than its 128 rows per plane; first-appearance order
        stages thousands) and no range may overflow the 16-bit slots — one small reduction and one host read-back per level."""
        code = int(_capi.load().bevamd_spconv_wgrad_slab_block_rows(int(cin)))
        if code == 0:
            return None
        if self._slab128 is None:
            self._slab128 = {}
        if code not in self._slab128:
            meta = None
            if (self.subm and self.symmetric and self.kernel_volume == 27 and self.num_out >= 128 and self.num_in == self.num_out
                    and os.environ.get("BEVAMD_SPCONV_WGRAD_SLAB", "1") != "0" and False not in self._slab128.values()):
                m = slab_build(self.nbr, self.num_out, None, code)
                nblk = (self.num_out + 127) // 128
                cnt = m.hdr[:nblk * 24].view(torch.int32).view(nblk, 3, 2)[:, :, 1] & 0x3FFFFFFF
                staged, flag = (int(v) for v in torch.stack((cnt.sum(dtype=torch.int64), m.status[0].to(torch.int64))).tolist())
                meta = m if flag == 0 and staged <= 4 * 128 * 3 * nblk else False
            self._slab128[code] = meta if meta is not None else False
        return self._slab128[code] or None

    @property
    def nbr_stride(self):
        return self.nbr.shape[1]

    def conv_tables(self):
        """(nbr, nbr_t) the convolution kernels walk.  They are the rulebook itself except for a SubM rulebook without
        the mirror symmetry (dilated, or even kernel sizes), where the reference's identity shortcut is applied to the
        offset with the most pairs (see `_subm_shortcut_`) so that results match its indice_subm_conv."""
        if not self.subm or self.symmetric:
            return self.nbr, self.nbr_transposed()
        if self._conv_tables is None:
            counts = (self.nbr[:, :max(self.num_out, 1)] >= 0).sum(1)
            nbr = _subm_shortcut_(self.nbr.clone(), counts, self.num_out)
            nbr_t = _subm_shortcut_(self.nbr_transposed().clone(), counts, self.num_in)
            self._conv_tables = (nbr, nbr_t)
        return self._conv_tables

    def nbr_transposed(self):
        """Input-stationary view (rows = inputs) for the input-gradient pass."""
        if self._nbr_t is None:
            if self.subm and self.symmetric:
                # submanifold symmetry: in = out + (k - c)  <=>  out = in + ((K-1-k) - c)
                self._nbr_t = self.nbr.flip(0).contiguous()
            else:
                lib = _capi.load()
                t = torch.empty((self.kernel_volume, max(self.num_in, 1)), dtype=torch.int32, device=self.nbr.device)
                with torch.cuda.device(self.nbr.device):
                    rc = lib.bevamd_spconv_transpose_nbr(_capi.ptr(self.nbr), self.nbr_stride, self.num_out,
                                                         self.kernel_volume, _capi.ptr(t), t.shape[1],
                                                         _capi.stream_ptr(self.nbr.device))
                _capi.check(rc, "spconv_transpose_nbr")
                self._nbr_t = t
        return self._nbr_t

    def indice_pairs(self):
        """(indice_pairs [K,2,num_in] int32 -1 padded, indice_num [K] int32) as spconv_ops.h:56-59."""
        if self._pairs is None:
            lib = _capi.load()
            dev = self.nbr.device
            # third dimension = number of INPUT rows, as the reference allocates it (spconv_ops.h:56-58); an
            # input row appears at most once per offset, so no offset can hold more pairs than that
            K, L = self.kernel_volume, max(self.num_in, 1)
            pairs = torch.empty((K, 2, L), dtype=torch.int32, device=dev)
            num = torch.empty((K,), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                wsb = lib.bevamd_spconv_pairs_workspace_bytes(self.num_out, K)
                ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                rc = lib.bevamd_spconv_pairs_from_nbr(_capi.ptr(self.nbr), self.nbr_stride, self.num_out, K,
                                                      _capi.ptr(pairs), L, _capi.ptr(num), _capi.ptr(ws), wsb,
                                                      _capi.stream_ptr(dev))
            _capi.check(rc, "spconv_pairs_from_nbr")
            if self.num_in == 0:
                pairs = pairs[:, :, :0]
            self._pairs = (pairs, num)
        return self._pairs


def _lift_to_3d(indices, lists, ndim):
    """2D geometry as 3D with a unit last axis (kernel 1, stride 1, padding 0, dilation 1): the offset numbering
    (kx*Ky + ky) and the ascending-linear-index row order are unchanged by it."""
    if ndim == 3:
        return indices, lists
    pad_col = torch.zeros((indices.shape[0], 1), dtype=indices.dtype, device=indices.device)
    indices = torch.cat([indices, pad_col], dim=1)
    fill = dict(shape=1, out_shape=1, ksize=1, stride=1, padding=0, dilation=1)
    return indices, {k: list(v) + [fill[k]] for k, v in lists.items()}


def build_rulebook(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1, subm=False,
                   transpose=False, out_padding=0, out_shape=None):
    """Native counterpart of `get_indice_pairs` (ops.py:45-125): returns a `Rulebook` (and syncs once, for strided
    convs, to learn the number of active outputs — tensor shapes need it on the host).  2D and 3D; `transpose` builds
    the rulebook of a transposed convolution (geometry.h:86-141)."""
    _require_cuda(indices, "indices")
    lib = _capi.load()
    ndim = indices.shape[1] - 1
    if ndim not in (2, 3):
        raise NotImplementedError(f"{ndim}D sparse convolution is not implemented (2D and 3D are)")
    ksize, stride, padding, dilation, out_padding = (_as_list(v, ndim) for v in (ksize, stride, padding, dilation,
                                                                                out_padding))
    in_shape = [int(s) for s in spatial_shape]
    if subm:
        out_shape = list(in_shape)
    elif out_shape is not None:          # the pybind entry points take it from the caller (spconv_ops.h:30)
        out_shape = [int(v) for v in out_shape]
    elif transpose:
        out_shape = get_deconv_output_size(in_shape, ksize, stride, padding, dilation, out_padding)
    else:
        out_shape = get_conv_output_size(in_shape, ksize, stride, padding, dilation)
    user_out_shape = list(out_shape)
    user_indices = indices
    indices = indices.contiguous()
    if indices.dtype != torch.int32:
        indices = indices.int()
    indices, g = _lift_to_3d(indices, dict(shape=in_shape, out_shape=out_shape, ksize=ksize, stride=stride,
                                           padding=padding, dilation=dilation), ndim)
    in_shape, out_shape, ksize, stride, padding, dilation = (g[k] for k in ("shape", "out_shape", "ksize", "stride",
                                                                            "padding", "dilation"))
    n = indices.shape[0]
    dev = indices.device
    K = ksize[0] * ksize[1] * ksize[2]
    ks, st, dl = _capi.ints(ksize), _capi.ints(stride), _capi.ints(dilation)
    import ctypes

    host = ctypes.c_int(0)
    count_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        cap = max(int(lib.bevamd_spconv_max_outputs_ex(n, ks, st, dl, int(bool(subm)), int(bool(transpose)))), 1)
        if not subm:
            vol = batch_size * out_shape[0] * out_shape[1] * out_shape[2]
            cap = max(min(cap, vol), 1)
        nbr = torch.empty((K, cap), dtype=torch.int32, device=dev)
        out_indices = indices if subm else torch.empty((cap, 4), dtype=torch.int32, device=dev)
        wsb = lib.bevamd_spconv_rulebook_workspace_bytes(n, int(batch_size), _capi.ints(out_shape), int(bool(subm)))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = lib.bevamd_spconv_build_rulebook(
            _capi.ptr(indices), n, int(batch_size), _capi.ints(in_shape), _capi.ints(out_shape), ks, st,
            _capi.ints(padding), dl, int(bool(subm)), int(bool(transpose)), _capi.ptr(out_indices), cap, _capi.ptr(nbr),
            cap, _capi.ptr(count_dev), ctypes.byref(host), _capi.ptr(ws), wsb, _capi.stream_ptr(dev))
    _capi.check(rc, "spconv_build_rulebook")
    m = int(host.value)
    if subm:
        out_indices = user_indices
    else:
        out_indices = out_indices[:m]
        if ndim == 2:
            out_indices = out_indices[:, :3].contiguous()
    symmetric = bool(subm) and all(k % 2 == 1 for k in ksize) and all(d == 1 for d in dilation)
    return Rulebook(out_indices, nbr, m, n, K, subm, user_out_shape, symmetric)


def inverse_rulebook(rulebook, in_indices, in_spatial_shape):
    """Rulebook of the "inverse" convolution coupled to `rulebook` (conv.py:153-158): the same pairs with the roles
    of inputs and outputs swapped, so its output rows are the coupled convolution's input rows."""
    inv = Rulebook(in_indices, rulebook.nbr_transposed(), rulebook.num_in, rulebook.num_out, rulebook.kernel_volume,
                   False, in_spatial_shape, symmetric=False)
    inv._nbr_t = rulebook.nbr
    return inv


def prepare_filters(filters, transpose_io=False):
    """MFMA-friendly filter image [Cout_pad][K][Cin_pad] of `filters` [kx,ky,kz,Cin,Cout] (a ~microsecond
    kernel).  Not cached here: a (data_ptr, version) key is unsafe across tensor lifetimes; modules cache it
    against their own Parameter (see conv.py)."""
    lib = _capi.load()
    f = filters.detach().contiguous()
    cin, cout = f.shape[-2], f.shape[-1]
    K = f.numel() // (cin * cout)
    dt = _dtype_code(f)
    with torch.cuda.device(f.device):
        elems = lib.bevamd_spconv_prepared_filter_elems(dt, K, cin, cout, int(transpose_io))
        out = torch.empty(elems, dtype=f.dtype, device=f.device)
        rc = lib.bevamd_spconv_prepare_filters(_capi.ptr(f), dt, K, cin, cout, int(transpose_io), _capi.ptr(out),
                                               _capi.stream_ptr(f.device))
    _capi.check(rc, "spconv_prepare_filters")
    return out


def tiled_supported(dtype, cin, cout):
    """True when the tiled MFMA kernels (csrc/spconv_tile.h) serve this convolution: 16-bit features, <= 128 channels."""
    if dtype not in (torch.float16, torch.bfloat16):
        return False
    return bool(_capi.load().bevamd_spconv_tiled_supported(_DT[dtype], int(cin), int(cout)))


def padded_channels(cin):
    """Row pitch (elements) the tiled kernels read for `cin` input channels."""
    for c in (8, 16, 32, 64, 128):
        if cin <= c:
            return c
    raise RuntimeError(f"tiled sparse conv supports at most 128 channels, got {cin}")


def make_filter_image(filters, transpose_io=False):
    """Filter [kx,ky,kz,Cin,Cout] (16-bit) -> MFMA-fragment-ordered image for `sparse_conv_tiled`.  Build once per
    weight (modules cache it against their Parameter)."""
    lib = _capi.load()
    f = filters.detach().contiguous()
    _require_cuda(f, "filters")
    cin, cout = f.shape[-2], f.shape[-1]
    K = f.numel() // (cin * cout)
    dt = _dtype_code(f)
    with torch.cuda.device(f.device):
        elems = lib.bevamd_spconv_filter_image_elems(K, cin, cout, int(transpose_io))
        if elems == 0:
            raise RuntimeError(f"no tiled kernel for {cin} -> {cout} channels")
        img = torch.empty(elems, dtype=f.dtype, device=f.device)
        rc = lib.bevamd_spconv_make_filter_image(_capi.ptr(f), dt, K, cin, cout, int(transpose_io), _capi.ptr(img),
                                                 _capi.stream_ptr(f.device))
    _capi.check(rc, "spconv_make_filter_image")
    return img


def make_filter_images(specs, dtype, device):
    """Filter images of many convolutions in ONE launch (bevamd_spconv_make_filter_images).  specs: list of (filters
    [kx,ky,kz,Cin,Cout] fp32 or `dtype`, transpose_io, mirror) -> list of image tensors (views of one buffer).  `mirror` reverses the
    kernel offsets (k -> K - 1 - k): with transpose_io it is the filter the input gradient of a symmetric SubM layer runs over the
    layer's own table."""
    import ctypes

    lib = _capi.load()
    out = []
    for lo in range(0, len(specs), 48):
        part = specs[lo:lo + 48]
        n = len(part)
        Ks, cins, couts, flags, elems, srcs = [], [], [], [], [], []
        for f, tr, mirror in part:
            f = f.detach()
            if not f.is_contiguous():
                f = f.contiguous()
            if f.dtype not in (torch.float32, dtype):
                f = f.to(dtype)
            cin, cout = f.shape[-2], f.shape[-1]
            K = f.numel() // (cin * cout)
            e = int(lib.bevamd_spconv_filter_image_elems(K, cin, cout, int(bool(tr))))
            if e == 0:
                raise RuntimeError(f"no tiled kernel for {cin} -> {cout} channels")
            Ks.append(K); cins.append(cin); couts.append(cout); elems.append((e + 127) // 128 * 128); srcs.append(f)
            flags.append(int(bool(tr)) | (2 if mirror else 0) | (4 if f.dtype == torch.float32 else 0))
        buf = torch.empty(sum(elems), dtype=dtype, device=device)
        imgs, at = [], 0
        for e in elems:
            imgs.append(buf[at:at + e])
            at += e
        VP = ctypes.c_void_p * n
        with torch.cuda.device(device):
            rc = lib.bevamd_spconv_make_filter_images(n, VP(*[f.data_ptr() for f in srcs]), VP(*[i.data_ptr() for i in imgs]), _capi.ints(Ks),
                                                      _capi.ints(cins), _capi.ints(couts), _capi.ints(flags), _DT[dtype],
                                                      _capi.stream_ptr(device))
        _capi.check(rc, "spconv_make_filter_images")
        out.extend(imgs)
    return out


def sparse_conv_tiled(features, image, nbr, num_out, kernel_volume, cin, cout, bias=None, bn_scale=None, bn_shift=None,
                      residual=None, relu=False, num_out_dev=None, out=None, variant=0):
    """One fused launch: out[o] = relu?(bn_scale * (sum_k features[nbr[k, o]] @ W[k] + bias) + bn_shift + residual[o]).

    features [num_in, pitch >= padded_channels(cin)] 16-bit with zero padding channels; `image` from
    `make_filter_image`; `num_out` bounds the launch, `num_out_dev` (int32 device scalar, optional) is the live row
    count so that no host sync is needed.  Rows >= the live count of `out` are left untouched."""
    lib = _capi.load()
    _require_cuda(features, "features")
    if features.stride(1) != 1:
        features = features.contiguous()
    dt = _dtype_code(features)
    if out is None:
        out = torch.empty((num_out, cout), dtype=features.dtype, device=features.device)
    if num_out == 0:
        return out
    if residual is not None and residual.stride(1) != 1:
        residual = residual.contiguous()
    with torch.cuda.device(features.device):
        rc = lib.bevamd_spconv_conv_forward_tiled(
            _capi.ptr(features), dt, features.stride(0), features.shape[0], _capi.ptr(image), _capi.ptr(nbr),
            nbr.stride(0), int(num_out), _capi.ptr(num_out_dev), int(kernel_volume), int(cin), int(cout), _capi.ptr(out),
            out.stride(0), _capi.ptr(bias), _capi.ptr(bn_scale), _capi.ptr(bn_shift), _capi.ptr(residual),
            residual.stride(0) if residual is not None else 0, int(bool(relu)), int(variant),
            _capi.stream_ptr(features.device))
    _capi.check(rc, "spconv_conv_forward_tiled")
    return out


def sparse_conv_tiled_slots(features, image, meta, num_out, cin, cout, bias=None, bn_scale=None, bn_shift=None, residual=None,
                            relu=False, num_out_dev=None, out=None, variant=0):
    """`sparse_conv_tiled` for a 3x3x3 convolution whose rulebook is slab metadata (`SlabMeta`) instead of the int32 table."""
    lib = _capi.load()
    _require_cuda(features, "features")
    if features.stride(1) != 1:
        features = features.contiguous()
    dt = _dtype_code(features)
    if out is None:
        out = torch.empty((num_out, cout), dtype=features.dtype, device=features.device)
    if num_out == 0:
        return out
    if residual is not None and residual.stride(1) != 1:
        residual = residual.contiguous()
    with torch.cuda.device(features.device):
        rc = lib.bevamd_spconv_conv_forward_tiled_slots(
            _capi.ptr(features), dt, features.stride(0), features.shape[0], _capi.ptr(image), _capi.ptr(meta.hdr),
            _capi.ptr(meta.slots), meta.block_rows, int(num_out), _capi.ptr(num_out_dev), int(cin), int(cout), _capi.ptr(out),
            out.stride(0), _capi.ptr(bias), _capi.ptr(bn_scale), _capi.ptr(bn_shift), _capi.ptr(residual),
            residual.stride(0) if residual is not None else 0, int(bool(relu)), int(variant),
            _capi.stream_ptr(features.device))
    _capi.check(rc, "spconv_conv_forward_tiled_slots")
    return out


# ---- slab (staged-rows) SubM convolution: csrc/spconv_slab.h ------------------------------------------------------------
def slab_block_rows(cin, variant=0):
    """Rows per block of slab variant `variant` (0 = default) for a cin -> cin 3x3x3 SubM convolution; 0 = not built."""
    return int(_capi.load().bevamd_spconv_slab_block_rows(int(cin), int(variant)))


def slab_variants(cin):
    lib = _capi.load()
    codes = (_capi.c_int * 32)()
    n = lib.bevamd_spconv_slab_variants(int(cin), codes, 32)
    return [int(codes[i]) for i in range(min(n, 32))]


def slab_grid_ok(shape, block_rows):
    return bool(_capi.load().bevamd_spconv_slab_grid_ok(_capi.ints(shape), int(block_rows)))


class SlabMeta:
    """Block metadata of a 3x3x3 SubM neighbour table over rows in ascending linear index (bevamd_spconv_slab_build)."""

    def __init__(self, hdr, slots, block_rows, status):
        # block_rows: what bevamd_spconv_slab_block_rows returned — rows per block in the low half, the slot-format code in the
        # upper half (0 = raw 16-bit slots / the 64-byte baked form implied by 64-row blocks, 1 = baked 128-byte rows)
        self.hdr, self.slots, self.block_rows, self.status = hdr, slots, block_rows, status


def slab_build(nbr, m_cap, m_dev, block_rows, stream_ptr=None, status=None):
    lib = _capi.load()
    dev = nbr.device
    assert nbr.shape[0] == 27 and nbr.dtype == torch.int32
    with torch.cuda.device(dev):
        hdr = torch.empty(max(lib.bevamd_spconv_slab_hdr_bytes(int(m_cap), int(block_rows)), 16), dtype=torch.uint8, device=dev)
        slots = torch.empty(max(lib.bevamd_spconv_slab_slot_bytes(int(m_cap), int(block_rows)), 16), dtype=torch.uint8, device=dev)
        if status is None:   # callers on a hot path hand in a slice of one pre-zeroed pool (a 5 us fill kernel per product otherwise)
            status = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = lib.bevamd_spconv_slab_build(_capi.ptr(nbr), nbr.stride(0), int(m_cap), _capi.ptr(m_dev), int(block_rows),
                                          _capi.ptr(hdr), _capi.ptr(slots), _capi.ptr(status),
                                          stream_ptr if stream_ptr is not None else _capi.stream_ptr(dev))
    _capi.check(rc, "spconv_slab_build")
    return SlabMeta(hdr, slots, int(block_rows), status)


def slab_build_from_index(indices, m_cap, m_dev, batch, shape, index_kind, index, index_n_cap, block_rows, stream_ptr=None, status=None):
    """The same metadata straight from the voxel set's index (bevamd_spconv_slab_build_from_index): no neighbour table."""
    lib = _capi.load()
    dev = indices.device
    with torch.cuda.device(dev):
        hdr = torch.empty(max(lib.bevamd_spconv_slab_hdr_bytes(int(m_cap), int(block_rows)), 16), dtype=torch.uint8, device=dev)
        slots = torch.empty(max(lib.bevamd_spconv_slab_slot_bytes(int(m_cap), int(block_rows)), 16), dtype=torch.uint8, device=dev)
        if status is None:   # callers on a hot path hand in a slice of one pre-zeroed pool (a 5 us fill kernel per product otherwise)
            status = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = lib.bevamd_spconv_slab_build_from_index(_capi.ptr(indices), int(m_cap), _capi.ptr(m_dev), int(batch),
                                                     _capi.ints(shape), int(index_kind), _capi.ptr(index), int(index_n_cap),
                                                     int(block_rows), _capi.ptr(hdr), _capi.ptr(slots), _capi.ptr(status),
                                                     stream_ptr if stream_ptr is not None else _capi.stream_ptr(dev))
    _capi.check(rc, "spconv_slab_build_from_index")
    return SlabMeta(hdr, slots, int(block_rows), status)


def sorted_index_build(indices, n_cap, n_dev, batch, shape, stream_ptr=None, status=None):
    """Sorted-key index (keys + x-plane directory, one uint8 buffer) of a voxel set whose rows are in ascending linear index;
    returns (index, status) — status bit 1 (value 2) is set on the device if the rows are not strictly ascending."""
    lib = _capi.load()
    dev = indices.device
    with torch.cuda.device(dev):
        nbytes = int(lib.bevamd_spconv_sorted_index_bytes(int(n_cap), int(batch), _capi.ints(shape)))
        index = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        if status is None:   # callers on a hot path hand in a slice of one pre-zeroed pool (a 5 us fill kernel per product otherwise)
            status = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = lib.bevamd_spconv_sorted_index_build(_capi.ptr(indices), int(n_cap), _capi.ptr(n_dev), int(batch), _capi.ints(shape),
                                                  _capi.ptr(index), nbytes, _capi.ptr(status),
                                                  stream_ptr if stream_ptr is not None else _capi.stream_ptr(dev))
    _capi.check(rc, "spconv_sorted_index_build")
    return index, status


def slab_build_from_sorted(out_indices, m_cap, m_dev, batch, in_shape, out_shape, stride, padding, subm, in_index, in_n_cap,
                           block_rows, stream_ptr=None, status=None):
    """Slab metadata of a 3x3x3 convolution (submanifold, or strided with active outputs `out_indices`) from the sorted-key
    index of its input set (bevamd_spconv_slab_build_from_sorted): no neighbour table."""
    lib = _capi.load()
    dev = out_indices.device
    with torch.cuda.device(dev):
        hdr = torch.empty(max(lib.bevamd_spconv_slab_hdr_bytes(int(m_cap), int(block_rows)), 16), dtype=torch.uint8, device=dev)
        slots = torch.empty(max(lib.bevamd_spconv_slab_slot_bytes(int(m_cap), int(block_rows)), 16), dtype=torch.uint8, device=dev)
        if status is None:   # callers on a hot path hand in a slice of one pre-zeroed pool (a 5 us fill kernel per product otherwise)
            status = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = lib.bevamd_spconv_slab_build_from_sorted(_capi.ptr(out_indices), int(m_cap), _capi.ptr(m_dev), int(batch),
                                                      _capi.ints(in_shape), _capi.ints(out_shape), _capi.ints(stride),
                                                      _capi.ints(padding), int(bool(subm)), _capi.ptr(in_index), int(in_n_cap),
                                                      int(block_rows), _capi.ptr(hdr), _capi.ptr(slots), _capi.ptr(status),
                                                      stream_ptr if stream_ptr is not None else _capi.stream_ptr(dev))
    _capi.check(rc, "spconv_slab_build_from_sorted")
    return SlabMeta(hdr, slots, int(block_rows), status)


def sparse_conv_slab(features, image, meta, num_out, cin, cout, bias=None, bn_scale=None, bn_shift=None, residual=None,
                     relu=False, num_out_dev=None, out=None, variant=0):
    """`sparse_conv_tiled` for a 3x3x3 convolution over linear-index-ordered rows, through the slab kernels (SubM with
    cin == cout in {32, 64, 128}; cin <= 16 with cout in {16, 32}: SubM or strided, depending on what `meta` describes)."""
    lib = _capi.load()
    _require_cuda(features, "features")
    if features.stride(1) != 1:
        features = features.contiguous()
    dt = _dtype_code(features)
    if out is None:
        out = torch.empty((num_out, cout), dtype=features.dtype, device=features.device)
    if num_out == 0:
        return out
    if residual is not None and residual.stride(1) != 1:
        residual = residual.contiguous()
    with torch.cuda.device(features.device):
        rc = lib.bevamd_spconv_conv_forward_slab(
            _capi.ptr(features), dt, features.stride(0), features.shape[0], _capi.ptr(image), _capi.ptr(meta.hdr),
            _capi.ptr(meta.slots), meta.block_rows, int(num_out), _capi.ptr(num_out_dev), int(cin), int(cout), _capi.ptr(out),
            out.stride(0), _capi.ptr(bias), _capi.ptr(bn_scale), _capi.ptr(bn_shift), _capi.ptr(residual),
            residual.stride(0) if residual is not None else 0, int(bool(relu)), int(variant),
            _capi.stream_ptr(features.device))
    _capi.check(rc, "spconv_conv_forward_slab")
    return out


def _pad_channels(features, pitch):
    if features.shape[1] == pitch:
        return features
    return torch.nn.functional.pad(features, (0, pitch - features.shape[1]))


# fp32 features on the bf16 matrix cores by three-way operand splitting (csrc/spconv_tile_f32x3.hip, round 5): x = hi + mid + lo
# exactly, six MFMAs per product, fp32 accumulate — error of the order of one fp32 rounding, 2.5-4 x the rate of the exact-chain fp32
# kernel on the encoder's layers.  BEVAMD_SPCONV_F32X3 = "auto" (default: from _F32X3_MIN_ROWS output rows on, where the fp32
# training step spends its time; small problems keep the exact-chain kernel), "1" (whenever the shape is served), "0" (never).
_F32X3 = os.environ.get("BEVAMD_SPCONV_F32X3", "auto")
_F32X3_MIN_ROWS = 4096


def f32x3_supported(cin, cout):
    return bool(_capi.load().bevamd_spconv_f32x3_supported(int(cin), int(cout)))


def make_filter_image3(filters, transpose_io=False):
    """fp32 filter [kx,ky,kz,Cin,Cout] -> the three bf16 images (hi, mid, lo by truncation) in MFMA-fragment order."""
    lib = _capi.load()
    f = filters.detach().contiguous().float()
    _require_cuda(f, "filters")
    cin, cout = f.shape[-2], f.shape[-1]
    K = f.numel() // (cin * cout)
    with torch.cuda.device(f.device):
        elems = lib.bevamd_spconv_filter_image3_elems(K, cin, cout, int(transpose_io))
        if elems == 0:
            raise RuntimeError(f"no f32x3 kernel for {cin} -> {cout} channels")
        img = torch.empty(elems, dtype=torch.int16, device=f.device)
        rc = lib.bevamd_spconv_make_filter_image3(_capi.ptr(f), K, cin, cout, int(transpose_io), _capi.ptr(img), _capi.stream_ptr(f.device))
    _capi.check(rc, "spconv_make_filter_image3")
    return img


def sparse_conv_f32x3(features, image3, nbr, num_out, kernel_volume, cin, cout, bias=None, bn_scale=None, bn_shift=None,
                      residual=None, relu=False, num_out_dev=None, out=None):
    """fp32 rows [N, cin] (cin = 16 | 32 | 64 | 128) x make_filter_image3 -> fp32 [num_out, cout], same fused epilogue contract."""
    _require_cuda(features, "features")
    lib = _capi.load()
    if features.dtype != torch.float32:
        raise RuntimeError("sparse_conv_f32x3: fp32 features only")
    features = features.contiguous()
    if out is None:
        out = torch.empty((num_out, cout), dtype=torch.float32, device=features.device)
    if num_out == 0:
        return out
    bias = None if bias is None else bias.float().contiguous()
    residual = None if residual is None else residual.float().contiguous()
    with torch.cuda.device(features.device):
        rc = lib.bevamd_spconv_conv_forward_f32x3(
            _capi.ptr(features), features.stride(0), features.shape[0], _capi.ptr(image3), _capi.ptr(nbr), nbr.shape[1], int(num_out),
            _capi.ptr(num_out_dev), int(kernel_volume), int(cin), int(cout), _capi.ptr(out), out.stride(0), _capi.ptr(bias),
            _capi.ptr(bn_scale), _capi.ptr(bn_shift), _capi.ptr(residual), 0 if residual is None else residual.stride(0),
            int(bool(relu)), _capi.stream_ptr(features.device))
    _capi.check(rc, "spconv_conv_forward_f32x3")
    return out


def sparse_conv(features, filters, nbr, num_out, bias=None, bn_scale=None, bn_shift=None, residual=None, relu=False,
                prepared=None, transpose_io=False):
    """out[o] = epilogue(sum_k features[nbr[k, o]] @ W[k]) — one fused launch.  filters [kx,ky,kz,Cin,Cout]."""
    _require_cuda(features, "features")
    lib = _capi.load()
    features = features.contiguous()
    filters = filters.to(features.dtype) if filters.dtype != features.dtype else filters
    cin, cout = filters.shape[-2], filters.shape[-1]
    if transpose_io:
        cin, cout = cout, cin
    K = nbr.shape[0]
    if features.shape[1] != cin:
        raise RuntimeError(f"features have {features.shape[1]} channels, filters expect {cin}")
    if prepared is None and tiled_supported(features.dtype, cin, cout):
        # 16-bit features: tiled MFMA kernels (rows padded to the pitch they read)
        image = make_filter_image(filters, transpose_io)
        feats = _pad_channels(features, padded_channels(cin))
        if bias is not None:
            bias = bias.to(features.dtype).contiguous()
        if residual is not None:
            residual = residual.to(features.dtype)
        return sparse_conv_tiled(feats, image, nbr, num_out, K, cin, cout, bias=bias, bn_scale=bn_scale,
                                 bn_shift=bn_shift, residual=residual, relu=relu)
    # fp32 on the bf16 matrix cores by three-way operand splitting (csrc/spconv_tile_f32x3.hip): NOT the exact fp32 chain — three of
    # the nine piece products are dropped (<= 1.2e-6 of float64 relative to 1 + max|ref|), an Inf input becomes NaN (hi = Inf,
    # remainder = Inf - Inf) and bf16-range denormals of a piece may flush.  Its preconditions (16-byte aligned rows, a feature matrix
    # below 2 GiB) are checked here, so that a tensor it cannot serve takes the exact-chain kernel below instead of raising.
    if (prepared is None and features.dtype == torch.float32 and _F32X3 != "0" and f32x3_supported(cin, cout)
            and (_F32X3 == "1" or num_out >= _F32X3_MIN_ROWS)
            and features.data_ptr() % 16 == 0 and (features.stride(0) * 4) % 16 == 0
            and features.shape[0] * features.stride(0) * 4 < 2 ** 31
            and (residual is None or (residual.data_ptr() % 16 == 0 and (residual.stride(0) * 4) % 16 == 0))):
        return sparse_conv_f32x3(features, make_filter_image3(filters, transpose_io), nbr, num_out, K, cin, cout, bias=bias,
                                 bn_scale=bn_scale, bn_shift=bn_shift, residual=residual, relu=relu)
    if prepared is None:
        prepared = prepare_filters(filters, transpose_io)
    out = torch.empty((num_out, cout), dtype=features.dtype, device=features.device)
    if num_out == 0:
        return out
    dt = _dtype_code(features)
    if bias is not None:
        bias = bias.to(features.dtype).contiguous()
    if residual is not None:
        residual = residual.to(features.dtype).contiguous()
    with torch.cuda.device(features.device):
        rc = lib.bevamd_spconv_conv_forward(
            _capi.ptr(features), dt, _capi.ptr(prepared), _capi.ptr(nbr), nbr.shape[1], int(num_out), None, K, cin, cout,
            _capi.ptr(out), _capi.ptr(bias), _capi.ptr(bn_scale), _capi.ptr(bn_shift), _capi.ptr(residual), int(bool(relu)),
            _capi.stream_ptr(features.device))
    _capi.check(rc, "spconv_conv_forward")
    return out


def sparse_conv_wgrad_slab(features, out_grad, meta, cin, cout):
    """filter_grad [27, cin, cout] of a 3x3x3 SubM convolution from slab metadata (bevamd_spconv_conv_wgrad_slab)."""
    lib = _capi.load()
    dev = features.device
    fgrad = torch.empty((27, cin, cout), dtype=features.dtype, device=dev)
    with torch.cuda.device(dev):
        wsb = lib.bevamd_spconv_wgrad_slab_workspace_bytes(cin, cout)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = lib.bevamd_spconv_conv_wgrad_slab(_capi.ptr(features), features.stride(0), features.shape[0], _capi.ptr(out_grad),
                                               out_grad.stride(0), _dtype_code(features), _capi.ptr(meta.hdr), _capi.ptr(meta.slots),
                                               int(meta.block_rows), out_grad.shape[0], int(cin), int(cout), _capi.ptr(fgrad),
                                               _capi.ptr(ws), wsb, _capi.stream_ptr(dev))
    _capi.check(rc, "spconv_conv_wgrad_slab")
    return fgrad


def sparse_conv_backward(features, filters, out_grad, rulebook_nbr, nbr_t, num_in, rulebook=None):
    """(in_grad [num_in, Cin], filter_grad like filters) for out = sparse_conv(features, filters, nbr).  With the `rulebook`
    the tables came from, a SubM layer over ordered rows takes the staged-rows filter gradient."""
    lib = _capi.load()
    features = features.contiguous()
    out_grad = out_grad.contiguous().to(features.dtype)
    filters = filters.to(features.dtype) if filters.dtype != features.dtype else filters
    cin, cout = filters.shape[-2], filters.shape[-1]
    K = rulebook_nbr.shape[0]
    num_out = out_grad.shape[0]
    # input gradient: the same fused kernel on (out_grad, W^T, input-stationary table)
    in_grad = sparse_conv(out_grad, filters, nbr_t, num_in, transpose_io=True)
    # filter gradient
    if (rulebook is not None and features.dtype in (torch.float16, torch.bfloat16)
            and lib.bevamd_spconv_wgrad_slab_supported(_dtype_code(features), cin, cout)
            and features.stride(0) % 8 == 0 and out_grad.stride(0) % 8 == 0):
        meta = rulebook.slab_meta_wgrad(cin)
        if meta is not None:
            return in_grad, sparse_conv_wgrad_slab(features, out_grad, meta, cin, cout).view(filters.shape)
    fgrad = torch.empty_like(filters.contiguous())
    dt = _dtype_code(features)
    with torch.cuda.device(features.device):
        wsb = lib.bevamd_spconv_wgrad_workspace_bytes(K, cin, cout)
        ws = torch.empty(wsb, dtype=torch.uint8, device=features.device)
        rc = lib.bevamd_spconv_conv_wgrad(_capi.ptr(features), _capi.ptr(out_grad), dt, _capi.ptr(rulebook_nbr),
                                          rulebook_nbr.shape[1], num_out, K, cin, cout, _capi.ptr(fgrad), _capi.ptr(ws), wsb,
                                          _capi.stream_ptr(features.device))
    _capi.check(rc, "spconv_conv_wgrad")
    return in_grad, fgrad


# --------------------------------------------------------------------------------------------
# reference-named API (ops.py:45-211) and the pybind module's functions (all.cc:21-51)
# --------------------------------------------------------------------------------------------
def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1, out_padding=0,
                     subm=False, transpose=False, grid=None, out_shape=None):
    """ops.py:45-125 -> (out_indices [M,1+ndim], indice_pairs [K,2,N] int32, indice_num [K] int32).  `grid` (the
    reference's optional caller-owned dense grid) is accepted and unused: no dense grid exists here."""
    rb = build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm, transpose, out_padding,
                        out_shape)
    pairs, num = rb.indice_pairs()
    return rb.out_indices, pairs, num


def _nbr_from_pairs(indice_pairs, indice_num, num_rows, inverse):
    lib = _capi.load()
    indice_pairs = indice_pairs.contiguous()
    indice_num = indice_num.contiguous().to(indice_pairs.device)
    K, _, L = indice_pairs.shape
    nbr = torch.empty((K, max(num_rows, 1)), dtype=torch.int32, device=indice_pairs.device)
    with torch.cuda.device(indice_pairs.device):
        rc = lib.bevamd_spconv_nbr_from_pairs(_capi.ptr(indice_pairs), L, _capi.ptr(indice_num), K, int(bool(inverse)),
                                              _capi.ptr(nbr), nbr.shape[1], _capi.stream_ptr(indice_pairs.device))
    _capi.check(rc, "spconv_nbr_from_pairs")
    return nbr


def _subm_shortcut_(nbr, counts, rows):
    """The reference's SubM shortcut, in place and without a host sync (spconv_ops.h:272-276,300-303,309): the offset
    with the most pairs (first maximum) is run as the identity map over all rows.  It already is the identity for any
    undilated SubM rulebook, where this changes nothing; for a dilated one no offset is, and the reference's result is
    reproduced all the same.  `counts` [K] = pairs per offset."""
    K = nbr.shape[0]
    order = torch.arange(K - 1, -1, -1, device=nbr.device, dtype=torch.int64)
    kmax = (counts.to(torch.int64) * K + order).argmax().view(1)   # scores are unique: the FIRST maximum count wins
    ident = torch.full((1, nbr.shape[1]), -1, dtype=torch.int32, device=nbr.device)
    ident[0, :rows] = torch.arange(rows, dtype=torch.int32, device=nbr.device)
    nbr.index_copy_(0, kmax, ident)
    return nbr


def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse=False, subm=False):
    """ops.py:128-163 / spconv_ops.h:260-361 on reference-shaped pairs."""
    if filters.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise NotImplementedError
    nbr = _nbr_from_pairs(indice_pairs, indice_pair_num, num_activate_out, inverse)
    if subm:
        _subm_shortcut_(nbr, indice_pair_num.to(nbr.device), int(num_activate_out))
    return sparse_conv(features, filters, nbr, int(num_activate_out))


def fused_indice_conv(features, filters, bias, indice_pairs, indice_pair_num, num_activate_out, inverse=False, subm=False):
    """ops.py:166-182: `indice_conv` with the bias added by the same kernel."""
    nbr = _nbr_from_pairs(indice_pairs, indice_pair_num, num_activate_out, inverse)
    if subm:
        _subm_shortcut_(nbr, indice_pair_num.to(nbr.device), int(num_activate_out))
    return sparse_conv(features, filters, nbr, int(num_activate_out), bias=bias)


def indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num, inverse=False, subm=False):
    """ops.py:185-211 / spconv_ops.h:363-456 -> [in_grad, filter_grad]."""
    num_out = out_bp.shape[0]
    nbr = _nbr_from_pairs(indice_pairs, indice_pair_num, num_out, inverse)
    nbr_t = _nbr_from_pairs(indice_pairs, indice_pair_num, features.shape[0], not inverse)
    if subm:
        counts = indice_pair_num.to(nbr.device)
        _subm_shortcut_(nbr, counts, num_out)
        _subm_shortcut_(nbr_t, counts, features.shape[0])
    return list(sparse_conv_backward(features, filters, out_bp, nbr, nbr_t, features.shape[0]))


# --------------------------------------------------------------------------------------------
# sparse max pooling (ops.py:192-211 / pool_ops.h:25-97)
# --------------------------------------------------------------------------------------------
def sparse_maxpool(features, nbr, num_out):
    """out[o] = max(0, max_k features[nbr[k, o]]) in one launch (the reference's output starts from zeros)."""
    _require_cuda(features, "features")
    lib = _capi.load()
    if features.stride(1) != 1:
        features = features.contiguous()
    out = torch.empty((int(num_out), features.shape[1]), dtype=features.dtype, device=features.device)
    if num_out == 0 or features.shape[1] == 0:
        return out
    with torch.cuda.device(features.device):
        rc = lib.bevamd_spconv_maxpool_forward(_capi.ptr(features), _dtype_code(features), features.stride(0),
                                               _capi.ptr(nbr), nbr.stride(0), int(num_out), nbr.shape[0],
                                               features.shape[1], _capi.ptr(out), out.stride(0),
                                               _capi.stream_ptr(features.device))
    _capi.check(rc, "spconv_maxpool_forward")
    return out


def sparse_maxpool_backward(features, out_features, out_grad, nbr_t):
    """in_grad[i] = sum_k [out_features[nbr_t[k, i]] == features[i]] * out_grad[nbr_t[k, i]] (element-wise)."""
    _require_cuda(features, "features")
    lib = _capi.load()
    features = features.contiguous()
    out_features = out_features.contiguous().to(features.dtype)
    out_grad = out_grad.contiguous().to(features.dtype)
    in_grad = torch.zeros_like(features)
    if features.numel() == 0 or out_features.shape[0] == 0:
        return in_grad
    with torch.cuda.device(features.device):
        rc = lib.bevamd_spconv_maxpool_backward(_capi.ptr(features), _capi.ptr(out_features), _capi.ptr(out_grad),
                                                _dtype_code(features), _capi.ptr(nbr_t), nbr_t.stride(0),
                                                features.shape[0], nbr_t.shape[0], features.shape[1], _capi.ptr(in_grad),
                                                _capi.stream_ptr(features.device))
    _capi.check(rc, "spconv_maxpool_backward")
    return in_grad


def indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out):
    """ops.py:192-202 on reference-shaped pairs."""
    if features.dtype not in _DT:
        raise NotImplementedError
    nbr = _nbr_from_pairs(indice_pairs, indice_pair_num, num_activate_out, False)
    return sparse_maxpool(features, nbr, int(num_activate_out))


def indice_maxpool_backward(features, out_features, out_bp, indice_pairs, indice_pair_num):
    """ops.py:205-211."""
    if features.dtype not in _DT:
        raise NotImplementedError
    nbr_t = _nbr_from_pairs(indice_pairs, indice_pair_num, features.shape[0], True)
    return sparse_maxpool_backward(features, out_features, out_bp, nbr_t)


class _SparseConvExt:
    """Drop-in for the pybind module `sparse_conv_ext` (all.cc:21-51).  The 4D entry raises: no 4D rulebook here."""

    @staticmethod
    def get_indice_pairs_3d(indices, batch_size, out_shape, spatial_shape, ksize, stride, padding, dilation,
                            out_padding, subm, transpose):
        return list(get_indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding,
                                     bool(subm), bool(transpose), out_shape=out_shape))

    get_indice_pairs_2d = get_indice_pairs_3d   # the dimensionality is read off `indices`

    @staticmethod
    def get_indice_pairs_4d(*args, **kwargs):
        raise NotImplementedError("4D sparse convolution is not implemented (2D and 3D are)")

    @staticmethod
    def get_indice_pairs_grid_3d(indices, grid, batch_size, out_shape, spatial_shape, ksize, stride, padding, dilation,
                                 out_padding, subm, transpose):
        """all.cc:24-27: same result with a caller-owned dense grid, which this implementation has no use for."""
        return list(get_indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding,
                                     bool(subm), bool(transpose), out_shape=out_shape))

    get_indice_pairs_grid_2d = get_indice_pairs_grid_3d

    @staticmethod
    def indice_maxpool_fp32(features, indice_pairs, indice_num, num_act_out):
        return indice_maxpool(features.float(), indice_pairs, indice_num, num_act_out)

    @staticmethod
    def indice_maxpool_half(features, indice_pairs, indice_num, num_act_out):
        return indice_maxpool(features.half(), indice_pairs, indice_num, num_act_out)

    @staticmethod
    def indice_maxpool_backward_fp32(features, out_features, out_grad, indice_pairs, indice_num):
        return indice_maxpool_backward(features.float(), out_features.float(), out_grad.float(), indice_pairs, indice_num)

    @staticmethod
    def indice_maxpool_backward_half(features, out_features, out_grad, indice_pairs, indice_num):
        return indice_maxpool_backward(features.half(), out_features.half(), out_grad.half(), indice_pairs, indice_num)

    @staticmethod
    def indice_conv_fp32(features, filters, indice_pairs, indice_num, num_act_out, inverse, subm):
        return indice_conv(features.float(), filters.float(), indice_pairs, indice_num, num_act_out, bool(inverse), bool(subm))

    @staticmethod
    def indice_conv_half(features, filters, indice_pairs, indice_num, num_act_out, inverse, subm):
        return indice_conv(features.half(), filters.half(), indice_pairs, indice_num, num_act_out, bool(inverse), bool(subm))

    @staticmethod
    def fused_indice_conv_fp32(features, filters, bias, indice_pairs, indice_num, num_act_out, inverse, subm):
        """all.cc:32-37 / fused_spconv_ops.h: convolution + bias in one call (here: the bias rides in the epilogue)."""
        return fused_indice_conv(features.float(), filters.float(), bias.float(), indice_pairs, indice_num, num_act_out,
                                 bool(inverse), bool(subm))

    @staticmethod
    def fused_indice_conv_half(features, filters, bias, indice_pairs, indice_num, num_act_out, inverse, subm):
        return fused_indice_conv(features.half(), filters.half(), bias.half(), indice_pairs, indice_num, num_act_out,
                                 bool(inverse), bool(subm))

    @staticmethod
    def indice_conv_backward_fp32(features, filters, out_grad, indice_pairs, indice_num, inverse, subm):
        return indice_conv_backward(features.float(), filters.float(), out_grad.float(), indice_pairs, indice_num,
                                    bool(inverse), bool(subm))

    @staticmethod
    def indice_conv_backward_half(features, filters, out_grad, indice_pairs, indice_num, inverse, subm):
        return indice_conv_backward(features.half(), filters.half(), out_grad.half(), indice_pairs, indice_num,
                                    bool(inverse), bool(subm))


sparse_conv_ext = _SparseConvExt()
