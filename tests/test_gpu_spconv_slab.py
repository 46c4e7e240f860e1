"""GPU parity of the slab (staged-rows) submanifold convolution (csrc/spconv_slab.h, through the C ABI): against the CPU
oracle (oracle.indice_conv, float64 accumulate) and against the gather kernel it replaces (bevamd_spconv_conv_forward_tiled).

Bars: |err| <= tol * (1 + max|ref|), tol = 1e-3 (fp16) / 8e-3 (bf16); BIT-IDENTICAL to the gather kernel for the variants
that stage whole rows (same summation order), within 2 ulps (of the largest magnitude) of it for the ones that run the channels in 32- / 64-wide passes
(pass-major order: Cin = 128 always, Cin = 64 with 32-channel rows); every built variant; ranges longer than the
staging buffer (pieces), empty kernel lines, ragged last block, device-side row count, fused epilogue; block metadata
(hdr / slots, per kernel plane) equal to a numpy restatement."""
import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import spconv
from bevfusion_amd.spconv import ops as sops

from conftest import record_parity

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}   # 2 x the observed maxima 4.8e-4 / 3.7e-3 (profiles/r05_parity_observed.json; 2e-3 / 1.6e-2 until round 4)


def sorted_indices(rng, B, shape, n, dense_planes=()):
    """Active voxels of B samples in ascending linear index; `dense_planes`: x-planes filled completely (long ranges)."""
    vol = int(np.prod(shape))
    out = []
    for b in range(B):
        lin = set(rng.choice(vol, size=min(n, vol), replace=False).tolist())
        for x in dense_planes:
            lin.update(range(x * shape[1] * shape[2], (x + 1) * shape[1] * shape[2]))
        lin = np.array(sorted(lin))
        out.append(np.concatenate([np.full((len(lin), 1), b), np.stack(np.unravel_index(lin, shape), 1)], 1))
    return np.concatenate(out).astype(np.int32)


def make_case(rng, dev, c, dtype, B=2, shape=(24, 20, 9), n=1500, dense_planes=()):
    indices = sorted_indices(rng, B, shape, n, dense_planes)
    oi, opairs, onum, _ = oracle.get_indice_pairs(indices, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), [1, 1, 1], 1, order="cuda")
    assert np.array_equal(oi, indices)
    w = (rng.standard_normal((3, 3, 3, c, c)) / np.sqrt(c * 27 / 4)).astype(np.float32)
    f = torch.from_numpy(rng.standard_normal((indices.shape[0], c)).astype(np.float32)).to(dtype)
    w = torch.from_numpy(w).to(dtype)
    ref = oracle.indice_conv(f.float().numpy(), w.float().numpy(), opairs, onum, oi.shape[0])
    rb = spconv.build_rulebook(torch.from_numpy(indices).to(dev), B, list(shape), [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, True)
    return f.to(dev), w.to(dev), rb, ref


def run_slab(f, w, rb, variant=0, m_dev=None, **kw):
    c = w.shape[-1]
    meta = sops.slab_build(rb.nbr, rb.num_out, m_dev, sops.slab_block_rows(c, variant))
    out = sops.sparse_conv_slab(f, sops.make_filter_image(w), meta, rb.num_out, c, c, variant=variant, num_out_dev=m_dev, **kw)
    assert int(meta.status.item()) == 0
    return out, meta


def run_gather(f, w, rb, **kw):
    c = w.shape[-1]
    return sops.sparse_conv_tiled(f, sops.make_filter_image(w), rb.nbr, rb.num_out, 27, c, c, **kw)


def same_order(c, variant):
    """Variant code = KC*10000 + ...: rows staged whole (KC == Cin) keep the gather kernels' summation order."""
    if variant >= 4000000:                                           # filter-stationary kernels: v_mfma_32x32x16 sums 16 channels per step
        return False
    kc = (variant or sops.slab_variants(c)[0]) % 1000000 // 10000   # 1xxxxxx: the register-filter kernels, same KC field
    return kc == c


def assert_same(out, base, c, variant):
    if same_order(c, variant):
        assert torch.equal(out, base), f"variant {variant} is not bit-identical to the gather kernel"
    else:
        assert ulp_close(out, base, 2), f"variant {variant}"


def assert_close(out, ref, dtype):
    err = np.max(np.abs(out.float().cpu().numpy().astype(np.float64) - ref))
    record_parity(f"slab kernels, random cases vs float64 oracle ({str(dtype).split('.')[-1]}, C={out.shape[1]})", err / (1.0 + np.max(np.abs(ref))), TOL[dtype])
    assert err <= TOL[dtype] * (1.0 + np.max(np.abs(ref))), err


def ulp_close(a, b, ulps):
    """Within `ulps` units in the last place of the LARGEST magnitude (pass-major summation moves the fp32 sum by a few of
    its own ulps before the single 16-bit rounding; near zero that is many ulps of the small value)."""
    scale = float(torch.maximum(a.float().abs().max(), b.float().abs().max()))
    eps = 2.0 ** -10 if a.dtype == torch.float16 else 2.0 ** -7
    return float((a.float() - b.float()).abs().max()) <= ulps * eps * max(scale, 1.0)


def test_block_metadata_matches_numpy(dev):
    rng = np.random.default_rng(5)
    f, w, rb, _ = make_case(rng, dev, 32, torch.float16, n=900)
    nbr = rb.nbr.cpu().numpy()[:, : rb.num_out]
    for bm in (128, 256):
        meta = sops.slab_build(rb.nbr, rb.num_out, None, bm)
        nblk = (rb.num_out + bm - 1) // bm
        hdr = meta.hdr.cpu().numpy().view(np.int32).reshape(-1)[: nblk * 6].reshape(nblk, 3, 2)
        slots = meta.slots.cpu().numpy().view(np.uint16)[: nblk * 27 * bm].reshape(nblk, 27, bm)
        for b in range(nblk):
            rows = slice(b * bm, min((b + 1) * bm, rb.num_out))
            for j in range(3):                                     # kernel plane kx = j: its nine (ky, kz) taps share one range
                v = nbr[9 * j: 9 * j + 9, rows]
                if (v >= 0).any():
                    lo, hi = v[v >= 0].min(), v[v >= 0].max()
                    assert tuple(hdr[b, j]) == (lo, hi - lo + 1)
                else:
                    lo = 0
                    assert hdr[b, j, 1] == 0
                want = np.where(v >= 0, v - lo, 0xFFFF).astype(np.uint16)
                got = slots[b, 9 * j: 9 * j + 9, : want.shape[1]]
                assert np.array_equal(got, want)
                assert (slots[b, 9 * j: 9 * j + 9, want.shape[1]:] == 0xFFFF).all()      # rows past the live count
        # rows in linear-index order: a plane's range is about one block long
        assert np.median(hdr[:, :, 1][hdr[:, :, 1] > 0]) <= 1.5 * bm


def test_baked_block_metadata_matches_numpy(dev):
    """64-row blocks carry BAKED slots (spconv_slab_meta.h): entry = LDS byte offset of staged row s = (s + 1) * 64 with the bank
    swizzle bits (((s + 1) >> 2) & 3) << 4 folded in, 0 = no neighbour; a range of more than 1022 rows keeps raw slots and sets
    bit 30 of its row count."""
    rng = np.random.default_rng(6)
    for shape, dense, want_raw, n in (((24, 20, 9), (), False, 700), ((6, 60, 20), (2,), True, 100)):
        ind = sorted_indices(rng, 1, shape, n, dense)
        rb = spconv.build_rulebook(torch.from_numpy(ind).to(dev), 1, list(shape), [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, True)
        nbr = rb.nbr.cpu().numpy()[:, : rb.num_out]
        bm = 64
        meta = sops.slab_build(rb.nbr, rb.num_out, None, bm)
        nblk = (rb.num_out + bm - 1) // bm
        hdr = meta.hdr.cpu().numpy().view(np.int32).reshape(-1)[: nblk * 6].reshape(nblk, 3, 2)
        slots = meta.slots.cpu().numpy().view(np.uint16)[: nblk * 27 * bm].reshape(nblk, 27, bm)
        saw_raw = False
        for b in range(nblk):
            rows = slice(b * bm, min((b + 1) * bm, rb.num_out))
            for j in range(3):
                v = nbr[9 * j: 9 * j + 9, rows]
                got = slots[b, 9 * j: 9 * j + 9, : v.shape[1]]
                if not (v >= 0).any():
                    assert hdr[b, j, 1] == 0 and (got == 0).all()
                    continue
                lo, hi = v[v >= 0].min(), v[v >= 0].max()
                cnt = hi - lo + 1
                raw = (cnt + 1) * 64 > 0xFFFF
                saw_raw |= raw
                assert tuple(hdr[b, j]) == (lo, cnt | (0x40000000 if raw else 0))
                e = v - lo + 1
                want = np.where(v >= 0, v - lo, 0xFFFF) if raw else np.where(v >= 0, e * 64 | (((e >> 2) & 3) << 4), 0)
                assert np.array_equal(got, want.astype(np.uint16))
        assert saw_raw == want_raw


@pytest.mark.parametrize("c", [32])
def test_filter_stationary_kernels_on_raw_flagged_ranges(dev, c):
    """A completely filled 60 x 20 x-plane: the kernel planes that look into it span > 1022 rows, their slots stay raw (HDR_RAW) and
    the filter-stationary kernels take their general path for them, piece by piece."""
    rng = np.random.default_rng(77)
    f, w, rb, ref = make_case(rng, dev, c, torch.float16, B=1, shape=(6, 60, 20), n=100, dense_planes=(2,))
    base = run_gather(f, w, rb)
    fs = [v for v in sops.slab_variants(c) if v >= 4000000]
    assert fs
    for v in fs:
        out, meta = run_slab(f, w, rb, variant=v)
        nblk = (rb.num_out + 63) // 64
        cnt = meta.hdr.cpu().numpy().view(np.int32).reshape(-1)[: nblk * 6].reshape(nblk, 3, 2)[:, :, 1]
        assert (cnt & 0x40000000).any()
        assert_close(out, ref, torch.float16)
        assert_same(out, base, c, v)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("c", [32, 64, 128])
def test_every_variant_vs_oracle_and_gather(dev, c, dtype):
    rng = np.random.default_rng(c)
    f, w, rb, ref = make_case(rng, dev, c, dtype, n=1700)      # 3400 rows: ragged last block for 128- and 256-row blocks
    base = run_gather(f, w, rb)
    assert_close(base, ref, dtype)
    variants = sops.slab_variants(c)
    assert variants and sops.slab_block_rows(c, 0) == sops.slab_block_rows(c, variants[0])
    for v in variants:
        out, _ = run_slab(f, w, rb, variant=v)
        assert_close(out, ref, dtype)
        assert_same(out, base, c, v)
        again, _ = run_slab(f, w, rb, variant=v)
        assert torch.equal(out, again)                         # bit-reproducible


@pytest.mark.parametrize("c", [32, 64, 128])
def test_long_ranges_take_the_piece_loop(dev, c):
    """Two completely filled x-planes next to sparse ones: the kernel planes that look into them span up to Y*Z = 360 extra rows
    (> the 192 / 384-row staging buffers) and are processed in pieces."""
    rng = np.random.default_rng(100 + c)
    f, w, rb, ref = make_case(rng, dev, c, torch.float16, B=1, shape=(12, 40, 9), n=400, dense_planes=(4, 5))
    base = run_gather(f, w, rb)
    for v in sops.slab_variants(c):
        out, meta = run_slab(f, w, rb, variant=v)
        bm = meta.block_rows & 0xFFFF          # the upper half is the slot-format code (baked 128-byte rows)
        nblk = (rb.num_out + bm - 1) // bm
        cnt = meta.hdr.cpu().numpy().view(np.int32).reshape(-1)[: nblk * 6].reshape(nblk, 3, 2)[:, :, 1] & 0x3FFFFFFF
        assert cnt.max() > (1.5 * bm if bm > 64 else 200), "the case must exercise multi-piece ranges"
        assert_close(out, ref, torch.float16)
        assert_same(out, base, c, v)


@pytest.mark.parametrize("c", [32, 64])
def test_isolated_voxels_empty_lines_and_tiny_sets(dev, c):
    rng = np.random.default_rng(3)
    for n in (1, 7, 130):                                         # far apart: only the centre plane has rows
        shape = (40, 40, 20)
        lin = np.sort(rng.choice(np.arange(0, 40 * 40 * 20, 97), size=n, replace=False))
        ind = np.concatenate([np.zeros((n, 1), np.int64), np.stack(np.unravel_index(lin, shape), 1)], 1).astype(np.int32)
        oi, opairs, onum, _ = oracle.get_indice_pairs(ind, 1, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), [1, 1, 1], 1, order="cuda")
        w = torch.from_numpy((rng.standard_normal((3, 3, 3, c, c)) * 0.05).astype(np.float32)).half()
        f = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)).half()
        ref = oracle.indice_conv(f.float().numpy(), w.float().numpy(), opairs, onum, n)
        rb = spconv.build_rulebook(torch.from_numpy(ind).to(dev), 1, list(shape), [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, True)
        for v in sops.slab_variants(c):
            out, _ = run_slab(f.to(dev), w.to(dev), rb, variant=v)
            assert_close(out, ref, torch.float16)


def test_row_pitches_wider_than_the_channel_count(dev):
    """Features, output and residual as column slices of wider tensors (pitch 48 / 40 / 56 elements for 32 channels): the
    filter-stationary kernel stages and stores by PITCH (LDS-DMA source offsets, 16-byte row stores); bias + BatchNorm + residual +
    ReLU; bf16 as well."""
    rng = np.random.default_rng(21)
    c = 32
    for dtype in (torch.float16, torch.bfloat16):
        f, w, rb, _ = make_case(rng, dev, c, dtype, n=900)
        m = rb.num_out
        wide_f = torch.zeros((f.shape[0], 48), dtype=dtype, device=dev)
        wide_f[:, :c] = f
        wide_r = torch.randn((m, 40), device=dev).to(dtype)
        bias = torch.randn(c, device=dev).to(dtype)
        scale, shift = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        kw = dict(bias=bias, bn_scale=scale, bn_shift=shift, relu=True)
        base = run_gather(f, w, rb, residual=wide_r[:, :c].contiguous(), **kw)
        for v in [x for x in sops.slab_variants(c) if x >= 4000000]:
            wide_o = torch.full((m, 56), 3.0, dtype=dtype, device=dev)
            meta = sops.slab_build(rb.nbr, m, None, sops.slab_block_rows(c, v))
            out = sops.sparse_conv_slab(wide_f[:, :c], sops.make_filter_image(w), meta, m, c, c, variant=v, residual=wide_r[:, :c],
                                        out=wide_o[:, :c], **kw)
            assert out.data_ptr() == wide_o.data_ptr() and out.stride(0) == 56
            assert ulp_close(out, base, 2)
            assert torch.all(wide_o[:, c:] == 3.0)                    # nothing written past the 32 channels of a row


@pytest.mark.parametrize("c", [32, 128])
def test_epilogue_and_device_row_count(dev, c):
    rng = np.random.default_rng(11 + c)
    dtype = torch.float16
    f, w, rb, ref = make_case(rng, dev, c, dtype, n=1000)
    m = rb.num_out
    bias = torch.from_numpy(rng.standard_normal(c).astype(np.float32)).to(dev).to(dtype)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)).to(dev)
    shift = torch.from_numpy(rng.standard_normal(c).astype(np.float32)).to(dev)
    res = torch.from_numpy(rng.standard_normal((m, c)).astype(np.float32)).to(dev).to(dtype)
    kw = dict(bias=bias, bn_scale=scale, bn_shift=shift, residual=res, relu=True)
    base = run_gather(f, w, rb, **kw)
    for v in sops.slab_variants(c):
        out, _ = run_slab(f, w, rb, variant=v, **kw)
        assert_same(out, base, c, v)
    y = (torch.from_numpy(ref).to(dev).to(dtype).float() + bias.float()).to(dtype).float()
    y = (y * scale + shift).to(dtype).float()
    y = torch.relu((y + res.float()).to(dtype).float())
    assert float((out.float() - y).abs().max()) <= 8 * TOL[dtype] * (1 + float(y.abs().max()))   # several rounding points (the bar of rounds 1-4)
    # live row count on the device, launch bounded by the capacity: rows past it stay untouched
    live = m - 77
    m_dev = torch.tensor([live], dtype=torch.int32, device=dev)
    sentinel = torch.full((m, c), 7.0, dtype=dtype, device=dev)
    rb_live_nbr = rb.nbr.clone()
    rb_live_nbr[rb_live_nbr >= live] = -1                         # a table built for the live rows only
    want = sops.sparse_conv_tiled(f, sops.make_filter_image(w), rb_live_nbr, m, 27, c, c, num_out_dev=m_dev, out=sentinel.clone())
    for v in sops.slab_variants(c):
        meta = sops.slab_build(rb_live_nbr, m, m_dev, sops.slab_block_rows(c, v))
        got = sops.sparse_conv_slab(f, sops.make_filter_image(w), meta, m, c, c, num_out_dev=m_dev, out=sentinel.clone(), variant=v)
        assert torch.equal(got[live:], sentinel[live:])
        assert_same(got[:live], want[:live], c, v)


@pytest.mark.parametrize("c,rows", [(32, 420000), (64, 300000)])
def test_persistent_kernels_walk_several_blocks_per_workgroup(dev, c, rows):
    """More blocks than the persistent grid has workgroups (768-1024 on an MI355X): every workgroup crosses block boundaries —
    next block's header, rows and slots prefetched under the last plane, epilogue, accumulators reset — with a residual and a
    device row count a few blocks short of the capacity.  Bit-identical to the one-block kernels of the same shape."""
    rng = np.random.default_rng(c)
    shape = (300, 300, 24)
    ind = torch.from_numpy(sorted_indices(rng, 1, shape, rows)).to(dev)
    n = ind.shape[0]
    rb = spconv.build_rulebook(ind, 1, list(shape), [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, True)
    f = torch.randn(n, c, device=dev).half()
    res = torch.randn(n, c, device=dev).half()
    img = sops.make_filter_image((torch.randn(3, 3, 3, c, c, device=dev) / (27 * c) ** 0.5).half())
    live = n - 1000
    m_dev = torch.tensor([live], dtype=torch.int32, device=dev)
    nbr = rb.nbr.clone()
    nbr[nbr >= live] = -1
    persistent = [v for v in sops.slab_variants(c) if v >= 2000000]
    assert persistent
    for v in persistent:
        bm = sops.slab_block_rows(c, v)
        assert (live + bm - 1) // bm > 1100
        meta = sops.slab_build(nbr, n, m_dev, bm)
        kw = dict(residual=res, relu=True, num_out_dev=m_dev)
        got = sops.sparse_conv_slab(f, img, meta, n, c, c, variant=v, out=torch.zeros_like(f), **kw)
        if v >= 4000000:   # filter-stationary: 64-row blocks of its own; the twin is the default register-filter kernel
            tv = [x for x in sops.slab_variants(c) if 1000000 <= x < 2000000][0]
            tmeta = sops.slab_build(nbr, n, m_dev, sops.slab_block_rows(c, tv))
            twin = sops.sparse_conv_slab(f, img, tmeta, n, c, c, variant=tv, out=torch.zeros_like(f), **kw)
            assert ulp_close(got, twin, 2)
            twin = got
        else:
            twin = sops.sparse_conv_slab(f, img, meta, n, c, c, variant=v - 1000000, out=torch.zeros_like(f), **kw)
        assert torch.equal(got, twin)
        assert float(got[:live].float().abs().sum()) > 0 and float(got[live:].float().abs().sum()) == 0


@pytest.mark.parametrize("bm", [64, 128, 256])
def test_metadata_straight_from_the_index_equals_the_table_route(dev, bm):
    """bevamd_spconv_slab_build_from_index (27 lookups per row, no int32 table) writes the same (range, slots) as
    bevamd_spconv_neighbors + bevamd_spconv_slab_build: on a hash-indexed set in linear order, on the rank-indexed output of a
    strided convolution, with a device row count below the capacity, and on an empty set."""
    from bevfusion_amd.spconv import fused

    rng = np.random.default_rng(bm)
    B, shape = 3, (26, 22, 11)
    ind = torch.from_numpy(sorted_indices(rng, B, shape, 1800, dense_planes=(7,))).to(dev)
    n = ind.shape[0]

    def both(lvl):
        direct = sops.slab_build_from_index(lvl.indices, lvl.n_cap, lvl.n_dev, lvl.batch, lvl.shape, lvl.index_kind, lvl.index,
                                            lvl.index_n_cap, bm)
        table = sops.slab_build(lvl.subm_neighbors((3, 3, 3)), lvl.n_cap, lvl.n_dev, bm)
        torch.cuda.synchronize()
        m = int(lvl.n_dev.item()) if lvl.n_dev is not None else lvl.n_cap
        nblk = (m + bm - 1) // bm
        assert int(direct.status.item()) == 0 and int(table.status.item()) == 0
        assert torch.equal(direct.hdr[: nblk * 24], table.hdr[: nblk * 24])
        assert torch.equal(direct.slots[: nblk * 27 * bm * 2], table.slots[: nblk * 27 * bm * 2])
        return nblk

    top = fused.Level(ind, n, None, B, list(shape))
    top.ensure_index()
    assert top.index_kind == fused.INDEX_HASH and both(top) > 1
    low, _ = top.downsample((3, 3, 3), (2, 2, 2), (1, 1, 1))          # rank index, device count < capacity
    assert low.index_kind == fused.INDEX_RANK and int(low.n_dev.item()) < low.n_cap
    assert both(low) >= 1
    short = fused.Level(ind, n, torch.tensor([n // 3], dtype=torch.int32, device=dev), B, list(shape))
    short.ensure_index()
    both(short)
    none = fused.Level(ind, n, torch.zeros(1, dtype=torch.int32, device=dev), B, list(shape))
    none.ensure_index()
    both(none)
