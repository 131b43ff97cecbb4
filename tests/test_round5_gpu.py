"""Round-5 kernels through the C ABI against fp32 torch:
  * the small-M weight-streaming conv kernel (wstream.hip, plan tile 9): 3x3 stride-1 convs of the 8x8 / 16x16 levels incl. the
    nearest-x2 upsample gather, 1x1 GEMMs, dead K-slice waves, odd batch (unet.py:435-468, :496-500);
  * GroupNorm(+SiLU) written by the conv's own slab combine ("twin", reduce_twin_kernel) instead of a GroupNorm launch
    (unet.py:430-451, :472-481, :528-531).
Tolerances as tests/test_ops_gpu.py: PSNR >= 60 dB, max |err| <= 4e-3 * max|ref| + 1e-3 (fp16 I/O, fp32 accumulate)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import psnr
from python_hip_stable_diffusion import _lib

pytestmark = pytest.mark.gpu


def h16(a):
    return np.asarray(a, np.float32).astype(np.float16)


def close(got, ref, what, min_psnr=60.0, rel=4e-3):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), what
    p = psnr.compute_psnr(got, ref)
    err = np.abs(got - ref).max()
    bound = rel * np.abs(ref).max() + 1e-3
    assert p >= min_psnr and err <= bound, f"{what}: PSNR {p:.1f} dB, max|err| {err:.3e} (bound {bound:.3e})"


def conv_ref(x, w, bias, res, upsample):
    xt = torch.from_numpy(x.astype(np.float32))
    if upsample:
        xt = F.interpolate(xt, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xt, torch.from_numpy(w.astype(np.float32)), None if bias is None else torch.from_numpy(bias), padding=w.shape[2] // 2)
    if res is not None:
        y = y + torch.from_numpy(res.astype(np.float32))
    return y


WS_CASES = [  # (B, Cin, H, W, Cout, k, upsample)
    (2, 256, 8, 8, 128, 3, False),      # the 8x8 level: two images per 128-pixel block
    (2, 1280, 8, 8, 1280, 3, False),    # SD2.1-base mid-block resnet conv at full size (5 slabs at 8 waves)
    (1, 96, 8, 8, 32, 3, False),        # odd batch (second sub-tile masked), 3 input slices: dead waves
    (3, 64, 8, 8, 64, 3, False),        # two blocks, the second half empty
    (2, 128, 16, 16, 96, 3, False),     # the 16x16 level: 8-row blocks, four of them
    (2, 64, 8, 8, 64, 3, True),         # Upsample2D (unet.py:492-500): 8x8 source, 16x16 output
    (1, 320, 16, 16, 64, 3, False),     # 10 slices: two K splits at 8 waves with a ragged tail
    (2, 128, 8, 8, 96, 1, False),       # 1x1
    (2, 1280, 8, 8, 1280, 1, False),    # proj_in / to_out of the mid block
    (5, 64, 4, 4, 32, 1, False),        # 1x1: M = 80 < 128 (masked rows)
]


@pytest.mark.parametrize("case", WS_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("tile", [9, 49], ids=["8waves", "4waves"])
def test_weight_streaming_conv_matches_torch(case, tile):
    b, cin, h, w_, cout, k, up = case
    rs = np.random.RandomState(cin + cout + h + k)
    x = h16(rs.randn(b, cin, h, w_))
    w = h16(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k))
    bias = (0.1 * rs.randn(cout)).astype(np.float32)
    ho = h * (2 if up else 1)
    res = h16(rs.randn(b, cout, ho, ho * w_ // h))
    ref = conv_ref(x, w, bias, res, up).numpy()
    out, _ = _lib.conv2d(x, w, bias, res, upsample=up, tile=tile)
    close(out, ref, f"wstream conv {case} tile {tile}")
    again, _ = _lib.conv2d(x, w, bias, res, upsample=up, tile=tile, iters=3)
    assert np.array_equal(out, again), "replays must be bit-identical (fixed-order slab sums)"
    # the same conv on the tiled kernels of igemm.hip: two kernels, one answer
    base, _ = _lib.conv2d(x, w, bias, res, upsample=up)
    close(out, base.astype(np.float32), f"wstream vs tiled kernel {case}", min_psnr=65.0)


def test_weight_streaming_rejects_ineligible_shapes():
    rs = np.random.RandomState(0)
    x = h16(rs.randn(1, 64, 32, 32))
    w = h16(rs.randn(64, 64, 3, 3) * 0.05)
    with pytest.raises(ValueError):
        _lib.conv2d(x, w, tile=9)               # 32-pixel-wide image: not the 8x8 / 16x16 form
    x = h16(rs.randn(1, 64, 8, 8))
    with pytest.raises(ValueError):
        _lib.conv2d(x, w, stride=2, tile=9)     # stride 2


TWIN_CASES = [  # (B, Cin, HW, Cout, k, silu, tile)
    (2, 128, 8, 1280, 3, True, 0),      # 40-channel groups, default plan forced onto the slab path
    (2, 128, 8, 1280, 3, True, 9),      # ... behind the weight-streaming kernel
    (2, 1280, 8, 1280, 3, True, 9),     # full-size resnet conv of the 8x8 level
    (2, 64, 16, 640, 3, True, 0),       # 20-channel groups at 16x16
    (2, 64, 16, 1280, 3, False, 9),     # SpatialTransformer norm (no SiLU) at 16x16: 10 items per thread
    (1, 64, 16, 2560, 1, True, 0),      # 80-channel groups: 20 items per thread
    (2, 256, 8, 1280, 1, False, 0),     # proj_out-like 1x1 GEMM forced onto the slab path
    (2, 64, 4, 256, 3, True, 0),        # 8-channel groups, 4x4 image
]


@pytest.mark.parametrize("case", TWIN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_groupnorm_twin_of_the_slab_combine(case):
    b, cin, hw, cout, k, silu, tile = case
    rs = np.random.RandomState(cin + cout + hw)
    x = h16(rs.randn(b, cin, hw, hw))
    w = h16(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k))
    bias = (0.1 * rs.randn(cout)).astype(np.float32) + 0.3
    res = h16(rs.randn(b, cout, hw, hw) * 0.5 + 0.2)
    gw = (1.0 + 0.2 * rs.randn(cout)).astype(np.float32)
    gb = (0.2 * rs.randn(cout)).astype(np.float32)
    eps = 1e-5
    conv_t, out_t, _, _ = _lib.conv2d_groupnorm(x, w, gw, gb, bias, res, 32, eps, silu, tile=tile, producer_stats=2)
    conv_b, out_b, _, _ = _lib.conv2d_groupnorm(x, w, gw, gb, bias, res, 32, eps, silu, tile=0, producer_stats=0)
    ref_conv = conv_ref(x, w, bias, res, False)
    close(conv_t, ref_conv.numpy(), f"conv {case}")
    # against the GroupNorm of OUR conv output (isolates the twin from the conv's rounding) and against the launch it replaces
    z = F.group_norm(torch.from_numpy(conv_t.astype(np.float32)), 32, torch.from_numpy(gw), torch.from_numpy(gb), eps)
    z = (F.silu(z) if silu else z).numpy()
    close(out_t, z, f"GroupNorm twin {case}")
    close(out_t, out_b.astype(np.float32), f"twin vs GroupNorm launch {case}", min_psnr=58.0, rel=1e-2)
    zr = F.group_norm(ref_conv.half().float(), 32, torch.from_numpy(gw), torch.from_numpy(gb), eps)
    zr = (F.silu(zr) if silu else zr).numpy()
    close(out_t, zr, f"conv -> GroupNorm {case}", min_psnr=55.0, rel=1e-2)
    _, again, _, _ = _lib.conv2d_groupnorm(x, w, gw, gb, bias, res, 32, eps, silu, tile=tile, producer_stats=2, iters=3)
    assert np.array_equal(out_t, again)


def test_groupnorm_twin_rejects_what_it_cannot_hold():
    rs = np.random.RandomState(1)
    x = h16(rs.randn(1, 64, 32, 32))
    w = h16(rs.randn(64, 64, 3, 3) * 0.05)
    g = np.ones(64, np.float32)
    with pytest.raises(ValueError):   # 32x32 = 1024 pixels per (sample, group) slice: more than a workgroup keeps in registers
        _lib.conv2d_groupnorm(x, w, g, g, groups=32, producer_stats=2)


STATS_CASES = [  # (B, Cin, HW, Cout, k, tile, entries expected)
    (2, 256, 8, 1280, 3, 9, 64),     # weight-streaming conv: one entry per pixel at 8x8
    (2, 64, 16, 640, 3, 9, 256),     # 16x16: 256 entries, the fold's limit
    (1, 64, 8, 1280, 1, 9, 64),      # 1x1 through the streaming kernel
    (4, 64, 16, 640, 3, 9, 128),     # batch 4: two pixels per workgroup
]


@pytest.mark.skipif(os.environ.get("SD_TUNE") is None or os.environ.get("SD_REDUCE_STATS") != "1",
                    reason="the statistics pass of the slab combine is off by default (measured: the step loses 0.25 ms); "
                           "run with SD_TUNE=1 SD_REDUCE_STATS=1")
@pytest.mark.parametrize("case", STATS_CASES, ids=lambda c: "x".join(map(str, c)))
def test_groupnorm_statistics_from_the_slab_combine(case):
    """Split-K / weight-streaming producers (every conv of the 8x8 / 16x16 levels): the GroupNorm statistics come out of the slab
    combine (splitk_reduce_stats_kernel) and the GroupNorm runs its parallel apply pass (unet.py:470-489)."""
    b, cin, hw, cout, k, tile, want = case
    rs = np.random.RandomState(cin + cout + hw + b)
    x = h16(rs.randn(b, cin, hw, hw))
    w = h16(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k))
    bias = (0.1 * rs.randn(cout)).astype(np.float32) + 0.2
    res = h16(rs.randn(b, cout, hw, hw) * 0.5)
    gw = (1.0 + 0.2 * rs.randn(cout)).astype(np.float32)
    gb = (0.2 * rs.randn(cout)).astype(np.float32)
    conv_a, out_a, entries, _ = _lib.conv2d_groupnorm(x, w, gw, gb, bias, res, 32, 1e-5, True, tile=tile, producer_stats=1)
    conv_b, out_b, none, _ = _lib.conv2d_groupnorm(x, w, gw, gb, bias, res, 32, 1e-5, True, tile=tile, producer_stats=0)
    assert entries == want and none == 0, (entries, none)
    assert np.array_equal(conv_a, conv_b), "the statistics pass must not change the conv output"
    z = F.silu(F.group_norm(torch.from_numpy(conv_a.astype(np.float32)), 32, torch.from_numpy(gw), torch.from_numpy(gb), 1e-5)).numpy()
    close(out_a, z, f"GroupNorm from the slab combine's statistics {case}")
    close(out_b, z, f"GroupNorm with its own statistics {case}")
    _, again, _, _ = _lib.conv2d_groupnorm(x, w, gw, gb, bias, res, 32, 1e-5, True, tile=tile, producer_stats=1, iters=3)
    assert np.array_equal(out_a, again)


KG2_CASES = [  # (B, Cin, H, Cout, k, stride)
    (2, 1280, 16, 1280, 1, 1),    # to_out / proj_in / proj_out of the 16x16 level: 20 K steps -> 10 + 10
    (2, 1344, 16, 192, 1, 1),     # 21 steps: 11 + 10 (the shorter half pads with a barrier-only step)
    (2, 320, 32, 320, 1, 1),      # 5 steps: 3 + 2
    (1, 256, 8, 64, 1, 1),        # 4 steps, M = 64
    (2, 128, 32, 128, 3, 2),      # Downsample2D through the im2col kernel, 18 steps
    (2, 64, 8, 64, 1, 1),         # 1 step: below the threshold, the plain ring runs
]


@pytest.mark.parametrize("case", KG2_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("tile", [123, 133, 122, 124], ids=["64x64ring3", "64x64ring4", "128x64ring3", "64x128ring3"])
@pytest.mark.parametrize("splitk", [1, 2])
def test_in_workgroup_split_k_matches_torch(case, tile, splitk):
    """igemm_kernel KG = 2: two K groups of four waves per workgroup, accumulators handed over through LDS (unet.py:62-118 1x1
    projections, :503-510 stride-2 conv)."""
    b, cin, h, cout, k, stride = case
    rs = np.random.RandomState(cin + cout + h + tile)
    x = h16(rs.randn(b, cin, h, h))
    w = h16(rs.randn(cout, cin, k, k) / np.sqrt(cin * k * k))
    bias = (0.1 * rs.randn(cout)).astype(np.float32)
    ho = (h + 2 * (k // 2) - k) // stride + 1
    res = h16(rs.randn(b, cout, ho, ho))
    y = F.conv2d(torch.from_numpy(x.astype(np.float32)), torch.from_numpy(w.astype(np.float32)), torch.from_numpy(bias), stride=stride,
                 padding=k // 2) + torch.from_numpy(res.astype(np.float32))
    out, _ = _lib.conv2d(x, w, bias, res, stride=stride, tile=tile, splitk=splitk)
    close(out, y.numpy(), f"KG2 conv {case} tile {tile} splitk {splitk}")
    again, _ = _lib.conv2d(x, w, bias, res, stride=stride, tile=tile, splitk=splitk, iters=3)
    assert np.array_equal(out, again)


GNF_CASES = [  # (B, Cin, H, C, k, tile of the producer, group mean offset)
    (2, 320, 64, 320, 3, 0, 0.0),     # SD2.1-base level 0: resnet conv2 -> SpatialTransformer.norm -> proj_in at full size
    (2, 640, 32, 640, 3, 0, 0.0),     # level 1
    (2, 128, 16, 256, 1, 3, 0.0),     # 1x1 producer on the 64 x 64 tile, 8 channels per group
    (1, 64, 8, 64, 1, 3, 0.0),        # one sample, HW = 64: a single M tile, 2 channels per group
    (3, 192, 16, 192, 3, 0, 0.0),     # odd batch, 6 channels per group (groups straddle the fragments' 8-channel runs)
    (2, 128, 32, 128, 3, 0, 6.0),     # groups far from zero (|mean| ~ 6 sigma): the subtraction-first form keeps its digits
]


@pytest.mark.parametrize("case", GNF_CASES, ids=lambda c: "x".join(map(str, c)))
def test_groupnorm_folded_into_proj_in(case):
    """SpatialTransformer.norm -> proj_in as ONE launch (unet.py:528-531 eps 1e-6 no SiLU, :553-556): the GroupNorm statistics
    come from the producing conv's epilogue, the 1x1 GEMM applies mean / scale / shift to its activation fragments
    (gemm_pipe_kernel GNF).  Checked against fp32 torch on the conv output the kernel itself produced, and against the
    GroupNorm launch + plain GEMM pair."""
    b, cin, hw, c, k, tile, off = case
    rs = np.random.RandomState(cin + c + hw + b)
    x = h16(rs.randn(b, cin, hw, hw))
    w = h16(rs.randn(c, cin, k, k) / np.sqrt(cin * k * k))
    bias = (0.1 * rs.randn(c)).astype(np.float32) + off * np.repeat(rs.randn(32), c // 32).astype(np.float32)
    res = h16(rs.randn(b, c, hw, hw) * 0.5)
    gw = (1.0 + 0.2 * rs.randn(c)).astype(np.float32)
    gb = (0.2 * rs.randn(c)).astype(np.float32)
    pw = h16(rs.randn(c, c) / np.sqrt(c))
    pb = (0.1 * rs.randn(c)).astype(np.float32)
    conv_a, out_a, entries, _ = _lib.conv2d_groupnorm_proj(x, w, gw, gb, pw, pb, bias, res, 32, 1e-6, fold=True, tile=tile)
    conv_b, out_b, none, _ = _lib.conv2d_groupnorm_proj(x, w, gw, gb, pw, pb, bias, res, 32, 1e-6, fold=False, tile=tile)
    assert entries >= 1 and none == 0, (entries, none)
    assert np.array_equal(conv_a, conv_b)
    z = F.group_norm(torch.from_numpy(conv_a.astype(np.float32)), 32, torch.from_numpy(gw), torch.from_numpy(gb), 1e-6)
    y = F.conv2d(z, torch.from_numpy(pw.astype(np.float32)).reshape(c, c, 1, 1), torch.from_numpy(pb)).numpy()
    close(out_b, y, f"GroupNorm launch + GEMM {case}")
    close(out_a, y, f"GroupNorm folded into the GEMM {case}")
    _, again, _, _ = _lib.conv2d_groupnorm_proj(x, w, gw, gb, pw, pb, bias, res, 32, 1e-6, fold=True, tile=tile, iters=3)
    assert np.array_equal(out_a, again)


# ---------------------------------------------------------------- attention8 at real-checkpoint logit magnitudes
def _attn_truth(q32, k, v, h):
    from oracle import attention_ref
    return attention_ref.original(q32.astype(np.float32), k.astype(np.float32), v.astype(np.float32), h, 64)


def test_attention8_first_key_tile_far_below_zero():
    """a query whose first 64 keys all score below -128 in log2 units made the seed of the running max multiply
    the (zero) accumulators by exp2(+big) = inf; the seed now only sets the max.  First tile anti-aligned with the queries."""
    b, h, sq, sk = 1, 2, 128, 256
    rs = np.random.RandomState(11)
    q = rs.randn(b, h * 64, 1, sq).astype(np.float32) * 3.0
    k = rs.randn(b, h * 64, 1, sk).astype(np.float32)
    v = rs.randn(b, h * 64, 1, sk).astype(np.float32)
    qm = q.reshape(b, h, 64, sq).mean(axis=3, keepdims=True)            # a direction every query of the head shares
    q = (q.reshape(b, h, 64, sq) + 6.0 * np.sign(qm)).reshape(b, h * 64, 1, sq)
    kk = k.reshape(b, h, 64, sk)
    kk[:, :, :, :64] = -8.0 * np.sign(qm) + 0.1 * kk[:, :, :, :64]     # first key tile: q.k / 8 * log2(e) far below -128
    k = kk.reshape(b, h * 64, 1, sk)
    q, k, v = h16(q), h16(k), h16(v)
    s0 = np.einsum("bhcq,bhck->bhqk", q.astype(np.float32).reshape(b, h, 64, sq), k.astype(np.float32).reshape(b, h, 64, sk)[:, :, :, :64])
    assert (s0.max(axis=3) / 8 * 1.4427 < -128).any(), "the case must reach the overflow range"
    ref = _attn_truth(q, k, v, h)
    for impl in ("ORIGINAL", "SPLIT_EINSUM"):
        out, _ = _lib.attention(impl, q, k, v, h, 64)
        close(out, ref, f"attention8 {impl}, first tile below -128")


def test_attention8_query_scale_rounding_at_large_logits():
    """Logits of real-checkpoint magnitude (|q.k| / 8 up to ~60, competing keys, per-head outliers).  The reference rounds q to
    fp16 and scales the scores in fp32 (attention.py:49); the general kernel does the same; attention8 takes q with the scale in
    it.  Measured against the fp32 result on UNROUNDED queries: q scaled in fp32 and rounded once (variant 2: the UNet's path)
    must be as good as the reference's order; scaling the rounded q again in fp16 (variant 0: a caller's plain q) may cost at
    most 3 dB."""
    from oracle import psnr
    b, h, sq, sk = 1, 4, 256, 512
    rs = np.random.RandomState(5)
    q32 = rs.randn(b, h * 64, 1, sq).astype(np.float32)
    k32 = rs.randn(b, h * 64, 1, sk).astype(np.float32)
    v = h16(rs.randn(b, h * 64, 1, sk))
    gain = np.repeat(np.array([2.0, 4.0, 6.0, 7.5], np.float32), 64).reshape(1, h * 64, 1, 1)   # per-head logit scale
    q32 = q32 * gain
    k32[:, :, :, 17] *= 3.0                                                                        # an outlier key
    k = h16(k32)
    c = np.float32(1.4426950408889634 / 8.0)
    truth = _attn_truth(q32, k, v, h)
    s = np.einsum("bhcq,bhck->bhqk", q32.reshape(b, h, 64, sq), k.astype(np.float32).reshape(b, h, 64, sk)) / 8
    assert 40 < np.abs(s).max() < 400, np.abs(s).max()
    got = {
        "general (fp16 q, fp32 scale)": _lib.attention("ORIGINAL", h16(q32), k, v, h, 64, variant=1)[0],
        "attention8, plain q": _lib.attention("ORIGINAL", h16(q32), k, v, h, 64, variant=0)[0],
        "attention8, pre-scaled q": _lib.attention("ORIGINAL", h16(q32 * c), k, v, h, 64, variant=2)[0],
    }
    db = {n: psnr.compute_psnr(np.asarray(o, np.float64), truth.astype(np.float64)) for n, o in got.items()}
    print("PSNR vs fp32 attention on unrounded q:", {n: round(x, 2) for n, x in db.items()})
    base = db["general (fp16 q, fp32 scale)"]
    assert base > 35, db
    assert db["attention8, pre-scaled q"] >= base - 1.0, db
    assert db["attention8, plain q"] >= base - 3.0, db


SMALL_N_CASES = [  # (B, Cin, H, W, Cout, upsample)
    (2, 320, 64, 64, 4, False),     # the UNet's conv_out at full size: 64 lanes, one 4-pixel group per wave
    (1, 128, 64, 96, 3, False),     # the VAE decoder's conv_out shape class: 16 lanes per group, four groups per wave, N = 3
    (1, 256, 8, 12, 4, False),      # 32 lanes per group, rows shorter than a wave's pixels, ragged last wave
    (1, 64, 5, 4, 2, False),        # 8 chunks on a 16-lane group (dead lanes), odd height
    (1, 640, 8, 8, 4, False),       # 80 chunks: above the row kernel's range -> the one-wave-per-pixel kernel
    (1, 64, 6, 6, 4, False),        # W % 4 != 0 -> the one-wave-per-pixel kernel
    (2, 64, 8, 8, 8, False),        # N = 8 (VAE encoder conv_out): the one-wave-per-pixel kernel
]


@pytest.mark.parametrize("case", SMALL_N_CASES, ids=lambda c: "x".join(map(str, c)))
def test_small_n_conv_matches_torch(case):
    """conv_out (unet.py:1046: 320 -> 4, 3x3) and the VAE decoder's / encoder's last conv on the small-N kernels: every border
    pixel, dead lanes, both kernels' shape ranges."""
    b, cin, h, w_, cout, up = case
    rs = np.random.RandomState(cin + cout + h + w_)
    x = h16(rs.randn(b, cin, h, w_))
    w = h16(rs.randn(cout, cin, 3, 3) / np.sqrt(cin * 9))
    bias = (0.1 * rs.randn(cout)).astype(np.float32)
    y = conv_ref(x, w, bias, None, up).numpy()
    out, _ = _lib.conv2d(x, w, bias)
    close(out, y, f"small-N conv {case}")
    ref_kernel, _ = _lib.conv2d(x, w, bias, force_generic=True)
    close(ref_kernel, y, f"direct conv {case}")
    again, _ = _lib.conv2d(x, w, bias, iters=3)
    assert np.array_equal(out, again)


# ---------------------------------------------------------------- VAE encoder / 64x64 decoder against the reference's own blocks
@pytest.mark.parametrize("name", ["mini", "sd"])
def test_vae_encoder_matches_the_golden_of_the_reference_blocks(name):
    """tests/golden/vae_encoder_*_golden.npz: quant_conv(encoder(x)) (torch2coreml.py:739-749) wired from the reference's
    ResnetBlock2D(temb_channels=None, eps=1e-6) and single-head attention.original plus F.pad((0,1,0,1)) / F.conv2d(stride=2)
    (oracle/pin_round5.py): arithmetic pinned by the reference's blocks, topology restated."""
    from conftest import load_golden
    from oracle import vae_ref, weights
    from oracle.pin_round5 import encoder_image
    from python_hip_stable_diffusion import HipVaeEncoder
    g = load_golden(f"vae_encoder_{name}_golden.npz")
    cfg = vae_ref.VAE_CONFIGS[name]
    hw = int(g["hw"])
    sd16 = weights.make_state_dict(vae_ref.vae_encoder_param_shapes(cfg), seed=int(g["seed"]), dtype=np.float16, gain=float(g["gain"]))
    enc = HipVaeEncoder(cfg, sd16, batch=1, height=hw, width=hw)
    out = enc(x=encoder_image(hw, int(g["x_seed"])))["latent"]
    p = psnr.compute_psnr(out, g["moments"])
    assert out.shape == g["moments"].shape and p >= 67.0, f"VAE encoder {name}: PSNR {p:.1f} dB vs the reference-block golden"
    enc.close()


def test_vae_decoder_matches_the_reference_block_golden_at_the_benchmarked_size():
    """the decode bench.py times: SD-sized decoder, 64x64 latents -> 512x512 image (oracle/pin_round5.py --vae64)"""
    from conftest import load_golden
    from oracle import vae_ref, weights
    from python_hip_stable_diffusion import HipVaeDecoder
    g = load_golden("vae_decoder_sd64_golden.npz")
    cfg = vae_ref.VAE_CONFIGS["sd"]
    hw = int(g["hw"])
    sd16 = weights.make_state_dict(vae_ref.vae_decoder_param_shapes(cfg), seed=int(g["seed"]), dtype=np.float16)
    vae = HipVaeDecoder(cfg, sd16, batch=1, latent_height=hw, latent_width=hw)
    z = weights.seeded_normal((1, cfg["latent_channels"], hw, hw), int(g["z_seed"])).astype(np.float16)
    out = vae(z=z)["image"]
    ref = g["image"].astype(np.float32)
    p = psnr.compute_psnr(out, ref)
    assert out.shape == ref.shape == (1, 3, 512, 512) and p >= 60.0, f"VAE decoder @64x64: PSNR {p:.1f} dB vs the reference-block golden"
    vae.close()


# ---------------------------------------------------------------- norm1 and conv_shortcut in one launch (gn_*_side_kernel)
SIDE_CASES = [  # (B, C0, C1, H, N): the resnets with a channel change, unet.py:470-489
    (2, 1280, 1280, 16, 1280),   # up_blocks.1.resnets.0 at full size: single-launch GroupNorm (64 workgroups) + 160 GEMM tiles
    (2, 1280, 640, 16, 1280),    # 1920 channels: 60 per group (vector width 4)
    (2, 640, 320, 32, 640),      # up_blocks.2.resnets.2: partial + apply pair, the GEMM rides in the apply launch; 30 per group (width 2)
    (2, 320, 320, 64, 320),      # up_blocks.3.resnets.1 at full size
    (2, 640, 0, 16, 1280),       # down_blocks.2.resnets.0: single source
    (1, 128, 64, 16, 64),        # one sample, M = 256: the smallest side GEMM
    (3, 64, 64, 24, 192),        # odd batch, ragged M tiles (M = 1728), groups of 4
]


@pytest.mark.parametrize("case", SIDE_CASES, ids=lambda c: "x".join(map(str, c)))
def test_groupnorm_and_shortcut_gemm_in_one_launch(case):
    """The GroupNorm's blocks and the shortcut GEMM's tiles in one grid: both results bit-identical to the two separate launches of
    the same kernels' code paths where they exist, and equal to torch (GroupNorm(32) + SiLU over the concat; 1x1 conv over the concat)."""
    b, c0, c1, hw, n = case
    rs = np.random.RandomState(c0 + c1 + hw + n)
    x0 = h16(rs.randn(b, c0, hw, hw))
    x1 = h16(rs.randn(b, c1, hw, hw) * 1.5 + 0.3) if c1 else None
    c = c0 + c1
    gw = (1.0 + 0.2 * rs.randn(c)).astype(np.float32)
    gb = (0.2 * rs.randn(c)).astype(np.float32)
    w = h16(rs.randn(n, c) / np.sqrt(c))
    bias = (0.1 * rs.randn(n)).astype(np.float32)
    gn_a, sc_a, _ = _lib.groupnorm_shortcut(x0, x1, gw, gb, w, bias, side=True)
    gn_b, sc_b, _ = _lib.groupnorm_shortcut(x0, x1, gw, gb, w, bias, side=False)
    xc = np.concatenate([x0] + ([x1] if c1 else []), axis=1).astype(np.float32)
    z = F.silu(F.group_norm(torch.from_numpy(xc), 32, torch.from_numpy(gw), torch.from_numpy(gb), 1e-5)).numpy()
    y = F.conv2d(torch.from_numpy(xc), torch.from_numpy(w.astype(np.float32)).reshape(n, c, 1, 1), torch.from_numpy(bias)).numpy()
    close(gn_a, z, f"GroupNorm beside the GEMM {case}")
    close(sc_a, y, f"shortcut GEMM beside the GroupNorm {case}")
    close(gn_b, z, f"GroupNorm alone {case}")
    close(sc_b, y, f"shortcut GEMM alone {case}")
    if hw * hw <= 256 or hw * hw > 1024:   # same GroupNorm kernel body in both modes (32x32: the stand-alone launch is the 1024-thread one)
        assert np.array_equal(gn_a, gn_b)
    again = _lib.groupnorm_shortcut(x0, x1, gw, gb, w, bias, side=True, iters=3)
    assert np.array_equal(gn_a, again[0]) and np.array_equal(sc_a, again[1])


def test_shortcut_gemm_too_small_for_the_side_launch_is_refused():
    """M = 128 (the 8x8 level): the shortcut stays a launch of its own (split-K weight stream); the op-level entry says so"""
    rs = np.random.RandomState(1)
    x0 = h16(rs.randn(2, 128, 8, 8))
    g = np.ones(128, np.float32)
    with pytest.raises(NotImplementedError):
        _lib.groupnorm_shortcut(x0, None, g, g, h16(rs.randn(64, 128)), side=True)


# ---------------------------------------------------------------- GroupNorm(+SiLU) in the 3x3 conv's halo loader
GNL_CASES = [  # (B, Cin, H, W, C, producer k, N2, staging of the 3x3 conv, silu, group mean offset)
    (2, 320, 64, 64, 320, 3, 320, 3, True, 0.0),    # SD2.1-base level 0: conv1 -> norm2 -> SiLU -> conv2 at full size, 4-stage ring, two workgroups per CU
    (2, 640, 32, 32, 640, 3, 640, 3, True, 0.0),    # level 1 (table 3.8 KB: one workgroup per CU)
    (2, 320, 64, 64, 320, 1, 320, 2, True, 0.0),    # 3-stage ring; 1x1 producer (attention to_out -> next resnet's norm1 -> conv1)
    (1, 64, 16, 16, 64, 1, 128, 3, True, 0.0),      # ONE channel chunk: the prologue's transform only, 2 channels per group
    (2, 128, 16, 16, 128, 3, 64, 3, True, 0.0),     # two chunks
    (3, 192, 16, 16, 192, 3, 192, 2, True, 0.0),    # three chunks, odd batch, 6 channels per group (groups straddle the 8-channel runs)
    (2, 128, 24, 24, 128, 3, 128, 3, True, 0.0),    # 24 x 24: ragged 8 x 16 tiles - halo pixels beyond the image on every side must stay zero
    (2, 128, 32, 32, 128, 3, 128, 3, True, 6.0),    # groups far from zero (|mean| ~ 6 sigma): the subtraction-first form keeps its digits
    (2, 256, 8, 16, 256, 3, 256, 3, True, 0.0),     # a single tile row per image, four chunks
]


@pytest.mark.parametrize("case", GNL_CASES, ids=lambda c: "x".join(map(str, c)))
def test_groupnorm_in_the_conv_halo_loader(case):
    """ResnetBlock2D norm -> SiLU -> 3x3 conv as ONE launch (unet.py:472-481): the GroupNorm statistics come from the producing
    conv's epilogue, every wave of conv3x3_halo_ks_kernel<D, 0, GNL> normalises the halo pieces it fetched, in LDS, and the conv's
    zero padding applies to the NORMALISED tensor.  Checked against fp32 torch on the producer output the kernel itself wrote, and
    against the GroupNorm launch + plain conv pair."""
    b, cin, hh, ww, c, k, n2, st2, silu, off = case
    rs = np.random.RandomState(cin + c + hh + 3 * ww + b + n2)
    x = h16(rs.randn(b, cin, hh, ww))
    w = h16(rs.randn(c, cin, k, k) / np.sqrt(cin * k * k))
    bias = (0.1 * rs.randn(c)).astype(np.float32) + off * np.repeat(rs.randn(32), c // 32).astype(np.float32)
    gw = (1.0 + 0.2 * rs.randn(c)).astype(np.float32)
    gb = (0.2 * rs.randn(c)).astype(np.float32)
    w2 = h16(rs.randn(n2, c, 3, 3) / np.sqrt(c * 9))
    b2 = (0.1 * rs.randn(n2)).astype(np.float32)
    res2 = h16(rs.randn(b, n2, hh, ww) * 0.5)
    kw = dict(bias2=b2, res2=res2, bias=bias, groups=32, eps=1e-5, silu=silu, staging2=st2)
    conv_a, out_a, entries, _ = _lib.conv2d_groupnorm_conv3x3(x, w, gw, gb, w2, fold=True, **kw)
    conv_b, out_b, none, _ = _lib.conv2d_groupnorm_conv3x3(x, w, gw, gb, w2, fold=False, **kw)
    assert entries >= 1 and none == 0, (entries, none)
    assert np.array_equal(conv_a, conv_b)
    z = F.group_norm(torch.from_numpy(conv_a.astype(np.float32)), 32, torch.from_numpy(gw), torch.from_numpy(gb), 1e-5)
    if silu:
        z = F.silu(z)
    y = (F.conv2d(z, torch.from_numpy(w2.astype(np.float32)), torch.from_numpy(b2), padding=1) + torch.from_numpy(res2.astype(np.float32))).numpy()
    close(out_b, y, f"GroupNorm launch + conv {case}")
    close(out_a, y, f"GroupNorm in the halo loader {case}")
    _, again, _, _ = _lib.conv2d_groupnorm_conv3x3(x, w, gw, gb, w2, fold=True, iters=3, **kw)
    assert np.array_equal(out_a, again)


def test_groupnorm_in_the_loader_refuses_what_it_cannot_take():
    rs = np.random.RandomState(5)
    x = h16(rs.randn(1, 64, 4, 4))               # 4 x 4 image: below the halo kernel's 8 x 8
    w = h16(rs.randn(64, 64, 1, 1) / 8)
    w2 = h16(rs.randn(64, 64, 3, 3) / 24)
    g = np.ones(64, np.float32)
    with pytest.raises(NotImplementedError):
        _lib.conv2d_groupnorm_conv3x3(x, w, g, g, w2, fold=True)
    x = h16(rs.randn(1, 64, 16, 16))
    with pytest.raises(NotImplementedError):     # no SiLU: the loader has no such form (no GroupNorm of the graph needs it)
        _lib.conv2d_groupnorm_conv3x3(x, w, g, g, w2, silu=False, fold=True)


# ---------------------------------------------------------------- the cross-attention branch as ONE launch
def xblock_ref(x, lw, lb, wq, k, v, wo, bo, heads, eps):
    """fp32: x + to_out(attention(to_q(LayerNormANE(x)), k, v)) + bo (unet.py:586-591, :87-118)."""
    from oracle import attention_ref
    xt = torch.from_numpy(x.astype(np.float32))                        # (B, C, 1, S)
    mu = xt.mean(dim=1, keepdim=True)
    var = ((xt - mu) ** 2).mean(dim=1, keepdim=True)
    n = (xt - mu) * torch.rsqrt(var + eps) * torch.from_numpy(lw).view(1, -1, 1, 1) + torch.from_numpy(lb).view(1, -1, 1, 1)
    q = F.conv2d(n, torch.from_numpy(wq.astype(np.float32))[:, :, None, None])
    a2 = attention_ref.IMPLS["SPLIT_EINSUM"](q.numpy(), k.astype(np.float32), v.astype(np.float32), heads, 64)
    o = F.conv2d(torch.from_numpy(np.asarray(a2, np.float32)), torch.from_numpy(wo.astype(np.float32))[:, :, None, None], torch.from_numpy(bo))
    return (xt + o).numpy()


XBLOCK_CASES = [  # (B, heads, Sq, Sk)
    (2, 5, 4096, 77),     # SD2.1-base level 0 at full size: 256 workgroups of five waves
    (2, 10, 1024, 77),    # level 1: ten waves, V^T fragments behind the softmax
    (1, 5, 32, 77),       # a single workgroup
    (3, 5, 96, 33),       # odd batch, fewer keys: two of the three key tiles partly / fully masked
    (1, 10, 64, 96),      # the key capacity
    (2, 5, 576, 1),       # one key: softmax of a single score; SDXL's 24 x 24 token count
]


@pytest.mark.parametrize("case", XBLOCK_CASES, ids=lambda c: "x".join(map(str, c)))
def test_cross_attention_block_in_one_launch(case):
    """norm2 -> to_q -> softmax(q k^T) v -> to_out -> + residual as ONE launch (xattn_out.hip: 32 tokens x all heads per workgroup,
    weights global -> VGPR in fragment order) against fp32 torch and against the two-launch path it replaces."""
    b, heads, sq, sk = case
    c = heads * 64
    rs = np.random.RandomState(sq + 3 * sk + heads + b)
    x = h16(rs.randn(b, c, 1, sq) * 1.5 + 0.3)
    lw = (1.0 + 0.2 * rs.randn(c)).astype(np.float32)
    lb = (0.2 * rs.randn(c)).astype(np.float32)
    wq = h16(rs.randn(c, c) / np.sqrt(c))
    k = h16(rs.randn(b, c, 1, sk))
    v = h16(rs.randn(b, c, 1, sk))
    wo = h16(rs.randn(c, c) / np.sqrt(c))
    bo = (0.1 * rs.randn(c)).astype(np.float32)
    ref = xblock_ref(x, lw, lb, wq, k, v, wo, bo, heads, 1e-5)
    two, _ = _lib.cross_attention_block(x, lw, lb, wq, k, v, wo, bo, heads, fused=False)
    one, _ = _lib.cross_attention_block(x, lw, lb, wq, k, v, wo, bo, heads, fused=True)
    close(two, ref, f"cross-attention + to_out, two launches {case}")
    close(one, ref, f"cross-attention + to_out, one launch {case}")
    again, _ = _lib.cross_attention_block(x, lw, lb, wq, k, v, wo, bo, heads, fused=True, iters=3)
    assert np.array_equal(one, again)


XPRE_CASES = [(2, 5, 4096, 77), (1, 5, 32, 77), (3, 5, 96, 40)]   # (B, heads, Sq, Sk)


@pytest.mark.parametrize("case", XPRE_CASES, ids=lambda c: "x".join(map(str, c)))
def test_self_attention_out_projection_and_cross_attention_block_in_one_launch(case):
    """attn1.to_out + residual -> norm2 -> to_q -> attention -> attn2.to_out + residual (unet.py:588-590) as ONE launch: h1 never
    goes to HBM.  Against fp32 torch (h1 rounded to fp16 where the separate launch would store it) and the three-launch path."""
    b, heads, sq, sk = case
    c = heads * 64
    rs = np.random.RandomState(7 * sq + sk + heads + b)
    h0 = h16(rs.randn(b, c, 1, sq) * 1.5 + 0.3)
    a1 = h16(rs.randn(b, c, 1, sq))
    wo1 = h16(rs.randn(c, c) / np.sqrt(c))
    bo1 = (0.1 * rs.randn(c)).astype(np.float32)
    lw = (1.0 + 0.2 * rs.randn(c)).astype(np.float32)
    lb = (0.2 * rs.randn(c)).astype(np.float32)
    wq = h16(rs.randn(c, c) / np.sqrt(c))
    k = h16(rs.randn(b, c, 1, sk))
    v = h16(rs.randn(b, c, 1, sk))
    wo = h16(rs.randn(c, c) / np.sqrt(c))
    bo = (0.1 * rs.randn(c)).astype(np.float32)
    h1 = torch.from_numpy(h0.astype(np.float32)) + F.conv2d(torch.from_numpy(a1.astype(np.float32)),
                                                            torch.from_numpy(wo1.astype(np.float32))[:, :, None, None], torch.from_numpy(bo1))
    ref = xblock_ref(h16(h1.numpy()), lw, lb, wq, k, v, wo, bo, heads, 1e-5)
    kw = dict(a1=a1, wo1=wo1, bo1=bo1)
    three, _ = _lib.cross_attention_block(h0, lw, lb, wq, k, v, wo, bo, heads, fused=False, **kw)
    one, _ = _lib.cross_attention_block(h0, lw, lb, wq, k, v, wo, bo, heads, fused=True, **kw)
    close(three, ref, f"to_out + cross-attention + to_out, three launches {case}")
    close(one, ref, f"to_out + cross-attention + to_out, one launch {case}")
    again, _ = _lib.cross_attention_block(h0, lw, lb, wq, k, v, wo, bo, heads, fused=True, iters=3, **kw)
    assert np.array_equal(one, again)


def test_cross_attention_block_refuses_other_head_counts():
    rs = np.random.RandomState(3)
    c = 20 * 64
    z = np.zeros((1, c, 1, 64), np.float16)
    kk = np.zeros((1, c, 1, 77), np.float16)
    w = np.zeros((c, c), np.float16)
    g = np.ones(c, np.float32)
    with pytest.raises(NotImplementedError):     # 20 heads: a workgroup would need 20 waves
        _lib.cross_attention_block(z, g, g, w, kk, kk, w, g, 20, fused=True)


# ---------------------------------------------------------------- the tail of a SpatialTransformer as ONE launch
FFP_CASES = [(2, 4096), (1, 32), (3, 96), (2, 576)]   # (B, S) at C = 320


@pytest.mark.parametrize("case", FFP_CASES, ids=lambda c: "x".join(map(str, c)))
def test_ffn_out_and_proj_out_in_one_launch(case):
    """ff.net.2 + residual -> proj_out + residual (unet.py:591, :561-563) as ONE launch (xattn_out.hip ffn_proj_kernel) against fp32
    torch (the intermediate rounded to fp16 where the separate launch would store it) and against the two GEMM launches; the
    GroupNorm statistics it leaves for the next resnet's norm1 against the sums of its own fp16 output."""
    b, s_ = case
    c = 320
    rs = np.random.RandomState(11 * s_ + b)
    g = h16(rs.randn(b, 4 * c, 1, s_))
    w1 = h16(rs.randn(c, 4 * c) / np.sqrt(4 * c))
    b1 = (0.1 * rs.randn(c)).astype(np.float32)
    res1 = h16(rs.randn(b, c, 1, s_))
    w2 = h16(rs.randn(c, c) / np.sqrt(c))
    b2 = (0.1 * rs.randn(c)).astype(np.float32)
    res2 = h16(rs.randn(b, c, 1, s_))
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    h3 = t(res1) + F.conv2d(t(g), t(w1)[:, :, None, None], t(b1))
    ref = (t(res2) + F.conv2d(t(h16(h3.numpy())), t(w2)[:, :, None, None], t(b2))).numpy()
    two, sums_two, _ = _lib.ffn_out_proj(g, w1, b1, res1, w2, b2, res2, groups=32, fused=False)
    one, sums, _ = _lib.ffn_out_proj(g, w1, b1, res1, w2, b2, res2, groups=32, fused=True)
    close(two, ref, f"ff.net.2 + proj_out, two launches {case}")
    close(one, ref, f"ff.net.2 + proj_out, one launch {case}")
    o32 = one.astype(np.float64).reshape(b, 32, c // 32, s_)
    want = np.stack([o32.sum(axis=(2, 3)), (o32 ** 2).sum(axis=(2, 3))], axis=-1)
    assert np.isfinite(sums).all()
    assert np.abs(sums - want).max() <= 2e-4 * np.abs(want).max() + 1e-3, np.abs(sums - want).max()
    again, sums2, _ = _lib.ffn_out_proj(g, w1, b1, res1, w2, b2, res2, groups=32, fused=True, iters=3)
    assert np.array_equal(one, again) and np.array_equal(sums, sums2)


def test_ffn_out_proj_refuses_other_widths():
    z = lambda *sh: np.zeros(sh, np.float16)
    with pytest.raises(NotImplementedError):
        _lib.ffn_out_proj(z(1, 2560, 1, 32), z(640, 2560), np.zeros(640, np.float32), z(1, 640, 1, 32), z(640, 640), np.zeros(640, np.float32),
                          z(1, 640, 1, 32), fused=True)
