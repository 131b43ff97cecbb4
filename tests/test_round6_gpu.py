"""Round-6 kernels through the C ABI against fp32 torch:
  * the weight-stationary GEGLU projection of the 320-channel level (wsgemm.hip, plan tile 10): norm3 folded into
    ff.net.0.proj, value * gelu(gate) (unet.py:583-591, :609-617) - against the fp32 reference AND against the tiled GEMM
    kernels it replaces, ragged row counts, one / odd / even numbers of row tiles per workgroup, bit-reproducibility.
Tolerances as tests/test_ops_gpu.py: PSNR >= 60 dB, max |err| <= 4e-3 * max|ref| + 1e-3 (fp16 I/O, fp32 accumulate)."""
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


def geglu_ln_ref(x, w, bias, ln_w, ln_b, eps=1e-5):
    xt = torch.from_numpy(x.astype(np.float32))
    if ln_w is not None:
        xt = F.layer_norm(xt, (xt.shape[1],), torch.from_numpy(ln_w), torch.from_numpy(ln_b), eps)   # unet.py:583-591 norm3
    h = xt @ torch.from_numpy(w.astype(np.float32)).T
    if bias is not None:
        h = h + torch.from_numpy(bias)
    val, gate = h.chunk(2, dim=1)                                                                   # unet.py:616-617
    return (val * F.gelu(gate)).numpy()


def make_case(m, n2, seed, ln=True, offset=0.0):
    rs = np.random.RandomState(seed)
    c = 320
    x = h16(rs.randn(m, c) * (1.0 + rs.rand(m, 1)) + offset)        # rows of different scale: the fold's statistics matter
    w = h16(rs.randn(n2, c) / np.sqrt(c))
    bias = (0.1 * rs.randn(n2)).astype(np.float32)
    ln_w = (1.0 + 0.2 * rs.randn(c)).astype(np.float32) if ln else None
    ln_b = (0.1 * rs.randn(c)).astype(np.float32) if ln else None
    return x, w, bias, ln_w, ln_b


WSG_CASES = [  # (M, N2): row tiles of 64 over min(tiles, 256 / (N2 / 256)) workers
    (8192, 2560),    # SD2.1-base / SD1.5 ff.net.0.proj of the 64x64 level at CFG batch 2: 128 tiles on 25 workers -> 6 | 5 per worker
    (2048, 2560),    # 32 tiles: one or two per worker
    (2048, 256),     # ONE column group: 32 workers, one tile each (no pipelining at all)
    (4096, 512),     # two column groups, 64 tiles on 64 workers
    (4100, 2560),    # ragged: the last tile has 4 live rows
    (12288, 2560),   # 192 tiles on 24 workers: 8 per worker (even trip count)
    (2112, 512),     # 33 tiles: odd, last tile whole
]


@pytest.mark.parametrize("m,n2", WSG_CASES, ids=lambda v: str(v))
@pytest.mark.parametrize("ln", [True, False], ids=["ln-fold", "plain"])
def test_wsgemm_geglu_matches_torch(m, n2, ln):
    x, w, bias, ln_w, ln_b = make_case(m, n2, m + n2, ln)
    out, _ = _lib.geglu_ln(x, w, bias, ln_w, ln_b, kernel=2)
    close(out, geglu_ln_ref(x, w, bias, ln_w, ln_b), f"wsgemm geglu M={m} N2={n2} ln={ln}")


def test_wsgemm_geglu_matches_the_tiled_kernels():
    """Same arithmetic as the launches it replaces (fp16 operands, fp32 accumulate, one rounding): both within tolerance of
    the fp32 reference and of each other; the statistics' summation order is the only difference."""
    x, w, bias, ln_w, ln_b = make_case(8192, 2560, 7)
    new, _ = _lib.geglu_ln(x, w, bias, ln_w, ln_b, kernel=2)
    old, _ = _lib.geglu_ln(x, w, bias, ln_w, ln_b, kernel=1)
    ref = geglu_ln_ref(x, w, bias, ln_w, ln_b)
    close(new, ref, "wsgemm vs fp32")
    close(old, ref, "tiled vs fp32")
    close(new, old.astype(np.float32), "wsgemm vs tiled", min_psnr=66.0)
    dflt, _ = _lib.geglu_ln(x, w, bias, ln_w, ln_b, kernel=0)
    assert np.array_equal(dflt, new), "the library's own plan for this shape is the weight-stationary kernel"


def test_wsgemm_geglu_rows_far_from_zero_mean():
    """LayerNorm fold with a large common offset (mean 6, std ~1.5): E[x^2] - mean^2 in fp32 from fp16 inputs."""
    x, w, bias, ln_w, ln_b = make_case(4096, 512, 11, offset=6.0)
    out, _ = _lib.geglu_ln(x, w, bias, ln_w, ln_b, kernel=2)
    close(out, geglu_ln_ref(x, w, bias, ln_w, ln_b), "wsgemm geglu offset rows")


def test_wsgemm_geglu_bit_reproducible():
    x, w, bias, ln_w, ln_b = make_case(8192, 2560, 3)
    a, _ = _lib.geglu_ln(x, w, bias, ln_w, ln_b, kernel=2)
    for _ in range(3):
        b, _ = _lib.geglu_ln(x, w, bias, ln_w, ln_b, kernel=2)
        assert np.array_equal(a, b)


def test_wsgemm_refuses_other_shapes():
    rs = np.random.RandomState(0)
    x = h16(rs.randn(2048, 640))
    w = h16(rs.randn(5120, 640) / 25.0)
    with pytest.raises(ValueError):
        _lib.geglu_ln(x, w, None, None, None, kernel=2)      # K = 640: not this kernel's shape
    out, _ = _lib.geglu_ln(x, w, None, None, None, kernel=0)  # ... the library's plan still runs it on the tiled kernels
    close(out, geglu_ln_ref(x, w, None, None, None), "tiled geglu K=640")


# ---- bvgemm.hip (plan tile 11): weights global -> VGPR, activations alone in LDS --------------------------------------------
def conv1x1_ref(x, w, bias, res):
    y = F.conv2d(torch.from_numpy(x.astype(np.float32)), torch.from_numpy(w.astype(np.float32)), None if bias is None else torch.from_numpy(bias))
    if res is not None:
        y = y + torch.from_numpy(res.astype(np.float32))
    return y


BV_CASES = [  # (B, Cin, H, W, Cout)
    (2, 1280, 16, 16, 1280),   # attn.to_out / proj_in of the 16x16 level (M = 512)
    (2, 5120, 16, 16, 1280),   # ff.net.2 of the 16x16 level: 80 K stages
    (2, 1280, 8, 8, 1280),     # M = 128: one row tile, half of it (BM = 128: exactly one)
    (1, 256, 9, 15, 256),      # ragged M = 135: rows past the end, the smallest K / N (4 stages = the ring depth)
    (3, 320, 20, 20, 512),     # M = 1200: last row tile ragged for both tile heights, 5 stages (tail of the 4-stage trip)
    (2, 1280, 32, 32, 2560),   # M = 2048, ten column tiles
    (2, 448, 16, 16, 256),     # 7 K stages: three-stage tail
]


@pytest.mark.parametrize("case", BV_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("tile", [110, 111, 112, 113, 114, 115, 116], ids=["auto", "64x8x32", "128x8x32", "128x4x64", "128x4x32", "32x2x32", "128x2x32"])
@pytest.mark.parametrize("with_res", [True, False], ids=["res", "nores"])
def test_bvgemm_matches_torch(case, tile, with_res):
    b, cin, hh, ww, cout = case
    rs = np.random.RandomState(sum(case) + tile)
    x = h16(rs.randn(b, cin, hh, ww))
    w = h16(rs.randn(cout, cin, 1, 1) / np.sqrt(cin))
    bias = (0.1 * rs.randn(cout)).astype(np.float32)
    res = h16(rs.randn(b, cout, hh, ww)) if with_res else None
    out, _ = _lib.conv2d(x, w, bias, res, tile=tile)
    close(out, conv1x1_ref(x, w, bias, res), f"bvgemm {case} tile {tile} res={with_res}")
    again, _ = _lib.conv2d(x, w, bias, res, tile=tile)
    assert np.array_equal(out, again), "bit-reproducible"


@pytest.mark.parametrize("m,c,n2", [(512, 1280, 10240), (2048, 640, 5120), (8192, 320, 2560), (1000, 640, 512)], ids=lambda v: str(v))
@pytest.mark.parametrize("kernel", [3, 4, 5, 6, 7, 8, 9], ids=["auto", "64x8x32", "128x8x32", "128x4x64", "128x4x32", "32x2x32", "128x2x32"])
@pytest.mark.parametrize("ln", [True, False], ids=["ln-fold", "plain"])
def test_bvgemm_geglu_matches_torch(m, c, n2, kernel, ln):
    rs = np.random.RandomState(m + c + kernel)
    x = h16(rs.randn(m, c) * (1.0 + rs.rand(m, 1)))
    w = h16(rs.randn(n2, c) / np.sqrt(c))
    bias = (0.1 * rs.randn(n2)).astype(np.float32)
    ln_w = (1.0 + 0.2 * rs.randn(c)).astype(np.float32) if ln else None
    ln_b = (0.1 * rs.randn(c)).astype(np.float32) if ln else None
    out, _ = _lib.geglu_ln(x, w, bias, ln_w, ln_b, kernel=kernel)
    close(out, geglu_ln_ref(x, w, bias, ln_w, ln_b), f"bvgemm geglu M={m} C={c} N2={n2} kernel={kernel} ln={ln}")


@pytest.mark.parametrize("case", [(2, 640, 32, 32, 640), (2, 2560, 32, 32, 640), (1, 1280, 24, 24, 1920), (2, 1280, 64, 64, 320)], ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("tile", [110, 114, 116], ids=["auto", "128x4x32", "128x2x32"])
def test_bvgemm_128_column_tiles(case, tile):
    """N a multiple of 128 (64) but not of 256 (the 640- / 320-channel levels' to_out / ff.net.2): only the narrow variants apply."""
    b, cin, hh, ww, cout = case
    if tile == 114 and cout % 128 != 0:
        pytest.skip("128-column tiles need N % 128 == 0")
    rs = np.random.RandomState(sum(case))
    x = h16(rs.randn(b, cin, hh, ww))
    w = h16(rs.randn(cout, cin, 1, 1) / np.sqrt(cin))
    bias = (0.1 * rs.randn(cout)).astype(np.float32)
    res = h16(rs.randn(b, cout, hh, ww))
    out, _ = _lib.conv2d(x, w, bias, res, tile=tile)
    close(out, conv1x1_ref(x, w, bias, res), f"bvgemm {case} tile {tile}")
    with pytest.raises(ValueError):
        _lib.conv2d(x, w, bias, res, tile=113)   # 256-column variants refuse it


# ---- fused q|k|v epilogue of bvgemm.hip (LayerNorm fold, pre-scaled queries, V^T in attention8's key order) --------------------
def qkv_ref(x, ln_w, ln_b, w, batch, q_scale, vt_perm, eps=1e-5):
    xt = F.layer_norm(torch.from_numpy(x.astype(np.float32)), (x.shape[1],), torch.from_numpy(ln_w), torch.from_numpy(ln_b), eps)
    y = (xt @ torch.from_numpy(w.astype(np.float32)).T).numpy()
    c = x.shape[1]
    qk = np.concatenate([y[:, :c] * q_scale, y[:, c:2 * c]], axis=1)
    hw = x.shape[0] // batch
    vt = y[:, 2 * c:].reshape(batch, hw, c).transpose(0, 2, 1)                 # [B][C][HW]
    if vt_perm:                                                                # position p of a row holds token t(p) (AttnDesc::vt_perm)
        p = np.arange(hw)
        c8, e = (p % 16) // 8, p % 8
        t = (p // 16) * 16 + 4 * c8 + np.where(e < 4, e, 8 + (e - 4))
        vt = vt[:, :, t]
    return qk, vt


@pytest.mark.parametrize("batch,hw,c", [(2, 1024, 640), (2, 256, 1280), (16, 256, 1280), (3, 128, 320), (4, 1024, 640)], ids=lambda v: str(v))
@pytest.mark.parametrize("kernel", [1, 3, 5, 6, 7], ids=["tiled", "bv-auto", "bv-128x8x32", "bv-128x4x64", "bv-128x4x32"])
@pytest.mark.parametrize("perm", [True, False], ids=["vt-perm", "vt-plain"])
def test_qkv_ln_matches_torch(batch, hw, c, kernel, perm):
    if c % 128 != 0 and kernel >= 3:   # N = 3C must be a multiple of 128 (C = 320: the tiled kernels keep that shape)
        pytest.skip("bvgemm's q|k|v epilogue needs N % 128 == 0")
    if (3 * c) % 256 != 0 and kernel in (5, 6):
        pytest.skip("256-column tiles need N % 256 == 0")
    rs = np.random.RandomState(batch + hw + c)
    x = h16(rs.randn(batch * hw, c) * (1.0 + rs.rand(batch * hw, 1)))
    w = h16(rs.randn(3 * c, c) / np.sqrt(c))
    ln_w = (1.0 + 0.2 * rs.randn(c)).astype(np.float32)
    ln_b = (0.1 * rs.randn(c)).astype(np.float32)
    q_scale = 1.4426950408889634 / 8.0
    qk, vt, _ = _lib.qkv_ln(x, ln_w, ln_b, w, batch, q_scale=q_scale, vt_perm=perm, kernel=kernel)
    rqk, rvt = qkv_ref(x, ln_w, ln_b, w, batch, q_scale, perm)
    close(qk, rqk, f"q|k kernel {kernel} B={batch} HW={hw} C={c}")
    close(vt, rvt, f"V^T kernel {kernel} B={batch} HW={hw} C={c} perm={perm}")


@pytest.mark.parametrize("batch,hw", [(2, 4096), (1, 2112), (3, 1024), (16, 4096)], ids=lambda v: str(v))
@pytest.mark.parametrize("perm", [True, False], ids=["vt-perm", "vt-plain"])
def test_wsgemm_qkv_matches_torch_and_tiled(batch, hw, perm):
    """The 320-channel level's fused q|k|v on the weight-stationary kernel (five waves per workgroup, six column groups: four q|k,
    two V^T), against fp32 torch and against the tiled kernel.  Built, correct and NOT selected by the library: 21.2 vs 17.0 us at
    M = 8 192 (tools/r6_qkv_bench.py), so the library's own plan for the shape stays the tiled kernel."""
    c = 320
    rs = np.random.RandomState(batch * 7 + hw)
    x = h16(rs.randn(batch * hw, c) * (1.0 + rs.rand(batch * hw, 1)))
    w = h16(rs.randn(3 * c, c) / np.sqrt(c))
    ln_w = (1.0 + 0.2 * rs.randn(c)).astype(np.float32)
    ln_b = (0.1 * rs.randn(c)).astype(np.float32)
    q_scale = 1.4426950408889634 / 8.0
    qk, vt, _ = _lib.qkv_ln(x, ln_w, ln_b, w, batch, q_scale=q_scale, vt_perm=perm, kernel=2)
    rqk, rvt = qkv_ref(x, ln_w, ln_b, w, batch, q_scale, perm)
    close(qk, rqk, f"wsgemm q|k B={batch} HW={hw}")
    close(vt, rvt, f"wsgemm V^T B={batch} HW={hw} perm={perm}")
    if batch <= 3:
        tqk, tvt, _ = _lib.qkv_ln(x, ln_w, ln_b, w, batch, q_scale=q_scale, vt_perm=perm, kernel=1)
        close(qk, tqk.astype(np.float32), "wsgemm vs tiled q|k", min_psnr=66.0)
        close(vt, tvt.astype(np.float32), "wsgemm vs tiled V^T", min_psnr=66.0)
        dqk, dvt, _ = _lib.qkv_ln(x, ln_w, ln_b, w, batch, q_scale=q_scale, vt_perm=perm, kernel=0)
        assert np.array_equal(dqk, tqk) and np.array_equal(dvt, tvt), "the library's own plan for this shape stays the tiled kernel"


# ---- the head of a SpatialTransformer in one launch: GroupNorm apply -> proj_in -> LayerNorm -> q|k|v (xattn_out.hip) -----------
def gn_proj_qkv_ref(x_in, conv_w, gn_w, gn_b, proj_w, proj_b, ln_w, ln_b, wqkv, q_scale, vt_perm):
    x = F.conv2d(torch.from_numpy(x_in.astype(np.float32)), torch.from_numpy(conv_w.astype(np.float32))[:, :, None, None])
    x = x.half().float()                                                            # the producer stores fp16
    xn = F.group_norm(x, 32, torch.from_numpy(gn_w), torch.from_numpy(gn_b), 1e-6)  # unet.py:528-531
    b, c, hh, ww = xn.shape
    tok = xn.permute(0, 2, 3, 1).reshape(b * hh * ww, c)
    h = tok @ torch.from_numpy(proj_w.astype(np.float32)).T + torch.from_numpy(proj_b)
    qk, vt = qkv_ref(h.numpy(), ln_w, ln_b, wqkv, b, q_scale, vt_perm)
    return h.numpy(), qk, vt


@pytest.mark.parametrize("batch,hh,ww", [(2, 64, 64), (1, 32, 32), (3, 16, 32)], ids=lambda v: str(v))
@pytest.mark.parametrize("fused", [1, 2, 0], ids=["one-launch", "one-launch-64-tokens", "three-launches"])
@pytest.mark.parametrize("perm", [True, False], ids=["vt-perm", "vt-plain"])
def test_gn_proj_qkv_matches_torch(batch, hh, ww, fused, perm):
    c = 320
    rs = np.random.RandomState(batch + hh + ww)
    x_in = h16(rs.randn(batch, c, hh, ww))
    conv_w = h16(rs.randn(c, c) / np.sqrt(c) * (1.0 + rs.rand(c, 1)))              # groups of different scale
    gn_w = (1.0 + 0.2 * rs.randn(c)).astype(np.float32)
    gn_b = (0.1 * rs.randn(c)).astype(np.float32)
    proj_w = h16(rs.randn(c, c) / np.sqrt(c))
    proj_b = (0.1 * rs.randn(c)).astype(np.float32)
    ln_w = (1.0 + 0.2 * rs.randn(c)).astype(np.float32)
    ln_b = (0.1 * rs.randn(c)).astype(np.float32)
    wqkv = h16(rs.randn(3 * c, c) / np.sqrt(c))
    q_scale = 1.4426950408889634 / 8.0
    h, qk, vt, entries, _ = _lib.gn_proj_qkv(x_in, conv_w, gn_w, gn_b, proj_w, proj_b, ln_w, ln_b, wqkv, q_scale=q_scale, vt_perm=perm, fused=fused)
    rh, rqk, rvt = gn_proj_qkv_ref(x_in, conv_w, gn_w, gn_b, proj_w, proj_b, ln_w, ln_b, wqkv, q_scale, perm)
    if fused:
        assert entries >= 1, "the producer's statistics were folded by the fused launch"
    close(h, rh, f"proj_in output B={batch} {hh}x{ww} fused={fused}")
    # q|k|v are a second GEMM over the fp16-rounded h: compare against the reference run on the kernel's own h
    rqk2, rvt2 = qkv_ref(h.astype(np.float32), ln_w, ln_b, wqkv, batch, q_scale, perm)
    close(qk, rqk2, f"q|k B={batch} {hh}x{ww} fused={fused}")
    close(vt, rvt2, f"V^T B={batch} {hh}x{ww} fused={fused} perm={perm}")
    close(qk, rqk, "q|k end to end", min_psnr=55.0, rel=1e-2)
    close(vt, rvt, "V^T end to end", min_psnr=55.0, rel=1e-2)


def test_gn_proj_qkv_one_launch_vs_three():
    c = 320
    rs = np.random.RandomState(5)
    x_in = h16(rs.randn(2, c, 64, 64))
    conv_w = h16(rs.randn(c, c) / np.sqrt(c))
    args = (x_in, conv_w, (1 + 0.1 * rs.randn(c)).astype(np.float32), (0.1 * rs.randn(c)).astype(np.float32), h16(rs.randn(c, c) / np.sqrt(c)),
            (0.1 * rs.randn(c)).astype(np.float32), (1 + 0.1 * rs.randn(c)).astype(np.float32), (0.1 * rs.randn(c)).astype(np.float32),
            h16(rs.randn(3 * c, c) / np.sqrt(c)))
    h1, qk1, vt1, e1, _ = _lib.gn_proj_qkv(*args, q_scale=0.18, fused=True)
    h0, qk0, vt0, _, _ = _lib.gn_proj_qkv(*args, q_scale=0.18, fused=False)
    h64, qk64, vt64, e64, _ = _lib.gn_proj_qkv(*args, q_scale=0.18, fused=2)
    assert e64 == e1 and np.array_equal(h1, h64) and np.array_equal(qk1, qk64) and np.array_equal(vt1, vt64), "64-token workgroups: the same bits"
    close(h1, h0.astype(np.float32), "h one launch vs three", min_psnr=66.0)
    close(qk1, qk0.astype(np.float32), "q|k one launch vs three", min_psnr=60.0)
    close(vt1, vt0.astype(np.float32), "V^T one launch vs three", min_psnr=60.0)
    h2, qk2, vt2, e2, _ = _lib.gn_proj_qkv(*args, q_scale=0.18, fused=True)
    assert e1 == e2 and np.array_equal(h1, h2) and np.array_equal(qk1, qk2) and np.array_equal(vt1, vt2), "bit-reproducible"


# ---- attention8's balanced form: (query tile, key tile) units dealt out evenly, partial (m, l, O) merged by the last arriver --------
def attn_ref(q, k, v, heads):
    b, c, _, sq = q.shape
    d = c // heads
    qt = torch.from_numpy(q.astype(np.float32)).reshape(b, heads, d, sq)
    kt = torch.from_numpy(k.astype(np.float32)).reshape(b, heads, d, -1)
    vt = torch.from_numpy(v.astype(np.float32)).reshape(b, heads, d, -1)
    w = torch.softmax(torch.einsum("bhdq,bhdk->bhqk", qt, kt) * d ** -0.5, dim=-1)          # attention.py:24-40
    return torch.einsum("bhqk,bhdk->bhdq", w, vt).reshape(b, c, 1, sq).numpy()


@pytest.mark.parametrize("batch,heads,sq,sk,upw", [
    (2, 5, 4096, 4096, 0),      # the UNet's 64x64 level at batch 2: 40 key tiles per workgroup, every query tile merged from 2-3 partials
    (2, 10, 1024, 1024, 0),     # the 32x32 level: four-wave workgroups
    (1, 2, 256, 256, 3),        # three of four key tiles: segments of 3 | 1+2 | 2+1 | 3
    (1, 2, 256, 256, 1),        # one key tile per workgroup: four partials per query tile
    (1, 3, 512, 320, 7),        # a split that crosses into the next query tile mid-way, ragged end
    (1, 1, 128, 640, 4),        # one query tile, ten key tiles in three segments
    (2, 2, 256, 128, 5),        # more units per workgroup than a query tile has: whole tiles stored directly, three segments per workgroup
], ids=lambda v: str(v))
def test_attention8_balanced_form(batch, heads, sq, sk, upw):
    rs = np.random.RandomState(sq + sk + upw)
    c = heads * 64
    q = h16(rs.randn(batch, c, 1, sq) * 1.5)
    k = h16(rs.randn(batch, c, 1, sk) * 1.5)
    k[:, :, :, sk // 2:] *= 1.8                                  # the later keys carry the larger scores: the running max moves between segments
    v = h16(rs.randn(batch, c, 1, sk))
    got, _ = _lib.attention("ORIGINAL", q, k, v, heads, 64, variant=100 + upw)
    classic, _ = _lib.attention("ORIGINAL", q, k, v, heads, 64, variant=0)
    ref = attn_ref(q, k, v, heads)
    close(got, ref, f"balanced attention B={batch} h={heads} {sq}x{sk} upw={upw}", min_psnr=60.0, rel=6e-3)
    close(got, classic.astype(np.float32), "balanced vs classic grid", min_psnr=66.0, rel=4e-3)
    again, _ = _lib.attention("ORIGINAL", q, k, v, heads, 64, variant=100 + upw, iters=3)    # the counters come back to zero; arrival order does not matter
    assert np.array_equal(got, again), "bit-reproducible"


# ---- results must not depend on what the LDS held before a launch (LAB_NOTES Finding 20) ------------------------------------------
def test_tiny_unet_behind_poisoned_lds():
    """SD_POISON_LDS=1 (with SD_TUNE) fills every CU's LDS with NaN bit patterns in front of every op: a kernel that reads LDS it
    never wrote behind a zero weight (conv_small_cin_kernel did, for the tiny UNet's 48-wide prompt) turns the output into NaN."""
    import os
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, SD_TUNE="1", SD_POISON_LDS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_unet_gpu.py"), "-m", "gpu", "-q", "-x", "-k", "tiny or mini"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
