"""Unit parity of every HIP kernel family against a plain fp32 torch statement of the same op.

backend "emu": the SIMT-emulator build of the same kernel sources (index logic, runs without a GPU);
backend "hip": the real gfx950 library on a MI355X (-m gpu).  Both go through the C ABI.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from star_amd import lib as L
from util import BACKENDS, DTYPES, assert_close, make_ctx


@pytest.fixture(params=BACKENDS)
def backend(request):
    return request.param


@pytest.fixture(params=DTYPES)
def dtype(request):
    return request.param


@pytest.fixture
def ctx(backend, dtype, request):
    emu = request.getfixturevalue("emu_lib") if backend == "emu" else None
    c = make_ctx(backend, dtype, emu)
    yield c
    c.sync()
    c.close()


def dev(ctx, t):
    return t.to(ctx.torch_device)


PRODUCT_TILES = (0, 1, 2, 3, 4, 9, 10, 17, 18, 19, 30)


_BENCH_CTX = {}


def need_variant(ctx, product_ok):
    """A/B kernels and experimental tiles live only in -DSTAR_BENCH_VARIANTS builds (the emulator, tools/bench): the product
    library must REJECT them (test_product_library_rejects_bench_variants).  On hardware their parity tests run against the
    bench build when it has been built (`make bench`), else they are skipped.  Returns the context to use."""
    if product_ok or ctx.lib.has_bench_variants:
        return ctx
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench", "libstar_hip_bench.so")
    if not os.path.isfile(path):
        pytest.skip("bench-only variant: not in the product library (build tools/bench with `make bench` to test it on hardware)")
    key = ctx.dtype
    if key not in _BENCH_CTX:
        _BENCH_CTX[key] = L.Context(0, ctx.dtype, L.Library(path))
    return _BENCH_CTX[key]


def nhwc_rows(x):  # [N, C, H, W] -> [N*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


GEMM_CASES = [  # M, N, K, tile
    (300, 192, 128, 3), (256, 256, 64, 1), (515, 320, 128, 2), (260, 128, 64, 4), (100, 72, 64, 0),
    (77, 1280, 1024, 0), (1, 64, 64, 0), (600, 960, 320, 0), (513, 640, 192, 0), (515, 512, 256, 5), (700, 1024, 64, 5),
    (515, 512, 256, 7), (700, 640, 192, 8), (300, 320, 64, 8), (257, 256, 192, 7), (600, 960, 64, 8),
    (515, 512, 256, 9), (130, 256, 64, 9), (700, 640, 192, 10), (300, 320, 64, 10), (257, 960, 128, 10), (400, 264, 320, 9), (515, 512, 256, 14), (300, 264, 64, 14),
]


@pytest.mark.parametrize("M,N,K,tile", GEMM_CASES)
def test_gemm_bias_residual(ctx, dtype, M, N, K, tile):
    ctx = need_variant(ctx, tile in PRODUCT_TILES)
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g).to(dtype)
    out = ctx.gemm(dev(ctx, A), dev(ctx, W), bias=dev(ctx, b), res=dev(ctx, R), force_tile=tile)
    ref = A.float() @ W.float().T + b + R.float()
    assert_close(out, ref, dtype, what="gemm")
    out32 = ctx.gemm(dev(ctx, A), dev(ctx, W), out_f32=True, force_tile=tile)
    assert out32.dtype == torch.float32
    assert_close(out32, A.float() @ W.float().T, dtype, what="gemm f32 out")


def test_gemm_strided_views(ctx, dtype):
    """A and the output may be column slices of wider buffers (fused QKV, concat targets)."""
    g = torch.Generator().manual_seed(3)
    big = torch.randn(200, 256, generator=g).to(dtype)
    W = (torch.randn(64, 128, generator=g) * 0.1).to(dtype)
    bigd = dev(ctx, big)
    outbuf = torch.zeros(200, 192, dtype=dtype, device=ctx.torch_device)
    ctx.gemm(bigd[:, 64:192], dev(ctx, W), out=outbuf[:, 64:128])
    ref = big[:, 64:192].float() @ W.float().T
    assert_close(outbuf[:, 64:128], ref, dtype, what="gemm strided")
    assert float(outbuf[:, :64].abs().max()) == 0 and float(outbuf[:, 128:].abs().max()) == 0


@pytest.mark.parametrize("M,K,Nh", [(70, 64, 128), (300, 320, 1280)])
def test_gemm_geglu(ctx, dtype, M, K, Nh):
    """GEGLU epilogue (unet_v2v.py:496-504) with value/gate weight rows interleaved in 32-row blocks."""
    from star_amd.topology import geglu_interleave
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).to(dtype)
    Wg = (torch.randn(2 * Nh, K, generator=g) / math.sqrt(K)).to(dtype)
    bg = torch.randn(2 * Nh, generator=g)
    proj = A.float() @ Wg.float().T + bg
    ref = proj[:, :Nh] * F.gelu(proj[:, Nh:])
    idx = geglu_interleave(Nh)
    out = ctx.gemm(dev(ctx, A), dev(ctx, Wg[idx].contiguous()), bias=dev(ctx, bg[idx].contiguous()), geglu=True)
    assert_close(out, ref, dtype, what="geglu")


@pytest.mark.parametrize("M,N,K", [(300, 192, 128), (515, 640, 192), (77, 1280, 256)])
def test_gemm_gelu_tanh(ctx, dtype, M, N, K):
    """tanh-GELU epilogue (STAR_EPI_GELU_TANH: the MLP activation of the CogVideoX DiT block, sat's gelu_impl) on the auto tiles."""
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g)
    ref = F.gelu(A.float() @ W.float().T + b, approximate="tanh")
    out = ctx.gemm(dev(ctx, A), dev(ctx, W), bias=dev(ctx, b), gelu_tanh=True)
    assert_close(out, ref, dtype, what="gelu_tanh")
    with pytest.raises(L.StarError):      # the flavour exists for plain-A layers without a residual only
        ctx.gemm(dev(ctx, A), dev(ctx, W), bias=dev(ctx, b), res=dev(ctx, torch.zeros(M, N).to(dtype)), gelu_tanh=True)


def test_gather_offset_guard(ctx, dtype):
    """the implicit-GEMM gathers address their input with 32-bit offsets from a per-tile base (a few image rows): an image row
    pitch that cannot fit must be refused by the launcher, not wrapped (the check runs before anything is launched)."""
    A = torch.zeros(64, 64).to(dtype)
    W = torch.zeros(64, 9 * 64).to(dtype)
    with pytest.raises(L.StarError, match="32-bit gather offsets"):
        ctx.gemm(dev(ctx, A), dev(ctx, W), mode=L.A_CONV3X3, conv=(1, 1, 1 << 22, 64, 1, 1 << 22, 1, 1, 1),
                 out=dev(ctx, torch.zeros(8, 64).to(dtype)))


CONV_CASES = [  # NB, Cin, H, W, Cout
    (2, 64, 10, 8, 96), (1, 128, 18, 16, 64), (3, 320, 10, 8, 320),
]


@pytest.mark.parametrize("NB,Cin,H,Wd,Cout", CONV_CASES)
def test_conv3x3_variants(ctx, dtype, NB, Cin, H, Wd, Cout):
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn(NB, Cin, H, Wd, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dtype)
    b = torch.randn(Cout, generator=g)
    wp = L.pack_conv3x3_weight(w)   # K index (c // 64, tap, c % 64): the kernel walks the nine taps of a 64-channel block back to back
    xr, wpd, bd = dev(ctx, nhwc_rows(x)), dev(ctx, wp), dev(ctx, b)
    # ResBlock / Upsample conv: 3x3 stride 1 pad 1 (unet_v2v.py:612,639)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    out = ctx.gemm(xr, wpd, bias=bd, mode=L.A_CONV3X3, conv=(NB, H, Wd, Cin, H, Wd, 1, 1, 1))
    assert_close(out, nhwc_rows(ref), dtype, what="conv3x3 s1")
    # Downsample: stride 2, padding (2, 1) (unet_v2v.py:709-722)
    ref = F.conv2d(x.float(), w.float(), b, stride=2, padding=(2, 1))
    Ho, Wo = ref.shape[2:]
    assert Ho == H // 2 + 1 and Wo == Wd // 2
    out = ctx.gemm(xr, wpd, bias=bd, mode=L.A_CONV3X3, conv=(NB, H, Wd, Cin, Ho, Wo, 2, 2, 1))
    assert_close(out, nhwc_rows(ref), dtype, what="conv3x3 s2")
    # Upsample: nearest x2, drop first/last row, conv (unet_v2v.py:563-566)
    xu = F.interpolate(x.float(), scale_factor=2, mode="nearest")[..., 1:-1, :]
    ref = F.conv2d(xu, w.float(), b, padding=1)
    Ho, Wo = ref.shape[2:]
    out = ctx.gemm(xr, wpd, bias=bd, mode=L.A_CONV3X3_UP, conv=(NB, H, Wd, Cin, Ho, Wo, 1, 1, 1))
    assert_close(out, nhwc_rows(ref), dtype, what="conv3x3 up")


@pytest.mark.parametrize("Fr,H,Wd,C", [(5, 3, 4, 64), (8, 6, 8, 128), (1, 4, 4, 64)])
def test_temporal_conv(ctx, dtype, Fr, H, Wd, C):
    """Conv3d (3,1,1) pad (1,0,0) + identity (TemporalConvBlock_v2, unet_v2v.py:1209-1220,1277)."""
    g = torch.Generator().manual_seed(Fr)
    x = torch.randn(1, C, Fr, H, Wd, generator=g).to(dtype)
    w = (torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)).to(dtype)
    b = torch.randn(C, generator=g)
    res = torch.randn(Fr * H * Wd, C, generator=g).to(dtype)
    ref = F.conv3d(x.float(), w.float(), b, padding=(1, 0, 0))[0].permute(1, 2, 3, 0).reshape(-1, C) + res.float()
    a = x[0].permute(1, 2, 3, 0).reshape(-1, C).contiguous()
    wp = w[:, :, :, 0, 0].permute(0, 2, 1).reshape(C, 3 * C).contiguous()
    out = ctx.gemm(dev(ctx, a), dev(ctx, wp), bias=dev(ctx, b), res=dev(ctx, res), mode=L.A_TCONV3, temporal=(Fr, H * Wd, C))
    assert_close(out, ref, dtype, what="tconv")


@pytest.mark.parametrize("Fr,HW,C,tile", [(3, 300, 64, 3), (4, 300, 64, 2), (5, 257, 128, 3), (2, 700, 256, 17), (33, 130, 64, 3)])
def test_temporal_conv_frame_interleaved_walk(ctx, dtype, Fr, HW, C, tile):
    """A_TCONV3 with frames of at least one tile row: the launcher walks the tile rows frame-interleaved (gemm.h t_walk: the same pixel
    block of consecutive frames back to back, so that an XCD's workgroups share the input frame tiles in its L2).  Every tile row is
    still computed exactly once, whatever the ratio of frame length to tile height: the result matches torch's Conv3d (3,1,1)."""
    g = torch.Generator().manual_seed(Fr * HW + tile)
    x = torch.randn(1, C, Fr, HW, 1, generator=g).to(dtype)
    w = (torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)).to(dtype)
    b = torch.randn(C, generator=g)
    ref = F.conv3d(x.float(), w.float(), b, padding=(1, 0, 0))[0].permute(1, 2, 3, 0).reshape(-1, C)
    a = x[0].permute(1, 2, 3, 0).reshape(-1, C).contiguous()
    wp = w[:, :, :, 0, 0].permute(0, 2, 1).reshape(C, 3 * C).contiguous()
    out = ctx.gemm(dev(ctx, a), dev(ctx, wp), bias=dev(ctx, b), mode=L.A_TCONV3, temporal=(Fr, HW, C), force_tile=tile)
    assert_close(out, ref, dtype, what="tconv, frame-interleaved walk")


@pytest.mark.parametrize("M,N,K,wgs", [(515, 512, 256, 0), (300, 264, 64, 0), (700, 600, 192, 2), (257, 1288, 128, 3), (1030, 256, 448, 1), (64, 8, 64, 0), (700, 576, 128, 2), (700, 600, 64, 2),
                                        (2100, 1920, 640, 8)])
def test_gemm_persistent_tile(ctx, dtype, M, N, K, wgs):
    """tile 18 (gemm_p.h): resident workgroups walk their output tiles, the K tiles of all of them form one LDS-DMA stream (operands
    through hand-built buffer descriptors: ragged rows / columns read as zeros), zero-C MFMAs open a tile, wave-private epilogue.
    Bias / residual / folded-LayerNorm flavours against fp32 and BIT FOR BIT against the 8-wave tile (same k order per output);
    `wgs` resident workgroups (force_tile 2000 + n) so that a workgroup streams across several output tiles, with odd and single K
    tile counts and ragged edges; fp32 output is refused."""
    if M * N * K > 6e8 and ctx.lib.is_hostemu:
        pytest.skip("hardware-only size")
    g = torch.Generator().manual_seed(M + 3 * N + K)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g).to(dtype)
    rowab = torch.stack([torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)], dim=1).contiguous()
    colsum = W.float().sum(1).contiguous()
    Ad, Wd_, bd, Rd = dev(ctx, A), dev(ctx, W), dev(ctx, b), dev(ctx, R)
    ft = 2000 + wgs if wgs else 18
    acc = A.float() @ W.float().T
    for kw, ref in ((dict(bias=bd), acc + b), (dict(), acc), (dict(bias=bd, res=Rd), acc + b + R.float()),
                    (dict(bias=bd, rowab=dev(ctx, rowab), colsum=dev(ctx, colsum)), rowab[:, :1] * acc + rowab[:, 1:] * colsum[None] + b[None])):
        out = ctx.gemm(Ad, Wd_, force_tile=ft, **kw)
        assert torch.equal(out, ctx.gemm(Ad, Wd_, force_tile=1, **kw)), kw.keys()
        assert_close(out, ref, dtype, scale=6.0, what=f"gemm tile 18 {list(kw)}")
    with pytest.raises(L.StarError):
        ctx.gemm(Ad, Wd_, bias=bd, out_f32=True, force_tile=18)
    if N % 64 == 0:   # GEGLU (32-row value / gate blocks), plain and with the folded LayerNorm: bit for bit against the 8-wave tile
        for kw in (dict(bias=bd, geglu=True), dict(bias=bd, geglu=True, rowab=dev(ctx, rowab), colsum=dev(ctx, colsum))):
            out = ctx.gemm(Ad, Wd_, force_tile=ft, **kw)
            assert out.shape == (M, N // 2) and torch.equal(out, ctx.gemm(Ad, Wd_, force_tile=1, **kw)), kw.keys()
        with pytest.raises(L.StarError):
            ctx.gemm(Ad, Wd_, bias=bd, res=Rd, geglu=True, force_tile=18)


def test_gemm_tail_split_is_bit_identical(ctx, dtype):
    """The launcher gives a poorly filled LAST ROUND of big tiles (one workgroup per CU) to a second launch of 128 x 128 tiles over
    the remaining rows (gemm_impl.h: tail split; the kernel's m_off).  force_tile = 1000 + n balances the rounds for n CUs, so small
    problems split: plain / residual / GEGLU / folded-LayerNorm / fp32-output epilogues, the 3x3 conv (stride 1 and the stride-2
    Downsample) and the temporal conv -- whose gathers derive (frame, y, x) from the absolute row -- all bit-identical to the
    unsplit launch."""
    g = torch.Generator().manual_seed(91)
    n0 = ctx.lib.gemm_split_count(ctx.h)
    # plain A, 256 x 320 tiles (N = 1280: 4 tile columns): 3 x 4 tiles on "8 CUs" -> 2 tile rows in one full round + 88 rows of small tiles
    M, N, K = 600, 1280, 128
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g).to(dtype)
    Ad, Wd_, bd, Rd = dev(ctx, A), dev(ctx, W), dev(ctx, b), dev(ctx, R)
    whole = ctx.gemm(Ad, Wd_, bias=bd, res=Rd, force_tile=2)
    assert ctx.lib.gemm_split_count(ctx.h) == n0                    # a forced tile is never split
    split = ctx.gemm(Ad, Wd_, bias=bd, res=Rd, force_tile=1008)
    assert ctx.lib.gemm_split_count(ctx.h) == n0 + 1                # ... the automatic choice on "8 CUs" is
    assert torch.equal(whole, split)
    assert_close(split, A.float() @ W.float().T + b + R.float(), dtype, what="gemm tail split")
    # 256 x 256 tiles (N = 1536: 6 tile columns), GEGLU, the folded LayerNorm, both, fp32 output
    M, N, K = 600, 1536, 64
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g)
    rowab = torch.stack([torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)], dim=1).contiguous()
    colsum = W.float().sum(1).contiguous()
    Ad, Wd_, bd = dev(ctx, A), dev(ctx, W), dev(ctx, b)
    for kw in (dict(geglu=True), dict(rowab=dev(ctx, rowab), colsum=dev(ctx, colsum)), dict(geglu=True, rowab=dev(ctx, rowab), colsum=dev(ctx, colsum)), dict(out_f32=True)):
        whole = ctx.gemm(Ad, Wd_, bias=bd, force_tile=1, **kw)
        split = ctx.gemm(Ad, Wd_, bias=bd, force_tile=1008, **kw)
        assert torch.equal(whole, split), kw
    # gathered modes, 64 -> 192 channels (256 x 256 tiles, one tile column), > 4096 rows so that the big tiles are the automatic choice
    NB, Cin, H, Wd, Cout = 10, 64, 22, 20, 192                        # 4400 rows: 18 tile rows on "8 CUs" -> 16 + 304 rows
    x = torch.randn(NB, Cin, H, Wd, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dtype)
    b = torch.randn(Cout, generator=g)
    wp = L.pack_conv3x3_weight(w)   # K index (c // 64, tap, c % 64): the kernel walks the nine taps of a 64-channel block back to back
    xr, wpd, bd = dev(ctx, nhwc_rows(x)), dev(ctx, wp), dev(ctx, b)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    whole = ctx.gemm(xr, wpd, bias=bd, mode=L.A_CONV3X3, conv=(NB, H, Wd, Cin, H, Wd, 1, 1, 1), force_tile=1)
    split = ctx.gemm(xr, wpd, bias=bd, mode=L.A_CONV3X3, conv=(NB, H, Wd, Cin, H, Wd, 1, 1, 1), force_tile=1008)
    assert torch.equal(whole, split)
    assert_close(split, nhwc_rows(ref), dtype, what="conv3x3 tail split")
    # the same conv 64 -> 320 channels: 256 x 320 tiles (one tile column)
    w3 = (torch.randn(320, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dtype)
    b3 = torch.randn(320, generator=g)
    w3d, b3d = dev(ctx, L.pack_conv3x3_weight(w3)), dev(ctx, b3)
    whole = ctx.gemm(xr, w3d, bias=b3d, mode=L.A_CONV3X3, conv=(NB, H, Wd, Cin, H, Wd, 1, 1, 1), force_tile=2)
    split = ctx.gemm(xr, w3d, bias=b3d, mode=L.A_CONV3X3, conv=(NB, H, Wd, Cin, H, Wd, 1, 1, 1), force_tile=1008)
    assert torch.equal(whole, split)
    assert_close(split, nhwc_rows(F.conv2d(x.float(), w3.float(), b3, padding=1)), dtype, what="conv3x3 tail split, 256 x 320 main tiles")
    x4 = torch.randn(35, Cin, H, Wd, generator=g).to(dtype)           # Downsample: 35 frames -> 35 x 12 x 10 = 4200 output rows
    ref = F.conv2d(x4.float(), w.float(), b, stride=2, padding=(2, 1))
    Ho, Wo = ref.shape[2:]
    x4r = dev(ctx, nhwc_rows(x4))
    whole = ctx.gemm(x4r, wpd, bias=bd, mode=L.A_CONV3X3, conv=(35, H, Wd, Cin, Ho, Wo, 2, 2, 1), force_tile=1)
    split = ctx.gemm(x4r, wpd, bias=bd, mode=L.A_CONV3X3, conv=(35, H, Wd, Cin, Ho, Wo, 2, 2, 1), force_tile=1008)
    assert torch.equal(whole, split)
    assert_close(split, nhwc_rows(ref), dtype, what="conv3x3 s2 tail split")
    Fr, C = 22, 192                                                  # temporal conv: 22 frames of 14 x 14 = 4312 rows
    xt = torch.randn(1, C, Fr, 14, 14, generator=g).to(dtype)
    wt = (torch.randn(C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)).to(dtype)
    bt = torch.randn(C, generator=g)
    a = xt[0].permute(1, 2, 3, 0).reshape(-1, C).contiguous()
    wtp = wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(C, 3 * C).contiguous()
    ad, wtd, btd = dev(ctx, a), dev(ctx, wtp), dev(ctx, bt)
    ref = F.conv3d(xt.float(), wt.float(), bt, padding=(1, 0, 0))[0].permute(1, 2, 3, 0).reshape(-1, C) + a.float()
    whole = ctx.gemm(ad, wtd, bias=btd, res=ad, mode=L.A_TCONV3, temporal=(Fr, 196, C), force_tile=1)
    split = ctx.gemm(ad, wtd, bias=btd, res=ad, mode=L.A_TCONV3, temporal=(Fr, 196, C), force_tile=1008)
    assert torch.equal(whole, split)
    assert_close(split, ref, dtype, what="tconv tail split")
    assert ctx.lib.gemm_split_count(ctx.h) == n0 + 1 + 4 + 4        # every automatic launch above was split


@pytest.mark.parametrize("sched_tile", [17, 19])
@pytest.mark.parametrize("M,N,K", [(515, 512, 256), (300, 264, 64), (257, 256, 448), (130, 1280, 128), (600, 640, 192)])
def test_gemm_scheduled_tile(ctx, dtype, M, N, K, sched_tile):
    """tiles 17 / 19 (4 waves x 128 x 128 / 128 x 160 -- the 256 x 320 tile with 320 accumulators per wave, round 6 --, one wave per SIMD, hand-placed 2-stage loop): bias / residual epilogues against fp32,
    and bit-for-bit against the 8-wave tile (both add the k-steps of an output in the same order); odd tile counts exercise the
    first-tile / last-two-tiles paths of the loop.  The tile has no fp32-output and no GEGLU flavour: refused."""
    g = torch.Generator().manual_seed(M + 3 * N + K)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g).to(dtype)
    Ad, Wd_, bd, Rd = dev(ctx, A), dev(ctx, W), dev(ctx, b), dev(ctx, R)
    out = ctx.gemm(Ad, Wd_, bias=bd, res=Rd, force_tile=sched_tile)
    assert_close(out, A.float() @ W.float().T + b + R.float(), dtype, what="gemm scheduled tile +res")
    assert torch.equal(out, ctx.gemm(Ad, Wd_, bias=bd, res=Rd, force_tile=1))
    out = ctx.gemm(Ad, Wd_, bias=bd, force_tile=sched_tile)
    assert torch.equal(out, ctx.gemm(Ad, Wd_, bias=bd, force_tile=1))
    with pytest.raises(L.StarError):
        ctx.gemm(Ad, Wd_, out_f32=True, force_tile=sched_tile)
    if N % 64 == 0:
        with pytest.raises(L.StarError):
            ctx.gemm(Ad, Wd_, bias=bd, geglu=True, force_tile=sched_tile)


@pytest.mark.parametrize("sched_tile", [17, 19])
@pytest.mark.parametrize("NB,Cin,H,Wd,Cout", [(2, 64, 10, 8, 96), (1, 128, 18, 16, 256), (3, 320, 10, 8, 320)])
def test_conv_scheduled_tile(ctx, dtype, NB, Cin, H, Wd, Cout, sched_tile):
    """the gathered modes of tiles 17 / 19 (3x3 conv stride 1 and 2, temporal conv + residual) against torch and bit-for-bit against the auto tile"""
    g = torch.Generator().manual_seed(Cin + H + 1)
    x = torch.randn(NB, Cin, H, Wd, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to(dtype)
    b = torch.randn(Cout, generator=g)
    wp = L.pack_conv3x3_weight(w)   # K index (c // 64, tap, c % 64): the kernel walks the nine taps of a 64-channel block back to back
    xr, wpd, bd = dev(ctx, nhwc_rows(x)), dev(ctx, wp), dev(ctx, b)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    out = ctx.gemm(xr, wpd, bias=bd, mode=L.A_CONV3X3, conv=(NB, H, Wd, Cin, H, Wd, 1, 1, 1), force_tile=sched_tile)
    assert_close(out, nhwc_rows(ref), dtype, what="conv3x3 s1 scheduled tile")
    assert torch.equal(out, ctx.gemm(xr, wpd, bias=bd, mode=L.A_CONV3X3, conv=(NB, H, Wd, Cin, H, Wd, 1, 1, 1)))
    ref = F.conv2d(x.float(), w.float(), b, stride=2, padding=(2, 1))
    Ho, Wo = ref.shape[2:]
    out = ctx.gemm(xr, wpd, bias=bd, mode=L.A_CONV3X3, conv=(NB, H, Wd, Cin, Ho, Wo, 2, 2, 1), force_tile=sched_tile)
    assert_close(out, nhwc_rows(ref), dtype, what="conv3x3 s2 scheduled tile")
    with pytest.raises(L.StarError):      # no nearest-x2 mode in this tile
        ctx.gemm(xr, wpd, bias=bd, mode=L.A_CONV3X3_UP, conv=(NB, H, Wd, Cin, 2 * H - 2, 2 * Wd, 1, 1, 1), force_tile=sched_tile)
    # temporal conv: NB frames of H x Wd, Cin channels
    if Cin == Cout:
        wt = (torch.randn(Cin, 3 * Cin, generator=g) / math.sqrt(3 * Cin)).to(dtype)
        a = nhwc_rows(x)
        ad, wtd = dev(ctx, a), dev(ctx, wt)
        o17 = ctx.gemm(ad, wtd, bias=bd, res=ad, mode=L.A_TCONV3, temporal=(NB, H * Wd, Cin), force_tile=sched_tile)
        assert torch.equal(o17, ctx.gemm(ad, wtd, bias=bd, res=ad, mode=L.A_TCONV3, temporal=(NB, H * Wd, Cin)))


@pytest.mark.parametrize("tile", [0, 30])
@pytest.mark.parametrize("M,N,geglu", [(300, 128, False), (257, 192, True), (520, 960, False), (64, 2560, True)])
def test_gemm_folded_layer_norm_rows(ctx, dtype, M, N, geglu, tile):
    """STAR_EPI_ROWAFF: out = a_m * (x W^T) + b_m * colsum[n] + bias[n] (a LayerNorm folded into the projection behind it,
    unet_v2v.py:448-450 + :151-155 / :500) on the tiled kernel (tile 0 = auto) and on the A-stationary K = 320 kernel of
    gemm_as.h (tile 30: a wave's rows of A live in registers, W streams through LDS), plain and with the GEGLU epilogue, ragged
    row tails."""
    g = torch.Generator().manual_seed(M + N)
    K = 320
    x = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g)
    rowab = torch.stack([torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)], dim=1).contiguous()
    colsum = W.float().sum(1).contiguous()
    acc = x.float() @ W.float().t()
    full = rowab[:, :1] * acc + rowab[:, 1:] * colsum[None] + b[None]
    if geglu:    # rows interleaved in 32-row (value, gate) blocks
        fv = full.reshape(M, N // 64, 2, 32)
        ref = (fv[:, :, 0] * F.gelu(fv[:, :, 1])).reshape(M, N // 2)
    else:
        ref = full
    out = ctx.gemm(dev(ctx, x), dev(ctx, W), bias=dev(ctx, b), geglu=geglu, rowab=dev(ctx, rowab), colsum=dev(ctx, colsum), force_tile=tile)
    assert_close(out, ref, dtype, scale=6.0, what=f"gemm rowaff tile {tile}")


def _header_enum(name):
    """value of an enumerator of include/star_hip.h, parsed from the header text (a C caller sees only these names)"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "star_hip.h")).read()
    m = re.search(r"\b%s\s*=\s*(\d+)" % name, txt)
    assert m, f"{name} is not defined in include/star_hip.h"
    return int(m.group(1))


@pytest.mark.parametrize("M,N", [(300, 128), (520, 960)])
def test_folded_layer_norm_through_the_header_names(ctx, dtype, M, N):
    """A C caller's view of the folded LayerNorm: star_layer_norm_rowab for the row statistics, then star_gemm with
    STAR_EPI_ROWAFF | STAR_EPI_BIAS and the rowab / colsum fields of star_gemm_desc -- using only names and values the header
    defines -- equals LayerNorm (unet_v2v.py:448-450) followed by the Linear behind it (:151-155)."""
    import ctypes
    assert _header_enum("STAR_EPI_ROWAFF") == L.EPI_ROWAFF == 32 and _header_enum("STAR_EPI_BIAS") == 1
    g = torch.Generator().manual_seed(M * 7 + N)
    K = 320
    x = (torch.randn(M, K, generator=g) * 1.5 + 0.3).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    gamma, beta, b = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1, torch.randn(N, generator=g)
    ref = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ W.float().t() + b
    Wf = (W.float() * gamma[None]).to(dtype)                  # W' = gamma o W (rounded as it will be multiplied)
    colsum = Wf.float().sum(1).contiguous()
    bias2 = (W.float() @ beta + b).contiguous()
    xd, Wd, cs, b2 = dev(ctx, x), dev(ctx, Wf), dev(ctx, colsum), dev(ctx, bias2)
    rowab = torch.empty(M, 2, dtype=torch.float32, device=ctx.torch_device)
    rc = ctx.lib.layer_norm_rowab(ctx.h, ctypes.c_void_p(xd.data_ptr()), K, ctypes.c_void_p(rowab.data_ptr()), M, K, 1e-5, 0, None, None, 0, 0)
    assert rc == 0, ctx.lib.last_error(ctx.h)
    out = torch.empty(M, N, dtype=dtype, device=ctx.torch_device)
    d = L.GemmDesc()
    d.A, d.W, d.C, d.bias, d.res = xd.data_ptr(), Wd.data_ptr(), out.data_ptr(), b2.data_ptr(), None
    d.M, d.N, d.K, d.lda, d.ldc, d.ldr = M, N, K, K, N, 0
    d.mode, d.stride, d.pad_t, d.pad_l, d.up_crop = 0, 1, 1, 1, 1
    d.epi = _header_enum("STAR_EPI_ROWAFF") | _header_enum("STAR_EPI_BIAS")
    d.rowab, d.colsum = rowab.data_ptr(), cs.data_ptr()
    rc = ctx.lib.gemm(ctx.h, ctypes.byref(d))
    assert rc == 0, ctx.lib.last_error(ctx.h)
    ctx.sync()
    assert_close(out, ref, dtype, scale=6.0, what="folded LayerNorm through the header names")
    # the statistics entry refuses what it cannot produce
    assert ctx.lib.layer_norm_rowab(ctx.h, ctypes.c_void_p(xd.data_ptr()), K, None, M, K, 1e-5, 0, None, None, 0, 0) != 0


def ref_attention(q, k, v, heads):
    B = q.shape[0]
    sp = lambda t: t.float().reshape(t.shape[0], -1, heads, 64).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(q), sp(k).expand(B, -1, -1, -1), sp(v).expand(B, -1, -1, -1))
    return o.transpose(1, 2).reshape(B, -1, heads * 64)


@pytest.mark.parametrize("variant", [9, 34, 40, 41])
@pytest.mark.parametrize("B,heads,Nq,Nk", [(2, 2, 300, 300), (1, 5, 80, 80), (3, 1, 257, 64), (1, 2, 64, 1), (9, 1, 33, 130), (1, 1, 130, 129)])
def test_flash_attention_self(ctx, dtype, B, heads, Nq, Nk, variant):
    """spatial self-attention (unet_v2v.py:472 -> :184) on a fused QKV buffer, ragged q/k tails."""
    ctx = need_variant(ctx, variant == 9)
    if variant == 31 and dtype != torch.float16:
        pytest.skip("packed row sums (v_pk_add_f16) exist for f16 only")
    g = torch.Generator().manual_seed(Nq + Nk)
    C = heads * 64
    N = max(Nq, Nk)
    qkv = torch.randn(B, N, 3 * C, generator=g).to(dtype)
    q, k, v = qkv[:, :Nq, :C], qkv[:, :Nk, C:2 * C], qkv[:, :Nk, 2 * C:]
    qkvd = dev(ctx, qkv)
    out = ctx.attention(qkvd[:, :Nq, :C], qkvd[:, :Nk, C:2 * C], qkvd[:, :Nk, 2 * C:], heads, variant=variant)
    assert_close(out, ref_attention(q, k, v, heads), dtype, what="flash self")


def test_flash_attention_round_toward_zero_pack(ctx, dtype):
    """bench variant 35 (attn5.h RTZ; round 6, measured and not adopted): the probabilities packed with v_cvt_pkrtz_f16_f32.  The row
    sum is taken from the same rounded probabilities the PV MFMA multiplies, so the truncation's common shift cancels: the result
    must sit within the product kernel's tolerance of the fp32 reference and close to the nearest-even kernel; it exists for f16 with
    long key ranges only and says so otherwise.  (Exercises prim.h: cvt_pkrtz -- on the emulator its software statement, incl. the
    underflow-to-zero and overflow-to-largest-finite cases the first draft got wrong.)"""
    ctx = need_variant(ctx, False)
    g = torch.Generator().manual_seed(35)
    B, heads, Nq, Nk = 2, 2, 130, 1100
    qkv = (torch.randn(B, Nk, 3 * heads * 64, generator=g) * 1.5).to(dtype)
    C = heads * 64
    qkvd = dev(ctx, qkv)
    args = (qkvd[:, :Nq, :C], qkvd[:, :, C:2 * C], qkvd[:, :, 2 * C:], heads)
    if dtype != torch.float16:
        with pytest.raises(L.StarError):
            ctx.attention(*args, variant=35)
        return
    out = ctx.attention(*args, variant=35)
    ref = ref_attention(qkv[:, :Nq, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], heads)
    assert_close(out, ref, dtype, what="flash, round-toward-zero pack")
    rne = ctx.attention(*args, variant=9)
    assert float((out.float() - rne.float()).abs().max()) <= 2.0 ** -9 * max(1.0, float(ref.abs().max()))
    with pytest.raises(L.StarError):     # short key ranges keep fp32 row sums: no packed variant there
        ctx.attention(qkvd[:, :Nq, :C], qkvd[:, :77, C:2 * C], qkvd[:, :77, 2 * C:], heads, variant=35)


@pytest.mark.parametrize("rows,n,lds,ldp", [
    (5, 1000, 1024, 1024),        # one-read form, the row in registers, masked tail inside the last valid chunk
    (3, 1024, 1024, 1024),        # no padding at all
    (4, 70, 128, 128),            # fewer chunks than threads; padding chunks that are masked entirely
    (2, 26352, 26368, 26368),     # the VAE's mid-block row at cfg2 (122 x 216 tokens)
    (2, 30001, 30016, 30016),     # longer than 256 x 13 chunks: online maximum / sum, second read for the store
    (3, 77, 77, 80),              # lds not a multiple of 4: the three-read kernel
    (3, 100, 104, 100),           # ldp not a multiple of 8: the three-read kernel
])
def test_softmax_rows(ctx, dtype, rows, n, lds, ldp):
    """the logits pass of the VAE's one-head d = 512 attention (diffusers Attention; vae.cpp: attn): P = softmax(S * scale) in the 16-bit
    type with the padding columns written as zeros, by the one-read vector kernel where the layout allows and the scalar kernel elsewhere."""
    g = torch.Generator().manual_seed(rows * 131 + n)
    s = torch.randn(rows, lds, generator=g) * 30.0
    s[0, min(n - 1, 17)] = 500.0            # a dominant logit
    if rows > 1:
        s[1, :n] = -40.0                    # a flat row
    s[:, n:] = float("nan")                 # whatever lies beyond n must never be read into the result
    scale = 1.0 / math.sqrt(512.0)
    out = ctx.softmax_rows(dev(ctx, s), n, scale, ldp=ldp).float().cpu()
    ref = torch.zeros(rows, ldp)
    ref[:, :n] = torch.softmax(s[:, :n].double() * scale, dim=-1).float()
    assert torch.isfinite(out).all()
    assert float(out[:, n:].abs().max()) == 0.0 if ldp > n else True
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    assert float((out - ref).abs().max()) <= 2.0 * eps * float(ref.max()), float((out - ref).abs().max())
    assert float((out.sum(dim=1) - 1.0).abs().max()) <= 6e-3 if dtype == torch.float16 else 4e-2


def test_flash_attention_cross_77(ctx, dtype):
    """cross-attention to the 77 text tokens, K/V shared by all frames (unet_v2v.py:476)."""
    g = torch.Generator().manual_seed(77)
    B, heads, Nq = 3, 2, 200
    q = torch.randn(B, Nq, 128, generator=g).to(dtype)
    kv = torch.randn(1, 77, 256, generator=g).to(dtype)
    kvd = dev(ctx, kv)
    out = ctx.attention(dev(ctx, q), kvd[..., :128], kvd[..., 128:], heads)
    assert_close(out, ref_attention(q, kv[..., :128], kv[..., 128:], heads), dtype, what="flash cross")


def test_product_library_rejects_bench_variants(ctx, dtype):
    """ablation probes compute wrong answers by construction and losing A/B kernels are dead weight: the product ABI must
    refuse their ids (round-1 review: a caller passing the wrong integer silently got garbage)."""
    if ctx.lib.has_bench_variants:
        pytest.skip("this library is a bench / emulator build")
    q = torch.randn(1, 64, 192).to(dtype).to(ctx.torch_device)
    for variant in (2, 8, 11, 12, 16, 17, 40, 41):
        with pytest.raises(L.StarError):
            ctx.attention(q[..., :64], q[..., 64:128], q[..., 128:], 1, variant=variant)
    A = torch.randn(256, 64).to(dtype).to(ctx.torch_device)
    for tile in (5, 7, 11, 13, 16):
        with pytest.raises(L.StarError):
            ctx.gemm(A, A, force_tile=tile)
    ctx.attention(q[..., :64], q[..., 64:128], q[..., 128:], 1, variant=0)   # 0 = default = the product kernel


# the measured-and-lost A/B kernels of rounds 1-2 (bench build / emulator only): one smoke case each -- they do not ship, their full
# parity matrix ran in rounds 1-2 (profiles/r02_pytest_gpu_v1.txt)
LOSING_VARIANTS = [2, 3, 6, 7, 8, 10, 15, 21, 22, 23, 24, 27, 30, 31, 32, 33]


@pytest.mark.parametrize("variant", [9, 40, 41] + LOSING_VARIANTS)
def test_flash_attention_variants_agree(ctx, dtype, variant):
    """all kernel variants (baseline / v2 / v3 with the augmented-k running max) against the fp32 reference, incl. a
    ragged key tail, strongly negative logits in tile 0 and a late spike that forces the rescale branch."""
    ctx = need_variant(ctx, variant == 9)
    if variant == 31 and dtype != torch.float16:
        pytest.skip("packed row sums (v_pk_add_f16) exist for f16 only")
    g = torch.Generator().manual_seed(21)
    B, heads, Nq, Nk = 2, 2, 200, 333
    q = torch.randn(B, Nq, 128, generator=g)
    k = torch.randn(B, Nk, 128, generator=g)
    v = torch.randn(B, Nk, 128, generator=g)
    k[:, :64] = -q[:, :1].mean(1, keepdim=True) * 2.0     # first tile mostly anti-aligned with many queries
    k[:, 300, :64] = q[:, 17, :64] * 5.0                  # late spike for one query, head 0
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    out = ctx.attention(dev(ctx, q), dev(ctx, k), dev(ctx, v), heads, variant=variant)
    assert_close(out, ref_attention(q, k, v, heads), dtype, scale=6.0, what=f"flash variant {variant}")


@pytest.mark.parametrize("variant", [9, 40, 41])
def test_flash_attention_forced_rescale(ctx, dtype, variant):
    """a key spike late in the sequence forces the online-softmax rescale branch with a large max jump."""
    ctx = need_variant(ctx, variant == 9)
    if variant == 31 and dtype != torch.float16:
        pytest.skip("packed row sums (v_pk_add_f16) exist for f16 only")
    g = torch.Generator().manual_seed(11)
    B, heads, N = 1, 1, 400
    q = torch.randn(B, N, 64, generator=g)
    k = torch.randn(B, N, 64, generator=g)
    v = torch.randn(B, N, 64, generator=g)
    k[:, 333] = q[:, 7] * 4.0     # q.k ~ 4*64 = 256 against O(8) elsewhere
    k[:, 2] = q[:, 100] * 3.0     # and an early spike that later tiles must not disturb
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    out = ctx.attention(dev(ctx, q), dev(ctx, k), dev(ctx, v), heads, variant=variant)
    assert_close(out, ref_attention(q, k, v, heads), dtype, what="flash rescale")


@pytest.mark.parametrize("variant", [9, 40, 41])
def test_flash_attention_growing_max(ctx, dtype, variant):
    """scores that keep growing along the key axis (every tile moves the maximum by several binades, some by more than
    the fp16 exponent range) and a first tile far below everything that follows: the lazy-max variants must take their
    recompute path tile after tile and still match."""
    ctx = need_variant(ctx, variant == 9)
    if variant == 31 and dtype != torch.float16:
        pytest.skip("packed row sums (v_pk_add_f16) exist for f16 only")
    g = torch.Generator().manual_seed(31)
    B, heads, N = 1, 2, 520
    q = torch.randn(B, N, 128, generator=g)
    k = torch.randn(B, N, 128, generator=g) * 0.2
    v = torch.randn(B, N, 128, generator=g)
    ramp = torch.linspace(-3.0, 6.0, N)                      # key j adds ramp[j] * |q|^2 / 8 to the logit
    k = k + q.mean(1, keepdim=True) * 0 + (ramp[None, :, None] * torch.nn.functional.normalize(q[:, :1], dim=-1) * 3.0)
    k[:, 450, :64] = q[:, 5, :64] * 6.0                      # one jump of > 16 binades for a single row
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    out = ctx.attention(dev(ctx, q), dev(ctx, k), dev(ctx, v), heads, variant=variant)
    assert_close(out, ref_attention(q, k, v, heads), dtype, scale=6.0, what=f"flash growing max v{variant}")


@pytest.mark.parametrize("variant", [9, 40, 41])
@pytest.mark.parametrize("Nq,Nk", [(150, 1100), (70, 1024), (390, 1217)])
def test_flash_attention_long_key_range(ctx, dtype, variant, Nq, Nk):
    """key ranges >= 1024 take the packed 16-bit row sums (f16) and, in the one-wave-per-SIMD kernels (40 / 41), many trips of the
    query-block pipeline incl. odd / even tile counts, a ragged last tile and a late spike that moves the running maximum."""
    ctx = need_variant(ctx, variant == 9)
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    q = torch.randn(1, Nq, 64, generator=g)
    k = torch.randn(1, Nk, 64, generator=g)
    v = torch.randn(1, Nk, 64, generator=g)
    k[:, Nk - 100] = q[:, 3] * 3.0
    k[:, 700] = q[:, Nq - 1] * 2.0
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    out = ctx.attention(dev(ctx, q), dev(ctx, k), dev(ctx, v), 1, variant=variant)
    assert_close(out, ref_attention(q, k, v, 1), dtype, what=f"flash long keys v{variant}")


@pytest.mark.parametrize("Fr,HW,heads", [(5, 7, 2), (32, 9, 1), (40, 5, 2), (16, 130, 5), (1, 4, 1), (64, 3, 1), (80, 5, 2), (97, 3, 1), (128, 2, 1)])
def test_temporal_attention(ctx, dtype, Fr, HW, heads):
    """attention over frames per pixel (unet_v2v.py:483-489), tokens stay in [F*HW, C] order."""
    g = torch.Generator().manual_seed(Fr * 31 + HW)
    C = heads * 64
    qkv = torch.randn(Fr * HW, 3 * C, generator=g).to(dtype)
    qkvd = dev(ctx, qkv)
    out = ctx.temporal_attention(qkvd[:, :C], qkvd[:, C:2 * C], qkvd[:, 2 * C:], Fr, HW, heads)
    tr = lambda x: x.float().reshape(Fr, HW, heads, 64).permute(1, 2, 0, 3)
    ref = F.scaled_dot_product_attention(tr(qkv[:, :C]), tr(qkv[:, C:2 * C]), tr(qkv[:, 2 * C:]))
    ref = ref.permute(2, 0, 1, 3).reshape(Fr * HW, C)
    assert_close(out, ref, dtype, what="temporal attn")


@pytest.mark.parametrize("Fr,HW", [(32, 11), (16, 8), (5, 3), (32, 64), (1, 9)])
def test_temporal_projection_and_attention_fused(ctx, dtype, Fr, HW):
    """gemm_tq.h: the q | k | v projection of a temporal attention (LayerNorm folded: STAR_EPI_ROWAFF operands) and the attention over
    the frame axis in ONE kernel at the level-0 width (C = 320, 5 heads; unet_v2v.py:479-489) -- q | k | v stay in the wave's staging
    block.  Bit for bit the two-kernel path (star_gemm with the folded LayerNorm, then star_temporal_attn_fwd), and close to an fp32
    statement of LayerNorm -> Linear -> attention; ragged pixel counts (a workgroup covers 8 pixels), fewer than 32 frames (masked
    keys, unstored queries)."""
    if Fr * HW > 400 and ctx.lib.is_hostemu:
        pytest.skip("hardware-only size (the emulator runs 15 projection tiles of 320 k-elements per workgroup)")
    g = torch.Generator().manual_seed(Fr * 37 + HW)
    C, heads = 320, 5
    M = Fr * HW
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).to(dtype)
    Wq, Wk, Wv = ((torch.randn(C, C, generator=g) / math.sqrt(C)) for _ in range(3))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    Wf = (torch.cat([Wq, Wk, Wv]) * gamma[None]).to(dtype)                      # W' = gamma o W, rows q | k | v
    cb = (torch.cat([Wq, Wk, Wv]) @ beta).contiguous()                           # c_n = sum_k beta_k W[n][k] (no bias in to_q / to_k / to_v)
    cs = Wf.float().sum(1).contiguous()
    perm = torch.cat([torch.arange(64) + part * C + h * 64 for h in range(heads) for part in range(3)])   # (q_h, k_h, v_h) per head
    xd = dev(ctx, x)
    rowab = ctx.layer_norm_rowab(xd, eps=1e-5)
    # two kernels: the A-stationary K = 320 kernel (what the level-0 layers run: its row-affine epilogue and the tiled kernel's differ
    # by an fp32 rounding of the affine on hardware, 1 ulp of the 16-bit output), then the attention over frames
    qkv = ctx.gemm(xd, dev(ctx, Wf), bias=dev(ctx, cb), rowab=rowab, colsum=dev(ctx, cs), force_tile=30)
    two = ctx.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], Fr, HW, heads)
    # one kernel
    one = ctx.temporal_qkv_attn(xd, dev(ctx, Wf[perm].contiguous()), dev(ctx, cb[perm].contiguous()), dev(ctx, cs[perm].contiguous()), rowab, Fr, HW)
    assert one.shape == (M, C) and torch.equal(one, two)
    # fp32 statement
    h = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    tr = lambda t: t.reshape(Fr, HW, heads, 64).permute(1, 2, 0, 3)
    ref = F.scaled_dot_product_attention(tr(h @ Wq.t()), tr(h @ Wk.t()), tr(h @ Wv.t())).permute(2, 0, 1, 3).reshape(M, C)
    assert_close(one, ref, dtype, scale=12.0, what="fused temporal projection + attention")
    with pytest.raises(L.StarError):     # more than 32 frames per chunk keep the two kernels
        ctx.temporal_qkv_attn(xd, dev(ctx, Wf[perm].contiguous()), dev(ctx, cb[perm].contiguous()), dev(ctx, cs[perm].contiguous()), rowab, 33, HW)


@pytest.mark.parametrize("C", [320, 128, 2560, 960, 512, 1920])
@pytest.mark.parametrize("silu", [False, True])
def test_group_norm(ctx, dtype, C, silu):
    """GroupNorm(32): per-frame statistics (4-D call sites) and whole-chunk statistics (5-D call sites)."""
    g = torch.Generator().manual_seed(C)
    NF, HW = 3, 50
    x = (torch.randn(NF * HW, C, generator=g) * 2 + 0.5).to(dtype)
    gam, bet = torch.randn(C, generator=g), torch.randn(C, generator=g)
    for rps, eps in ((HW, 1e-5), (NF * HW, 1e-6)):
        out = ctx.group_norm(dev(ctx, x), dev(ctx, gam), dev(ctx, bet), rps, eps=eps, silu=silu)
        xr = x.float().reshape(-1, rps, C).permute(0, 2, 1)
        ref = F.group_norm(xr, 32, gam, bet, eps)
        if silu:
            ref = F.silu(ref)
        assert_close(out, ref.permute(0, 2, 1).reshape(-1, C), dtype, what=f"gn rps={rps}")


def test_group_norm_large_mean(ctx, dtype):
    """fp64 sum / sum-of-squares accumulation must survive |mean| >> std."""
    g = torch.Generator().manual_seed(9)
    C, rows = 64, 4096
    x = (torch.randn(rows, C, generator=g) * 0.05 + 6.0).to(dtype)
    gam, bet = torch.ones(C), torch.zeros(C)
    out = ctx.group_norm(dev(ctx, x), dev(ctx, gam), dev(ctx, bet), rows, eps=1e-5)
    ref = F.group_norm(x.float().t().unsqueeze(0), 32, gam, bet, 1e-5)[0].t()
    assert_close(out, ref, dtype, scale=8.0, what="gn large mean")


@pytest.mark.parametrize("C", [320, 512, 640, 1280, 2560, 3072, 5120])   # 3072 = CogVideoX-5B hidden (final LayerNorms of modules/dit.py)
def test_layer_norm_and_liem_gates(ctx, dtype, C):
    g = torch.Generator().manual_seed(C + 1)
    Fr, H, W = 2, 6, 5
    rows = Fr * H * W
    x = torch.randn(rows, C, generator=g).to(dtype)
    gam, bet = torch.randn(C, generator=g), torch.randn(C, generator=g)
    xd, gd, bd = dev(ctx, x), dev(ctx, gam), dev(ctx, bet)
    xf = x.float()
    assert_close(ctx.layer_norm(xd, gd, bd), F.layer_norm(xf, (C,), gam, bet), dtype, what="ln")
    # TemporalLocalAttention (unet_v2v.py:396-411): sigmoid(Linear(2,1)([max_c, mean_c])) * x
    w2 = torch.randn(2, generator=g)
    gate = torch.sigmoid(w2[0] * xf.max(-1, keepdim=True)[0] + w2[1] * xf.mean(-1, keepdim=True))
    out = ctx.layer_norm(xd, gd, bd, mode=L.LN_GATE_LINEAR, gate_w=dev(ctx, w2))
    assert_close(out, F.layer_norm(gate * xf, (C,), gam, bet), dtype, what="ln temporal gate")
    # SpatialAttention (unet_v2v.py:380-394): sigmoid(conv7x7([max_c, mean_c])) * x
    w7 = torch.randn(1, 2, 7, 7, generator=g) * 0.3
    maps = ctx.layer_norm(xd, None, None, mode=L.LN_STATS_ONLY)
    out = ctx.layer_norm(xd, gd, bd, mode=L.LN_GATE_MAP, gate_w=dev(ctx, w7.reshape(-1).contiguous()), maps=maps, H=H, W=W)
    xi = xf.reshape(Fr, H, W, C).permute(0, 3, 1, 2)
    wmap = torch.cat([xi.max(1, keepdim=True)[0], xi.mean(1, keepdim=True)], 1)
    gate = torch.sigmoid(F.conv2d(wmap, w7, padding=3))
    ref = F.layer_norm((gate * xi).permute(0, 2, 3, 1).reshape(rows, C), (C,), gam, bet)
    assert_close(out, ref, dtype, what="ln spatial gate")


def test_plumbing_kernels(ctx, dtype):
    g = torch.Generator().manual_seed(2)
    a = torch.randn(40, 64, generator=g).to(dtype)
    b = torch.randn(40, 32, generator=g).to(dtype)
    c = torch.randn(40, 32, generator=g).to(dtype)
    out = ctx.concat_add(dev(ctx, a), dev(ctx, b), dev(ctx, c))
    assert_close(out, torch.cat([a.float(), b.float() + c.float()], 1), dtype, what="concat_add")
    out = ctx.concat_add(dev(ctx, a), dev(ctx, b))
    assert_close(out, torch.cat([a.float(), b.float()], 1), dtype, what="concat")
    assert_close(ctx.add(dev(ctx, a), dev(ctx, a)), 2 * a.float(), dtype, what="add")
    # stem conv 4 -> 320 through im2col rows (unet_v2v.py:1353)
    lat = torch.randn(1, 4, 3, 6, 8, generator=g)
    w = torch.randn(320, 4, 3, 3, generator=g) * 0.2
    cols = ctx.stem_im2col(dev(ctx, lat))
    wp = torch.zeros(320, 64)
    wp[:, :36] = w.permute(0, 2, 3, 1).reshape(320, 36)
    o = ctx.gemm(cols, dev(ctx, wp.to(dtype)), out_f32=True)
    ref = F.conv2d(lat[0].permute(1, 0, 2, 3).to(dtype).float(), w.to(dtype).float(), padding=1).permute(0, 2, 3, 1).reshape(-1, 320)
    assert_close(o, ref, dtype, what="stem conv")
    r2l = ctx.rows_to_latent(o[:, :8].contiguous(), 4, 3, 6, 8)
    assert torch.equal(r2l[0].cpu(), o[:, :4].reshape(3, 6, 8, 4).permute(3, 0, 1, 2).cpu())
    xv = torch.randn(320, generator=g)
    Wv = (torch.randn(1280, 320, generator=g) * 0.1).to(dtype)
    bv = torch.randn(1280, generator=g)
    y = ctx.gemv(dev(ctx, xv), dev(ctx, Wv), dev(ctx, bv), silu_in=True, silu_out=True)
    ref = F.silu(F.silu(xv) @ Wv.float().T + bv)
    assert float((y.cpu() - ref).abs().max()) < 1e-4
    assert torch.equal(ctx.cast(dev(ctx, xv)).cpu(), xv.to(dtype))


def _pair_stats(out, M, N):
    """(sum, sum of squares) of a [M, N] tensor per 32-row slot and channel pair, in float64 -> [ceil(M/32), N/2, 2]"""
    o = out.double().cpu()
    pad = (-M) % 32
    if pad:
        o = torch.cat([o, torch.zeros(pad, N, dtype=torch.float64)])
    o = o.reshape(-1, 32, N // 2, 2)
    return torch.stack([o.sum(dim=(1, 3)), (o * o).sum(dim=(1, 3))], dim=-1)


GN_EPI_CASES = [   # mode, tile, (frames, H, W), Cin, Cout, residual
    ("plain", 2, (3, 11, 9), 192, 320, True),      # proj_out + x_in (SpatialTransformer -> TemporalTransformer.norm), 256 x 320 tile: 3 rows x 20 chunk columns per pass
    ("plain", 3, (3, 11, 9), 128, 192, True),      # 128 x 128 tile (the tail-split remainder), ragged last column tile
    ("plain", 17, (2, 17, 16), 128, 256, False),   # scheduled one-wave-per-SIMD tile
    ("conv", 2, (3, 11, 9), 64, 320, False),       # ResBlock conv1 -> out_layers.0 (per-frame statistics: 99 rows per frame cut the 32-row slots)
    ("conv", 3, (2, 10, 8), 64, 128, True),        # conv2 + skip
    ("conv", 17, (2, 17, 16), 64, 256, False),
    ("tconv", 2, (5, 7, 9), 320, 320, True),       # TemporalConvBlock_v2 conv4 + identity
    ("tconv", 3, (4, 6, 8), 64, 64, False),
    ("plain", 3, (4, 64, 50), 64, 64, True),       # 12 800 rows: the whole-chunk finalize is split into parts along the slots (two-stage reduction)
    # round 6: the power-of-two tiles that run the VAE's 128 / 256 / 512-wide layers (vae.cpp)
    ("conv", 4, (2, 11, 9), 64, 128, True),        # 256 x 128 tile: ResnetBlock2D conv2 + shortcut at the full-resolution levels
    ("conv", 1, (2, 11, 9), 64, 256, False),       # 256 x 256 tile
    ("tconv", 1, (3, 10, 8), 128, 512, True),      # TemporalResnetBlock conv2 + spatial branch
    ("plain", 4, (3, 11, 9), 64, 128, False),      # conv_in as an im2col GEMM (K = 64)
    # round 6: the scheduled 256 x 320 tile (128 x 160 per wave: 20 chunk columns per wave as tile 2)
    ("conv", 19, (3, 11, 9), 64, 320, False),
    ("plain", 19, (2, 17, 16), 128, 640, False),
    ("tconv", 19, (5, 7, 9), 320, 320, False),
    ("conv", 19, (2, 10, 8), 128, 320, True),      # ... and WITH the residual (ResBlock conv2 + skip): its fifth block column is in architectural registers
    ("tconv", 19, (4, 9, 8), 320, 640, True),
]


@pytest.mark.parametrize("mode,tile,geom,Cin,Cout,with_res", GN_EPI_CASES)
def test_group_norm_statistics_in_the_producer_epilogue(ctx, dtype, mode, tile, geom, Cin, Cout, with_res):
    """star_gemm_gn + star_group_norm_from_partials (gemm.h EPIF bit 4, norm.h gn_finalize_fused_kernel; unet_v2v.py:609-640,1209-1220):
    the layer's output is bit-identical to star_gemm's; the partials are the sums of the STORED outputs per 32-row slot and channel
    pair; the GroupNorm finalized from them (per-frame statistics whose boundaries cut the slots, and whole-chunk statistics) matches
    torch's on the stored tensor as closely as the stand-alone kernel does."""
    Fr, H, Wd = geom
    g = torch.Generator().manual_seed(Cin + Cout + tile)
    M = Fr * H * Wd
    res = dev(ctx, (torch.randn(M, Cout, generator=g) * 0.7 + 0.3).to(dtype)) if with_res else None
    b = dev(ctx, torch.randn(Cout, generator=g))
    if mode == "plain":
        a = dev(ctx, torch.randn(M, Cin, generator=g).to(dtype))
        w = dev(ctx, (torch.randn(Cout, Cin, generator=g) / math.sqrt(Cin)).to(dtype))
        kw = dict(mode=L.A_PLAIN)
    elif mode == "conv":
        x = torch.randn(Fr, Cin, H, Wd, generator=g).to(dtype)
        a = dev(ctx, nhwc_rows(x))
        w = dev(ctx, (torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(dtype))
        kw = dict(mode=L.A_CONV3X3, conv=(Fr, H, Wd, Cin, H, Wd, 1, 1, 1))
    else:
        a = dev(ctx, torch.randn(M, Cin, generator=g).to(dtype))
        w = dev(ctx, (torch.randn(Cout, 3 * Cin, generator=g) / math.sqrt(3 * Cin)).to(dtype))
        kw = dict(mode=L.A_TCONV3, temporal=(Fr, H * Wd, Cin))
    plain = ctx.gemm(a, w, bias=b, res=res, force_tile=tile, **kw)
    out, part = ctx.gemm(a, w, bias=b, res=res, force_tile=tile, gn_partial=True, **kw)
    assert part is not None, "the tile has the statistics flavour"
    assert torch.equal(out, plain)
    want = _pair_stats(out, M, Cout)
    got = part.double().cpu()
    scale = want[..., 1].abs().max().item() + 1.0
    assert (got - want).abs().max().item() <= 2e-5 * scale, (got - want).abs().max().item()
    if Cout % 64:
        return   # GroupNorm(32) needs an even number of channels per group for the pair partials
    gam, bet = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g)
    for rps, eps, silu in ((H * Wd, 1e-5, True), (M, 1e-6, False)):
        y = ctx.group_norm_from_partials(out, part, dev(ctx, gam), dev(ctx, bet), rps, eps=eps, silu=silu)
        y0 = ctx.group_norm(out, dev(ctx, gam), dev(ctx, bet), rps, eps=eps, silu=silu)
        xr = out.float().cpu().reshape(-1, rps, Cout).permute(0, 2, 1)
        ref = F.group_norm(xr, 32, gam, bet, eps)
        if silu:
            ref = F.silu(ref)
        ref = ref.permute(0, 2, 1).reshape(-1, Cout)
        assert_close(y, ref, dtype, what=f"gn from partials rps={rps}")
        assert (y.float() - y0.float()).abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 4e-3) * (1 + ref.abs().max().item())
    # a tile without the flavour: the output is still computed, no partials
    out4, part4 = ctx.gemm(a, w, bias=b, res=res, force_tile=9, gn_partial=True, **kw)
    assert part4 is None and out4.shape == out.shape


def test_group_norm_statistics_across_a_tail_split(ctx, dtype):
    """with the automatic tile choice a poorly filled last round goes to a second launch of 128 x 128 tiles (gemm_impl.h, tail
    split); BOTH launches write the same partial buffer.  4500 rows x 320 columns on 17 assumed CUs: 18 tiles of 256 x 320 = 17 in the
    main launch (4352 rows = 136 whole slots) + 148 rows in the remainder.  The output is bit-identical to the forced single launch and
    the partials are the pair sums of all 4500 stored rows."""
    g = torch.Generator().manual_seed(4500)
    M, K, N = 4500, 64, 320
    a = dev(ctx, torch.randn(M, K, generator=g).to(dtype))
    w = dev(ctx, (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype))
    b = dev(ctx, torch.randn(N, generator=g))
    res = dev(ctx, (torch.randn(M, N, generator=g) * 0.7 + 0.3).to(dtype))
    plain = ctx.gemm(a, w, bias=b, res=res, force_tile=2)
    n0 = ctx.lib.gemm_split_count(ctx.h)
    out, part = ctx.gemm(a, w, bias=b, res=res, force_tile=1017, gn_partial=True)
    assert ctx.lib.gemm_split_count(ctx.h) == n0 + 1, "the shape was chosen to trigger the tail split"
    assert part is not None and torch.equal(out, plain)
    want = _pair_stats(out, M, N)
    got = part.double().cpu()
    scale = want[..., 1].abs().max().item() + 1.0
    assert (got - want).abs().max().item() <= 2e-5 * scale, (got - want).abs().max().item()
    gam, bet = torch.randn(N, generator=g), torch.randn(N, generator=g)
    y = ctx.group_norm_from_partials(out, part, dev(ctx, gam), dev(ctx, bet), M // 4, eps=1e-5, silu=True)
    y0 = ctx.group_norm(out, dev(ctx, gam), dev(ctx, bet), M // 4, eps=1e-5, silu=True)
    assert (y.float() - y0.float()).abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 4e-3) * (1 + y0.float().abs().max().item())


def test_group_norm_statistics_with_the_frame_interleaved_walk(ctx, dtype):
    """a temporal conv whose frames are at least one tile row (HW >= 256) walks its tile rows frame-interleaved (gemm.h: t_walk);
    the statistics epilogue must address the slots of the tile it really computes.  3 frames of 16 x 17 = 272 pixels, 256 x 320 tile."""
    g = torch.Generator().manual_seed(272)
    Fr, HW, Cin, Cout = 3, 272, 64, 320
    M = Fr * HW
    a = dev(ctx, torch.randn(M, Cin, generator=g).to(dtype))
    w = dev(ctx, (torch.randn(Cout, 3 * Cin, generator=g) / math.sqrt(3 * Cin)).to(dtype))
    b = dev(ctx, torch.randn(Cout, generator=g))
    kw = dict(mode=L.A_TCONV3, temporal=(Fr, HW, Cin))
    ref3 = ctx.gemm(a, w, bias=b, force_tile=3, **kw)            # 128 x 128 tile: HW >= 128, walked frame-interleaved as well
    plain = ctx.gemm(a, w, bias=b, force_tile=2, **kw)
    out, part = ctx.gemm(a, w, bias=b, force_tile=2, gn_partial=True, **kw)
    assert part is not None and torch.equal(out, plain) and torch.equal(out, ref3)
    # against the conv itself: y[f] = sum_t W_t x[f + t - 1]
    x = a.float().cpu().reshape(Fr, HW, Cin)
    xp = torch.cat([torch.zeros(1, HW, Cin), x, torch.zeros(1, HW, Cin)])
    wt = w.float().cpu().reshape(Cout, 3, Cin)
    ref = sum(xp[t:t + Fr] @ wt[:, t].T for t in range(3)) + b.float().cpu()
    assert_close(out, ref.reshape(M, Cout), dtype, what="tconv (frame-interleaved walk)")
    want = _pair_stats(out, M, Cout)
    got = part.double().cpu()
    scale = want[..., 1].abs().max().item() + 1.0
    assert (got - want).abs().max().item() <= 2e-5 * scale, (got - want).abs().max().item()


ROWSTAT_CASES = [   # tile, (frames, H, W), K, N, residual
    (2, (2, 11, 9), 192, 320, True),     # out-projection + residual at level-0 width: one 320-column tile, two wave columns of 160 = 2 parts
    (2, (2, 11, 9), 128, 640, False),    # proj_in at level-1 width: 4 parts
    (3, (3, 10, 8), 64, 192, True),      # 128 x 128 tile, ragged last column tile (the second wave column of the last tile is empty)
    (3, (2, 9, 7), 64, 128, False),
]


@pytest.mark.parametrize("tile,geom,K,N,with_res", ROWSTAT_CASES)
def test_layer_norm_row_statistics_in_the_producer_epilogue(ctx, dtype, tile, geom, K, N, with_res):
    """star_gemm_rowstats + star_layer_norm_rowab_from_partials (gemm.h EPIF bit 5, norm.h ln_from_partials_kernel; unet_v2v.py:448-450,
    466-490): the layer's output is bit-identical to star_gemm's; the per-row records are (sum, sum of squares, max) of the STORED outputs
    over each column part; the row coefficients / LIEM maps derived from them match star_layer_norm_rowab / star_layer_norm (mode 3) on the
    stored tensor in all four modes -- also on rows whose mean is ten times their spread."""
    Fr, H, Wd = geom
    g = torch.Generator().manual_seed(K + N + tile)
    M = Fr * H * Wd
    a = dev(ctx, torch.randn(M, K, generator=g).to(dtype))
    w = dev(ctx, (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype))
    bias = torch.randn(N, generator=g)
    bias[: N // 2] += 6.0                                   # a large common offset: |row mean| >> spread for E[x^2] - mean^2
    b = dev(ctx, bias)
    res = dev(ctx, (torch.randn(M, N, generator=g) * 0.5).to(dtype)) if with_res else None
    plain = ctx.gemm(a, w, bias=b, res=res, force_tile=tile)
    out, part = ctx.gemm(a, w, bias=b, res=res, force_tile=tile, row_stats=True)
    assert part is not None and torch.equal(out, plain)
    o = out.double().cpu()
    p = part.double().cpu()
    assert (p[..., 0].sum(1) - o.sum(1)).abs().max().item() <= 1e-5 * (o.abs().sum(1).max().item() + 1)
    assert (p[..., 1].sum(1) - (o * o).sum(1)).abs().max().item() <= 2e-5 * ((o * o).sum(1).max().item() + 1)
    assert torch.equal(p[..., 2].max(1).values, o.max(1).values)
    # the consumers: maps, plain, linear gate, 7x7-map gate
    maps = ctx.layer_norm_rowab_from_partials(part, N, mode=L.LN_STATS_ONLY)
    maps0 = torch.empty(M, 2, dtype=torch.float32, device=ctx.torch_device)
    ctx.layer_norm(out, dev(ctx, torch.ones(N)), dev(ctx, torch.zeros(N)), mode=L.LN_STATS_ONLY, maps=maps0)
    assert (maps.cpu() - maps0.cpu()).abs().max().item() <= 1e-4
    w2 = dev(ctx, torch.randn(2, generator=g))
    w98 = dev(ctx, torch.randn(98, generator=g) * 0.2)
    for mode, gw, mp in ((L.LN_PLAIN, None, None), (L.LN_GATE_LINEAR, w2, None), (L.LN_GATE_MAP, w98, maps0)):
        ab = ctx.layer_norm_rowab_from_partials(part, N, mode=mode, gate_w=gw, maps=mp, H=H, W=Wd).cpu()
        ab0 = ctx.layer_norm_rowab(out, mode=mode, gate_w=gw, maps=mp, H=H, W=Wd).cpu()
        rel = ((ab - ab0).abs() / (ab0.abs() + 1e-3)).max().item()
        assert rel <= 2e-3, (mode, rel)
    out1, part1 = ctx.gemm(a, w, bias=b, res=res, force_tile=1, row_stats=True)   # a tile without the flavour
    assert part1 is None and out1.shape == out.shape
