"""Randomised shape sweep of the round-4 kernels on the SIMT emulator build (tools/hostemu/libstar_emu.so; no GPU needed):
  * tile 18 (gemm_p.h, persistent) with random (M, N, K), resident-workgroup counts, epilogue flavours  -- bit for bit against tile 1;
  * the tail split of the big tiles (gemm_impl.h: m_off / m_end)                                         -- bit for bit against tile 1;
  * the fused temporal projection + attention (gemm_tq.h) with random (frames, pixels)                    -- bit for bit against the
    A-stationary GEMM + star_temporal_attn_fwd;
  * gathered modes (3x3 conv, stride-2 Downsample, temporal conv) under the tail split                    -- bit for bit, unsplit tile 1;
  * the A-stationary K = 320 kernel (gemm_as.h)                                                           -- <= 1 ulp of tile 1;
  * flash_attn_v5_kernel / temporal_attn_kernel on ragged lengths, shared K / V, strided views            -- fp32 softmax, <= 4 ulp;
  * round 5: the statistics flavours of the epilogue (gemm.h EPIF 16 / 17 / 32 / 33: star_gemm_gn, star_gemm_rowstats) with random
    (frames, pixels, widths, tiles, residual)  -- output bit for bit against star_gemm, partials against float64 sums of the stored
    outputs, GroupNorm / LayerNorm coefficients from the partials against the stand-alone kernels.
The parametrised cases of tests/test_kernels.py are hand-picked edges; this draws the shapes.  `python tools/fuzz_emu.py --cases 40
--seed 1 > profiles/rNN_fuzz_emu.txt`.  Test tooling: the product never loads the emulator."""
import argparse
import math
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from star_amd import lib as L  # noqa: E402


def dev(ctx, t):
    return t.to(ctx.torch_device).contiguous()


def fuzz_persist(ctx, dtype, rng, budget):
    M = rng.choice([rng.randint(1, 64), rng.randint(65, 520), rng.randint(521, 900)])
    N = 8 * rng.randint(1, 80)
    K = 64 * rng.randint(1, 5)
    while M * N * K > budget:
        M = max(1, M // 2)
    wgs = rng.choice([0, 1, 2, 3, 5, 8])
    ft = 2000 + wgs if wgs else 18
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    b = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g).to(dtype)
    rowab = torch.stack([torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)], dim=1).contiguous()
    colsum = W.float().sum(1).contiguous()
    Ad, Wd, bd, Rd = dev(ctx, A), dev(ctx, W), dev(ctx, b), dev(ctx, R)
    flavours = [dict(), dict(bias=bd), dict(bias=bd, res=Rd), dict(res=Rd), dict(bias=bd, rowab=dev(ctx, rowab), colsum=dev(ctx, colsum))]
    if N % 64 == 0:
        flavours += [dict(bias=bd, geglu=True), dict(geglu=True), dict(bias=bd, geglu=True, rowab=dev(ctx, rowab), colsum=dev(ctx, colsum))]
    kw = rng.choice(flavours)
    out = ctx.gemm(Ad, Wd, force_tile=ft, **kw)
    ref = ctx.gemm(Ad, Wd, force_tile=1, **kw)
    ok = torch.equal(out, ref)
    return ok, f"persist M={M} N={N} K={K} wgs={wgs} epi={sorted(kw)}"


def fuzz_strided_persist(ctx, dtype, rng, budget):
    """A / C / residual as column slices of wider buffers (lda, ldc, ldr > width), the way the q | k | v and skip-concat call sites do."""
    M, K = rng.randint(1, 400), 64 * rng.randint(1, 3)
    N = 8 * rng.randint(1, 40)
    pa, pc = 8 * rng.randint(0, 3), 8 * rng.randint(0, 3)
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    Abig = torch.randn(M, K + pa, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    Rbig = torch.randn(M, N + pc, generator=g).to(dtype)
    b = torch.randn(N, generator=g)
    Ad, Wd, Rd, bd = dev(ctx, Abig), dev(ctx, W), dev(ctx, Rbig), dev(ctx, b)
    o1 = torch.zeros(M, N + pc, dtype=dtype, device=ctx.torch_device)
    o2 = torch.zeros_like(o1)
    off_a, off_c = rng.choice([0, pa]), rng.choice([0, pc])
    ctx.gemm(Ad[:, off_a:off_a + K], Wd, bias=bd, res=Rd[:, off_c:off_c + N], out=o1[:, off_c:off_c + N], force_tile=18)
    ctx.gemm(Ad[:, off_a:off_a + K], Wd, bias=bd, res=Rd[:, off_c:off_c + N], out=o2[:, off_c:off_c + N], force_tile=1)
    return torch.equal(o1, o2), f"persist-strided M={M} N={N} K={K} lda={K + pa} ldc={N + pc} offs=({off_a},{off_c})"


def fuzz_tq(ctx, dtype, rng, budget):
    Fr, HW = rng.randint(1, 32), rng.randint(1, 12)
    while Fr * HW > 200:
        HW = max(1, HW - 1)
    C, heads = 320, 5
    M = Fr * HW
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).to(dtype)
    Wqkv = torch.randn(3 * C, C, generator=g) / math.sqrt(C)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    Wf = (Wqkv * gamma[None]).to(dtype)
    cb = (Wqkv @ beta).contiguous()
    cs = Wf.float().sum(1).contiguous()
    perm = torch.cat([torch.arange(64) + part * C + h * 64 for h in range(heads) for part in range(3)])
    xd = dev(ctx, x)
    rowab = ctx.layer_norm_rowab(xd, eps=1e-5)
    qkv = ctx.gemm(xd, dev(ctx, Wf), bias=dev(ctx, cb), rowab=rowab, colsum=dev(ctx, cs), force_tile=30)
    two = ctx.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], Fr, HW, heads)
    one = ctx.temporal_qkv_attn(xd, dev(ctx, Wf[perm].contiguous()), dev(ctx, cb[perm].contiguous()), dev(ctx, cs[perm].contiguous()), rowab, Fr, HW)
    return torch.equal(one, two), f"tq F={Fr} HW={HW}"


def nhwc_rows(x):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def fuzz_conv_split(ctx, dtype, rng, budget):
    """3x3 conv (stride 1 / the stride-2 Downsample with pad (2, 1)) and the temporal conv on "8 CUs" (force_tile 1008: the launcher
    balances rounds for 8 CUs) against the unsplit 8-wave tile.  Output rows = 2 full rounds of 8 tile rows + a small remainder (and
    > 4096, where the big tiles are the automatic choice): the shape the tail split is for; `split=` says whether it was taken."""
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    Cin, Cout = 64, 8 * rng.randint(16, 32)      # K = 9 Cin must be a multiple of 64
    ncu = 8
    kind = rng.choice(["conv", "down", "tconv"])
    target = 4096 + rng.randint(1, 500)
    if kind == "tconv":
        Fr = rng.randint(1, 9)
        HW = target // Fr + 1
        C = 64 * rng.randint(2, 4)
        a = torch.randn(Fr * HW, C, generator=g).to(dtype)
        w = (torch.randn(C, 3 * C, generator=g) / math.sqrt(3 * C)).to(dtype)
        b = torch.randn(C, generator=g)
        args = dict(mode=L.A_TCONV3, temporal=(Fr, HW, C))
        Ad, Wd, bd = dev(ctx, a), dev(ctx, w), dev(ctx, b)
        what = f"tconv F={Fr} HW={HW} C={C}"
    else:
        NB, Ho = rng.randint(1, 12), rng.randint(7, 30)
        Wo = target // (NB * Ho) + 1
        if kind == "conv":
            H, Wd_ = Ho, Wo
            conv = (NB, H, Wd_, Cin, Ho, Wo, 1, 1, 1)
        else:                                         # Ho = (H + 3 - 3) // 2 + 1 with pad (2, 1) and stride 2 (unet_v2v.py Downsample)
            H, Wd_ = 2 * (Ho - 1) + rng.randint(0, 1), 2 * (Wo - 1) + rng.randint(0, 1)
            conv = (NB, H, Wd_, Cin, Ho, Wo, 2, 2, 2)
        x = torch.randn(NB, Cin, H, Wd_, generator=g).to(dtype)
        w = (torch.randn(Cout, 9 * Cin, generator=g) / math.sqrt(9 * Cin)).to(dtype)
        b = torch.randn(Cout, generator=g)
        args = dict(mode=L.A_CONV3X3, conv=conv)
        Ad, Wd, bd = dev(ctx, nhwc_rows(x)), dev(ctx, w), dev(ctx, b)
        what = f"{kind} NB={NB} {H}x{Wd_}->{Ho}x{Wo} {Cin}->{Cout}"
    n0 = ctx.lib.gemm_split_count(ctx.h)
    whole = ctx.gemm(Ad, Wd, bias=bd, force_tile=1, **args)
    split = ctx.gemm(Ad, Wd, bias=bd, force_tile=1000 + ncu, **args)
    took = ctx.lib.gemm_split_count(ctx.h) - n0
    return torch.equal(whole, split), what + f" rows={whole.shape[0]} split={took}"


def fuzz_astat(ctx, dtype, rng, budget):
    """the A-stationary K = 320 kernel (gemm_as.h, tile 30) against the 8-wave tile: folded LayerNorm, plain and GEGLU; the two
    row-affine epilogues may differ by one rounding of the 16-bit output, so the comparison is <= 1 ulp, not bit for bit"""
    M, N = rng.randint(1, 700), 64 * rng.randint(1, 12)
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    A = (torch.randn(M, 320, generator=g) * 1.2).to(dtype)
    W = (torch.randn(N, 320, generator=g) / math.sqrt(320)).to(dtype)
    b = torch.randn(N, generator=g)
    Ad, Wd, bd = dev(ctx, A), dev(ctx, W), dev(ctx, b)
    rowab = ctx.layer_norm_rowab(Ad, eps=1e-5)
    cs = dev(ctx, W.float().sum(1).contiguous())
    geglu = rng.random() < 0.5
    o30 = ctx.gemm(Ad, Wd, bias=bd, rowab=rowab, colsum=cs, geglu=geglu, force_tile=30).float()
    o1 = ctx.gemm(Ad, Wd, bias=bd, rowab=rowab, colsum=cs, geglu=geglu, force_tile=1).float()
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    err = float(((o30 - o1).abs() / (o1.abs() + 1e-3)).max())
    return err <= 1.01 * eps, f"astat M={M} N={N} geglu={geglu} max rel diff {err:.2e} ({'bit-identical' if err == 0 else '<= 1 ulp' if err <= 1.01 * eps else 'OFF'})"


def fuzz_attn(ctx, dtype, rng, budget):
    """flash_attn_v5_kernel (spatial self / cross attention, d = 64) on ragged Nq / Nk, shared K / V batch, strided q | k | v views,
    against fp32 softmax"""
    import torch.nn.functional as F
    B, heads = rng.randint(1, 3), rng.randint(1, 3)
    Nq = rng.choice([rng.randint(1, 70), rng.randint(71, 600)])
    Nk = rng.choice([Nq, 77, rng.randint(1, 1300)])
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    C = heads * 64
    spread = rng.choice([1.0, 3.0])
    if Nq == Nk and rng.random() < 0.5:        # one fused buffer, q | k | v column slices
        qkv = (torch.randn(B, Nq, 3 * C, generator=g) * spread).to(dtype)
        d = dev(ctx, qkv)
        q, k, v = d[:, :, :C], d[:, :, C:2 * C], d[:, :, 2 * C:]
        qh, kh, vh = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
        lay = "fused"
    else:
        kb = rng.choice([1, B])
        qh = (torch.randn(B, Nq, C, generator=g) * spread).to(dtype)
        kh = (torch.randn(kb, Nk, C, generator=g) * spread).to(dtype)
        vh = torch.randn(kb, Nk, C, generator=g).to(dtype)
        q, k, v = dev(ctx, qh), dev(ctx, kh), dev(ctx, vh)
        lay = f"kvbatch={kb}"
    out = ctx.attention(q, k, v, heads).float().cpu()
    tr = lambda t, n: t.float().expand(B, -1, -1).reshape(B, n, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(tr(qh, Nq), tr(kh, k.shape[1]), tr(vh, k.shape[1])).transpose(1, 2).reshape(B, Nq, C)
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    err = float((out - ref).abs().max())
    tol = 4.0 * eps * max(1.0, float(ref.abs().max())) * (2.0 if spread > 1 else 1.0)
    return bool(torch.isfinite(out).all()) and err <= tol, f"attn B={B} heads={heads} Nq={Nq} Nk={k.shape[1]} {lay} spread={spread} err={err:.2e} tol={tol:.2e}"


def fuzz_tattn(ctx, dtype, rng, budget):
    """temporal_attn_kernel over the frame axis (any F up to 128 frames: chunks longer than 32 take the multi-tile form)"""
    import torch.nn.functional as F
    Fr, HW, heads = rng.choice([rng.randint(1, 32), rng.randint(33, 100)]), rng.randint(1, 20), rng.choice([1, 2, 5])
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    C = heads * 64
    qkv = torch.randn(Fr * HW, 3 * C, generator=g).to(dtype)
    d = dev(ctx, qkv)
    out = ctx.temporal_attention(d[:, :C], d[:, C:2 * C], d[:, 2 * C:], Fr, HW, heads).float().cpu()
    tr = lambda t: t.float().reshape(Fr, HW, heads, 64).permute(1, 2, 0, 3)
    ref = F.scaled_dot_product_attention(tr(qkv[:, :C]), tr(qkv[:, C:2 * C]), tr(qkv[:, 2 * C:])).permute(2, 0, 1, 3).reshape(Fr * HW, C)
    eps = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    err = float((out - ref).abs().max())
    tol = 4.0 * eps * max(1.0, float(ref.abs().max()))
    return bool(torch.isfinite(out).all()) and err <= tol, f"tattn F={Fr} HW={HW} heads={heads} err={err:.2e} tol={tol:.2e}"


def fuzz_epi_stats(ctx, dtype, rng, budget):
    """GroupNorm partials / LayerNorm row statistics written by the producer's epilogue (round 5)"""
    import torch.nn.functional as F
    Fr, HW = rng.randint(1, 5), rng.randint(3, 130)
    M = Fr * HW
    K = 64 * rng.randint(1, 3)
    kind = rng.choice(["gn", "gn", "ln"])
    tile = rng.choice([2, 3] if kind == "ln" else [2, 3, 17])
    N = 64 * rng.randint(1, 6) if kind == "gn" else 8 * rng.randint(2, 60)
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    a = dev(ctx, torch.randn(M, K, generator=g).to(dtype))
    w = dev(ctx, (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype))
    b = dev(ctx, torch.randn(N, generator=g) + 2.0)
    res = dev(ctx, torch.randn(M, N, generator=g).to(dtype)) if (rng.random() < 0.5 and tile != 17) else None
    plain = ctx.gemm(a, w, bias=b, res=res, force_tile=tile)
    tag = f"epi-stats {kind} tile={tile} F={Fr} HW={HW} N={N} K={K} res={res is not None}"
    if kind == "gn":
        out, part = ctx.gemm(a, w, bias=b, res=res, force_tile=tile, gn_partial=True)
        if part is None or not torch.equal(out, plain):
            return False, tag
        o = out.double().cpu()
        pad = (-M) % 32
        o2 = torch.cat([o, torch.zeros(pad, N, dtype=torch.float64)]) if pad else o
        o2 = o2.reshape(-1, 32, N // 2, 2)
        want = torch.stack([o2.sum(dim=(1, 3)), (o2 * o2).sum(dim=(1, 3))], dim=-1)
        ok = (part.double().cpu() - want).abs().max().item() <= 3e-5 * (want[..., 1].abs().max().item() + 1.0)
        gam, bet = torch.randn(N, generator=g), torch.randn(N, generator=g)
        for rps in (HW, M):
            y = ctx.group_norm_from_partials(out, part, dev(ctx, gam), dev(ctx, bet), rps, eps=1e-5, silu=False).float().cpu()
            y0 = ctx.group_norm(out, dev(ctx, gam), dev(ctx, bet), rps, eps=1e-5, silu=False).float().cpu()
            ok = ok and (y - y0).abs().max().item() <= (3e-2 if dtype == torch.bfloat16 else 4e-3) * (1 + y0.abs().max().item())
        return ok, tag
    out, part = ctx.gemm(a, w, bias=b, res=res, force_tile=tile, row_stats=True)
    if part is None:
        return False, tag + " (no partials)"
    if not torch.equal(out, plain):
        return False, tag
    o = out.double().cpu()
    p = part.double().cpu()
    ok = (p[..., 0].sum(1) - o.sum(1)).abs().max().item() <= 2e-5 * (o.abs().sum(1).max().item() + 1)
    ok = ok and (p[..., 1].sum(1) - (o * o).sum(1)).abs().max().item() <= 3e-5 * ((o * o).sum(1).max().item() + 1)
    ok = ok and torch.equal(p[..., 2].max(1).values, o.max(1).values)
    ab = ctx.layer_norm_rowab_from_partials(part, N).cpu()
    ab0 = ctx.layer_norm_rowab(out).cpu()
    ok = ok and ((ab - ab0).abs() / (ab0.abs() + 1e-3)).max().item() <= 3e-3
    return ok, tag


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget", type=float, default=6e7, help="M*N*K bound of a GEMM case (the emulator runs ~1e7 MACs per second)")
    ap.add_argument("--kinds", default="persist,strided,tq,conv,astat,attn,tattn,epistats")
    args = ap.parse_args()
    emu = L.Library(os.path.join(ROOT, "tools", "hostemu", "libstar_emu.so"))
    assert emu.is_hostemu
    rng = random.Random(args.seed)
    fns = {"persist": fuzz_persist, "strided": fuzz_strided_persist, "tq": fuzz_tq, "conv": fuzz_conv_split, "astat": fuzz_astat,
           "attn": fuzz_attn, "tattn": fuzz_tattn, "epistats": fuzz_epi_stats}
    kinds = [k for k in args.kinds.split(",") if k]
    bad = 0
    t0 = time.time()
    for i in range(args.cases):
        dtype = rng.choice([torch.float16, torch.bfloat16])
        ctx = L.Context(0, dtype, emu)
        kind = kinds[i % len(kinds)]
        ok, what = fns[kind](ctx, dtype, rng, args.budget)
        bad += not ok
        print(f"[{i:3d}] {'ok  ' if ok else 'FAIL'} {str(dtype).split('.')[-1]:8s} {what}", flush=True)
    print(f"{args.cases} cases, {bad} failures, {time.time() - t0:.0f} s (seed {args.seed})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
