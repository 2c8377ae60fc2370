"""ctypes binding of libstar_hip.so (C ABI: include/star_hip.h).

This is the reference-side stub a STAR maintainer would add:
PyTorch-ROCm tensors own the device memory, this module passes raw pointers.

There is NO CPU fallback: the default library is the in-tree HIP build and it
needs a gfx950 device.  `tools/hostemu/libstar_emu.so` (the SIMT emulator build
of the very same sources) can only be loaded by passing its path explicitly,
which the tests do to check kernel index logic on GPU-less machines.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libstar_hip.so")

F16, BF16, F32 = 0, 1, 2
A_PLAIN, A_CONV3X3, A_CONV3X3_UP, A_TCONV3 = 0, 1, 2, 3
EPI_BIAS, EPI_RES, EPI_GEGLU, EPI_OUT_F32, EPI_GELU_TANH, EPI_ROWAFF = 1, 2, 4, 8, 16, 32

_TORCH2STAR = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}
_STAR2TORCH = {v: k for k, v in _TORCH2STAR.items()}


LN_PLAIN, LN_GATE_LINEAR, LN_GATE_MAP, LN_STATS_ONLY = 0, 1, 2, 3


class AttnDesc(ctypes.Structure):
    _fields_ = [
        ("Q", ctypes.c_void_p), ("K", ctypes.c_void_p), ("V", ctypes.c_void_p), ("O", ctypes.c_void_p),
        ("ldq", ctypes.c_int32), ("ldk", ctypes.c_int32), ("ldv", ctypes.c_int32), ("ldo", ctypes.c_int32),
        ("bsq", ctypes.c_int64), ("bsk", ctypes.c_int64), ("bsv", ctypes.c_int64), ("bso", ctypes.c_int64),
        ("Nq", ctypes.c_int32), ("Nk", ctypes.c_int32), ("heads", ctypes.c_int32), ("batch", ctypes.c_int32),
        ("scale", ctypes.c_float), ("variant", ctypes.c_int32), ("causal", ctypes.c_int32), ("reserved", ctypes.c_int32),
    ]


class TAttnDesc(ctypes.Structure):
    _fields_ = [
        ("Q", ctypes.c_void_p), ("K", ctypes.c_void_p), ("V", ctypes.c_void_p), ("O", ctypes.c_void_p),
        ("ldq", ctypes.c_int32), ("ldk", ctypes.c_int32), ("ldv", ctypes.c_int32), ("ldo", ctypes.c_int32),
        ("F", ctypes.c_int32), ("HW", ctypes.c_int32), ("heads", ctypes.c_int32),
        ("scale", ctypes.c_float),
    ]


PROF_KINDS = ["attn_self", "attn_cross", "temporal_attn", "gemm", "conv3x3", "tconv", "group_norm", "layer_norm", "misc"]


class ProfEntry(ctypes.Structure):
    _fields_ = [("ms", ctypes.c_double), ("flops", ctypes.c_double), ("bytes", ctypes.c_double), ("max_flops", ctypes.c_double),
                ("max_flops_ms", ctypes.c_double), ("launches", ctypes.c_int64)]


class StarError(RuntimeError):
    pass


class TqDesc(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("W", ctypes.c_void_p), ("O", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("colsum", ctypes.c_void_p), ("rowab", ctypes.c_void_p),
        ("lda", ctypes.c_int32), ("ldo", ctypes.c_int32), ("HW", ctypes.c_int32), ("F", ctypes.c_int32),
        ("C", ctypes.c_int32), ("heads", ctypes.c_int32), ("scale", ctypes.c_float),
    ]


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("W", ctypes.c_void_p), ("C", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("res", ctypes.c_void_p),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("lda", ctypes.c_int32), ("ldc", ctypes.c_int32), ("ldr", ctypes.c_int32),
        ("mode", ctypes.c_int32),
        ("H", ctypes.c_int32), ("Wd", ctypes.c_int32), ("Cin", ctypes.c_int32),
        ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32), ("stride", ctypes.c_int32),
        ("pad_t", ctypes.c_int32), ("pad_l", ctypes.c_int32),
        ("HW", ctypes.c_int32), ("F", ctypes.c_int32), ("up_crop", ctypes.c_int32),
        ("epi", ctypes.c_int32), ("force_tile", ctypes.c_int32),
        ("rowab", ctypes.c_void_p), ("colsum", ctypes.c_void_p),
    ]


def _sig(lib, name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


class Library:
    """A loaded libstar_hip.so (or, explicitly, the emulator build for tests)."""

    def __init__(self, path=None):
        self.path = path or DEFAULT_LIB
        if not os.path.isfile(self.path):
            raise StarError(
                f"{self.path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the HIP hot path)")
        self.cdll = ctypes.CDLL(self.path)
        c = self.cdll
        vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
        self.is_hostemu = bool(_sig(c, "star_is_hostemu", i32)())
        self.has_bench_variants = bool(_sig(c, "star_has_bench_variants", i32)())
        self.ctx_create = _sig(c, "star_ctx_create", i32, i32, i32, ctypes.POINTER(vp))
        self.ctx_destroy = _sig(c, "star_ctx_destroy", None, vp)
        self.last_error = _sig(c, "star_last_error", ctypes.c_char_p, vp)
        self.set_stream = _sig(c, "star_set_stream", i32, vp, vp)
        self.sync = _sig(c, "star_sync", i32, vp)
        self.pool_trim = _sig(c, "star_pool_trim", i32, vp)
        self.pool_bytes = _sig(c, "star_pool_bytes", sz, vp)
        self.pool_peak_bytes = _sig(c, "star_pool_peak_bytes", sz, vp)
        self.gemm_split_count = _sig(c, "star_gemm_split_count", i64, vp)
        self.gn_fused_count = _sig(c, "star_gn_fused_count", i64, vp)
        self.ln_fused_count = _sig(c, "star_ln_fused_count", i64, vp)
        self.gemm = _sig(c, "star_gemm", i32, vp, ctypes.POINTER(GemmDesc))
        self.gemm_gn = _sig(c, "star_gemm_gn", i32, vp, ctypes.POINTER(GemmDesc), vp, ctypes.POINTER(ctypes.c_int32))
        self.gemm_rowstats = _sig(c, "star_gemm_rowstats", i32, vp, ctypes.POINTER(GemmDesc), vp, i32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32))
        f32 = ctypes.c_float
        self.attn_fwd = _sig(c, "star_attn_fwd", i32, vp, ctypes.POINTER(AttnDesc))
        self.softmax_rows = _sig(c, "star_softmax_rows", i32, vp, vp, i32, vp, i32, i32, i32, ctypes.c_float)
        self.temporal_attn_fwd = _sig(c, "star_temporal_attn_fwd", i32, vp, ctypes.POINTER(TAttnDesc))
        self.temporal_qkv_attn = _sig(c, "star_temporal_qkv_attn", i32, vp, ctypes.POINTER(TqDesc))
        self.group_norm = _sig(c, "star_group_norm", i32, vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, f32, i32)
        self.group_norm_from_partials = _sig(c, "star_group_norm_from_partials", i32, vp, vp, i32, vp, i32, vp, vp, i32, i32, i32, f32, i32, vp)
        self.layer_norm = _sig(c, "star_layer_norm", i32, vp, vp, i32, vp, i32, vp, vp, i32, i32, f32, i32, vp, vp, i32, i32)
        self.layer_norm_rowab_from_partials = _sig(c, "star_layer_norm_rowab_from_partials", i32, vp, vp, i32, vp, i32, i32, ctypes.c_float, i32, vp, vp, i32, i32)
        self.layer_norm_rowab = _sig(c, "star_layer_norm_rowab", i32, vp, vp, i32, vp, i32, i32, f32, i32, vp, vp, i32, i32)
        self.concat_add = _sig(c, "star_concat_add", i32, vp, vp, vp, vp, vp, i32, i32, i32)
        self.add = _sig(c, "star_add", i32, vp, vp, vp, vp, i64)
        self.stem_im2col = _sig(c, "star_stem_im2col", i32, vp, vp, vp, i32, i32, i32, i32)
        self.rows_to_latent = _sig(c, "star_rows_to_latent", i32, vp, vp, vp, i32, i32, i64)
        self.gemv = _sig(c, "star_gemv", i32, vp, vp, vp, vp, vp, i32, i32, i32, i32)
        self.cast = _sig(c, "star_cast", i32, vp, vp, vp, i64)
        self.resize_pad = _sig(c, "star_resize_pad", i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32)
        self.plane_stats = _sig(c, "star_plane_stats", i32, vp, vp, vp, i32, i64, f32, f32, i32, f32)
        self.color_fix = _sig(c, "star_color_fix", i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32)
        self.text_build = _sig(c, "star_text_build", i32, vp, i32, i32, i32)
        self.text_forward = _sig(c, "star_text_forward", i32, vp, vp, i32, i32, i32, vp)
        self.color_fix_u8 = _sig(c, "star_color_fix_u8", i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32)
        self.adain_color_fix = _sig(c, "star_adain_color_fix", i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32)
        self.profile_begin = _sig(c, "star_profile_begin", i32, vp)
        self.profile_begin_kinds = _sig(c, "star_profile_begin_kinds", i32, vp, ctypes.c_uint32)
        self.profile_end = _sig(c, "star_profile_end", i32, vp, ctypes.POINTER(ProfEntry))


_default_library = None


def default_library():
    global _default_library
    if _default_library is None:
        _default_library = Library()
    return _default_library


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def device_index(device):
    """int | torch.device -> device ordinal; a 'cuda' device without an index means this process's CURRENT device (one rank per GPU:
    never silently GPU 0)"""
    if isinstance(device, torch.device):
        if device.index is not None:
            return device.index
        return torch.cuda.current_device() if device.type == "cuda" and torch.cuda.is_available() else 0
    return int(device)


def pack_conv3x3_weight(w):
    """[Cout, Cin, 3, 3] -> the [Cout, 9 * Cin] matrix star_gemm's 3x3 modes expect: K index = (c // 64, tap, c % 64), i.e. 64-channel
    blocks outermost and the nine taps (ky * 3 + kx) inside a block; Cin % 64 == 0."""
    Cout, Cin = w.shape[:2]
    assert Cin % 64 == 0 and tuple(w.shape[2:]) == (3, 3)
    return w.reshape(Cout, Cin // 64, 64, 9).permute(0, 1, 3, 2).reshape(Cout, 9 * Cin).contiguous()


class Context:
    """One star_ctx: a device, a compute dtype (fp16/bf16), a stream, a workspace pool."""

    def __init__(self, device=0, dtype=torch.float16, library=None):
        self.lib = library or default_library()
        device = device_index(device)
        if not self.lib.is_hostemu and not torch.cuda.is_available():
            raise StarError("star_amd needs a ROCm GPU (gfx950); no CPU fallback exists for the HIP hot path")
        self.device_index = int(device)
        self.torch_device = torch.device("cpu") if self.lib.is_hostemu else torch.device("cuda", self.device_index)
        self.dtype = dtype
        h = ctypes.c_void_p()
        rc = self.lib.ctx_create(self.device_index, _TORCH2STAR[dtype], ctypes.byref(h))
        if rc:
            why = {2: "unsupported dtype", 3: "no such device", 4: "hipSetDevice failed", 5: "out of device memory",
                   6: "the device is not a gfx950 (MI355X) with the 160 KB LDS opt-in"}.get(rc, "")
            raise StarError(f"star_ctx_create failed (rc={rc}) {why}")
        self.h = h
        if not self.lib.is_hostemu:
            self.lib.set_stream(self.h, ctypes.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream))

    def trim(self):
        """give the arena's cached blocks back to the driver (phase boundaries; the reference calls torch.cuda.empty_cache() there)"""
        self._check(self.lib.pool_trim(self.h), "pool_trim")

    def close(self):
        if getattr(self, "h", None):
            self.lib.ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _check(self, rc, what):
        if rc:
            raise StarError(f"{what} failed: {self.lib.last_error(self.h).decode()}")

    def _chk_tensor(self, t, dtype=None):
        if t is None:
            return
        if t.device != self.torch_device:
            raise StarError(f"tensor on {t.device}, context on {self.torch_device}")
        if dtype is not None and t.dtype != dtype:
            raise StarError(f"tensor dtype {t.dtype}, expected {dtype}")

    def sync(self):
        self._check(self.lib.sync(self.h), "sync")

    def use_current_stream(self):
        if not self.lib.is_hostemu:
            self.lib.set_stream(self.h, ctypes.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream))

    def profile_begin(self, kinds=None):
        """HIP events around every launch, or only around the families named in `kinds` (PROF_KINDS names): events cost stream time"""
        if kinds is None:
            self._check(self.lib.profile_begin(self.h), "profile_begin")
        else:
            mask = 0
            for k in kinds:
                mask |= 1 << PROF_KINDS.index(k)
            self._check(self.lib.profile_begin_kinds(self.h, mask), "profile_begin")

    def profile_end(self):
        """-> {family: {ms, flops, bytes, launches, max_flops, max_flops_ms}} measured with HIP events on the launch stream."""
        arr = (ProfEntry * len(PROF_KINDS))()
        self._check(self.lib.profile_end(self.h, arr), "profile_end")
        return {k: {"ms": arr[i].ms, "flops": arr[i].flops, "bytes": arr[i].bytes, "launches": arr[i].launches,
                    "max_flops": arr[i].max_flops, "max_flops_ms": arr[i].max_flops_ms} for i, k in enumerate(PROF_KINDS)}

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.torch_device)

    # ------------------------------------------------------------------ kernels
    def gemm(self, A, W, bias=None, res=None, out=None, *, mode=A_PLAIN, M=None, conv=None, temporal=None,
             geglu=False, out_f32=False, force_tile=0, up_crop=1, gelu_tanh=False, rowab=None, colsum=None, gn_partial=False, row_stats=False):
        """out[M, N] = epilogue(A' @ W^T).  A: [rows, lda] activations (channels-last tokens);
        W: [N, K]; conv=(NB,H,Wd,Cin,Ho,Wo,stride,pad_t,pad_l); temporal=(F,HW,Cin).
        gn_partial=True: star_gemm_gn -- returns (out, partial) with partial fp32 [ceil(M/32), N/2, 2] = the GroupNorm partial statistics of
        out written by the epilogue, or (out, None) when the launcher's tile has no such flavour.
        row_stats=True: star_gemm_rowstats -- returns (out, partial) with partial fp32 [M, parts, 4] = per-row (sum, sum of squares, max, 0) of
        out per column part, or (out, None)."""
        self._chk_tensor(A, self.dtype); self._chk_tensor(W, self.dtype)
        self._chk_tensor(bias, torch.float32); self._chk_tensor(res, self.dtype)
        N, K = W.shape
        d = GemmDesc()
        d.mode = mode
        d.stride, d.pad_t, d.pad_l = 1, 1, 1
        d.up_crop = up_crop
        if mode == A_PLAIN:
            M = A.shape[0] if M is None else M
        elif mode in (A_CONV3X3, A_CONV3X3_UP):
            NB, H, Wd, Cin, Ho, Wo, stride, pad_t, pad_l = conv
            d.H, d.Wd, d.Cin, d.Ho, d.Wo, d.stride, d.pad_t, d.pad_l = H, Wd, Cin, Ho, Wo, stride, pad_t, pad_l
            M = NB * Ho * Wo
        else:
            F_, HW, Cin = temporal
            d.F, d.HW, d.Cin = F_, HW, Cin
            M = F_ * HW
        n_out = N // 2 if geglu else N
        if out is None:
            out = torch.empty(M, n_out, dtype=torch.float32 if out_f32 else self.dtype, device=self.torch_device)
        d.A, d.W, d.C = A.data_ptr(), W.data_ptr(), out.data_ptr()
        d.bias = bias.data_ptr() if bias is not None else None
        d.res = res.data_ptr() if res is not None else None
        d.M, d.N, d.K = M, N, K
        d.lda, d.ldc = A.stride(0), out.stride(0)
        d.ldr = res.stride(0) if res is not None else 0
        d.epi = (EPI_BIAS if bias is not None else 0) | (EPI_RES if res is not None else 0) | \
                (EPI_GEGLU if geglu else 0) | (EPI_OUT_F32 if out_f32 else 0) | (EPI_GELU_TANH if gelu_tanh else 0)
        if rowab is not None:        # a LayerNorm folded into this projection: out = a_m * acc + b_m * colsum[n] + bias[n]
            self._chk_tensor(rowab, torch.float32); self._chk_tensor(colsum, torch.float32)
            assert tuple(rowab.shape) == (M, 2) and colsum.numel() == N and bias is not None
            d.rowab, d.colsum = rowab.data_ptr(), colsum.data_ptr()
            d.epi |= EPI_ROWAFF
        d.force_tile = force_tile
        if row_stats:
            cap = 2 * ((N + 127) // 128)
            part = torch.zeros(M, cap, 4, dtype=torch.float32, device=self.torch_device)
            wrote, parts = ctypes.c_int32(0), ctypes.c_int32(0)
            self._check(self.lib.gemm_rowstats(self.h, ctypes.byref(d), _ptr(part), cap, ctypes.byref(parts), ctypes.byref(wrote)), "gemm_rowstats")
            if not wrote.value:
                return out, None
            return out, part.reshape(-1)[: M * parts.value * 4].reshape(M, parts.value, 4)
        if gn_partial:
            part = torch.empty((M + 31) // 32, N // 2, 2, dtype=torch.float32, device=self.torch_device)
            wrote = ctypes.c_int32(0)
            self._check(self.lib.gemm_gn(self.h, ctypes.byref(d), _ptr(part), ctypes.byref(wrote)), "gemm_gn")
            return out, (part if wrote.value else None)
        self._check(self.lib.gemm(self.h, ctypes.byref(d)), "gemm")
        return out

    def attention(self, q, k, v, heads, out=None, scale=None, variant=9, causal=False):
        """softmax(q k^T * scale) v per (batch, head).  q: [B, Nq, heads*64]; k, v: [B or 1, Nk, heads*64]
        (a leading dim of 1 is shared by all batches).  Views with a row stride are fine (fused QKV buffers)."""
        for t in (q, k, v):
            self._chk_tensor(t, self.dtype)
            assert t.dim() == 3 and t.stride(2) == 1
        B, Nq, _ = q.shape
        Nk = k.shape[1]
        if out is None:
            out = torch.empty(B, Nq, heads * 64, dtype=self.dtype, device=self.torch_device)
        d = AttnDesc()
        d.Q, d.K, d.V, d.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
        d.ldq, d.ldk, d.ldv, d.ldo = q.stride(1), k.stride(1), v.stride(1), out.stride(1)
        d.bsq, d.bso = q.stride(0), out.stride(0)
        d.bsk = k.stride(0) if k.shape[0] > 1 else 0
        d.bsv = v.stride(0) if v.shape[0] > 1 else 0
        d.Nq, d.Nk, d.heads, d.batch = Nq, Nk, heads, B
        d.scale = float(scale if scale is not None else 64 ** -0.5)
        d.variant = variant
        d.causal = 1 if causal else 0
        self._check(self.lib.attn_fwd(self.h, ctypes.byref(d)), "attn_fwd")
        return out

    def softmax_rows(self, s, n, scale, ldp=None):
        """P[r, :n] = softmax(s[r, :n] * scale) in the context dtype, P[r, n:ldp] = 0 (the VAE mid-block attention's logits pass, vae.cpp).
        s: fp32 [rows, lds] (columns beyond n are ignored)."""
        self._chk_tensor(s, torch.float32)
        assert s.dim() == 2 and s.stride(1) == 1
        rows = s.shape[0]
        ldp = int(ldp if ldp is not None else s.shape[1])
        out = torch.empty(rows, ldp, dtype=self.dtype, device=self.torch_device)
        self._check(self.lib.softmax_rows(self.h, ctypes.c_void_p(s.data_ptr()), s.stride(0), ctypes.c_void_p(out.data_ptr()), ldp, rows, int(n), float(scale)),
                    "softmax_rows")
        return out

    def temporal_attention(self, q, k, v, F_, HW, heads, out=None, scale=None):
        """attention over the frame axis for every (pixel, head); q/k/v: [F*HW, >=heads*64] token matrices."""
        for t in (q, k, v):
            self._chk_tensor(t, self.dtype)
            assert t.dim() == 2 and t.stride(1) == 1
        if out is None:
            out = torch.empty(F_ * HW, heads * 64, dtype=self.dtype, device=self.torch_device)
        d = TAttnDesc()
        d.Q, d.K, d.V, d.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
        d.ldq, d.ldk, d.ldv, d.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
        d.F, d.HW, d.heads = F_, HW, heads
        d.scale = float(scale if scale is not None else 64 ** -0.5)
        self._check(self.lib.temporal_attn_fwd(self.h, ctypes.byref(d)), "temporal_attn_fwd")
        return out

    def temporal_qkv_attn(self, x, W_heads, bias, colsum, rowab, F_, HW, heads=5, out=None, scale=None):
        """q | k | v projection (LayerNorm folded) + attention over the frame axis in one kernel (gemm_tq.h): x [F*HW, 320] token rows,
        W_heads [960, 320] with 64-row tiles ordered (q_h, k_h, v_h) per head -> O [F*HW, 320]."""
        self._chk_tensor(x, self.dtype); self._chk_tensor(W_heads, self.dtype)
        for t in (bias, colsum, rowab):
            self._chk_tensor(t, torch.float32)
        C = x.shape[1]
        if out is None:
            out = torch.empty(F_ * HW, C, dtype=self.dtype, device=self.torch_device)
        d = TqDesc()
        d.A, d.W, d.O, d.bias, d.colsum, d.rowab = x.data_ptr(), W_heads.data_ptr(), out.data_ptr(), bias.data_ptr(), colsum.data_ptr(), rowab.data_ptr()
        d.lda, d.ldo, d.HW, d.F, d.C, d.heads = x.stride(0), out.stride(0), HW, F_, C, heads
        d.scale = float(scale if scale is not None else 64 ** -0.5)
        self._check(self.lib.temporal_qkv_attn(self.h, ctypes.byref(d)), "temporal_qkv_attn")
        return out

    def group_norm(self, x, gamma, beta, rows_per_stat, eps=1e-5, silu=False, out=None):
        """GroupNorm(32) over channels-last rows x: [rows, C]."""
        self._chk_tensor(x, self.dtype); self._chk_tensor(gamma, torch.float32); self._chk_tensor(beta, torch.float32)
        rows, C = x.shape
        if out is None:
            out = torch.empty(rows, C, dtype=self.dtype, device=self.torch_device)
        self._check(self.lib.group_norm(self.h, _ptr(x), x.stride(0), _ptr(out), out.stride(0), _ptr(gamma), _ptr(beta),
                                        rows, C, rows_per_stat, float(eps), int(silu)), "group_norm")
        return out

    def group_norm_from_partials(self, x, partial, gamma, beta, rows_per_stat, eps=1e-5, silu=False, out=None):
        """GroupNorm(32) of x: [rows, C] whose producer wrote `partial` (gemm(..., gn_partial=True)): no statistics pass."""
        self._chk_tensor(x, self.dtype); self._chk_tensor(gamma, torch.float32); self._chk_tensor(beta, torch.float32)
        self._chk_tensor(partial, torch.float32)
        rows, C = x.shape
        assert tuple(partial.shape) == ((rows + 31) // 32, C // 2, 2)
        if out is None:
            out = torch.empty(rows, C, dtype=self.dtype, device=self.torch_device)
        self._check(self.lib.group_norm_from_partials(self.h, _ptr(x), x.stride(0), _ptr(out), out.stride(0), _ptr(gamma), _ptr(beta),
                                                      rows, C, rows_per_stat, float(eps), int(silu), _ptr(partial)), "group_norm_from_partials")
        return out

    def layer_norm(self, x, gamma, beta, eps=1e-5, mode=LN_PLAIN, gate_w=None, maps=None, H=0, W=0, out=None):
        self._chk_tensor(x, self.dtype)
        rows, C = x.shape
        if mode == LN_STATS_ONLY:
            if maps is None:
                maps = torch.empty(rows, 2, dtype=torch.float32, device=self.torch_device)
            self._check(self.lib.layer_norm(self.h, _ptr(x), x.stride(0), None, 8, None, None, rows, C, float(eps), mode,
                                            None, _ptr(maps), H, W), "layer_norm")
            return maps
        if out is None:
            out = torch.empty(rows, C, dtype=self.dtype, device=self.torch_device)
        self._check(self.lib.layer_norm(self.h, _ptr(x), x.stride(0), _ptr(out), out.stride(0), _ptr(gamma), _ptr(beta),
                                        rows, C, float(eps), mode, _ptr(gate_w), _ptr(maps), H, W), "layer_norm")
        return out

    def layer_norm_rowab(self, x, eps=1e-5, mode=LN_PLAIN, gate_w=None, maps=None, H=0, W=0):
        """row statistics of a LayerNorm folded into the projection behind it: rowab[row] = (a, b), LN(gate x) = (a x + b) gamma + beta"""
        self._chk_tensor(x, self.dtype)
        rows, C = x.shape
        rowab = torch.empty(rows, 2, dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.layer_norm_rowab(self.h, _ptr(x), x.stride(0), _ptr(rowab), rows, C, float(eps), mode, _ptr(gate_w),
                                              _ptr(maps), H, W), "layer_norm_rowab")
        return rowab

    def layer_norm_rowab_from_partials(self, partial, C, eps=1e-5, mode=LN_PLAIN, gate_w=None, maps=None, H=0, W=0):
        """layer_norm_rowab (mode 3: the LIEM maps, returned instead) from the producer's row statistics (gemm(..., row_stats=True))"""
        self._chk_tensor(partial, torch.float32)
        rows, parts, _ = partial.shape
        assert partial.is_contiguous()
        if mode == LN_STATS_ONLY:
            maps = torch.empty(rows, 2, dtype=torch.float32, device=self.torch_device)
            rowab = None
        else:
            rowab = torch.empty(rows, 2, dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.layer_norm_rowab_from_partials(self.h, _ptr(partial), parts, _ptr(rowab), rows, C, float(eps), mode, _ptr(gate_w),
                                                            _ptr(maps), H, W), "layer_norm_rowab_from_partials")
        return maps if mode == LN_STATS_ONLY else rowab

    def concat_add(self, a, b, c=None):
        rows, C1 = a.shape
        C2 = b.shape[1]
        out = torch.empty(rows, C1 + C2, dtype=self.dtype, device=self.torch_device)
        self._check(self.lib.concat_add(self.h, _ptr(a), _ptr(b), _ptr(c), _ptr(out), rows, C1, C2), "concat_add")
        return out

    def add(self, a, b):
        out = torch.empty_like(a)
        self._check(self.lib.add(self.h, _ptr(a), _ptr(b), _ptr(out), a.numel()), "add")
        return out

    def stem_im2col(self, latent):
        """latent [1, Cl, F, H, W] fp32 -> [F*H*W, 64] im2col rows of the 3x3 stem conv."""
        self._chk_tensor(latent, torch.float32)
        _, Cl, F_, H, W = latent.shape
        out = torch.empty(F_ * H * W, 64, dtype=self.dtype, device=self.torch_device)
        self._check(self.lib.stem_im2col(self.h, _ptr(latent.contiguous()), _ptr(out), Cl, F_, H, W), "stem_im2col")
        return out

    def rows_to_latent(self, rows, Cl, F_, H, W):
        self._chk_tensor(rows, torch.float32)
        out = torch.empty(1, Cl, F_, H, W, dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.rows_to_latent(self.h, _ptr(rows), _ptr(out), Cl, rows.stride(0), F_ * H * W), "rows_to_latent")
        return out

    def gemv(self, x, W, b=None, silu_in=False, silu_out=False):
        self._chk_tensor(x, torch.float32); self._chk_tensor(W, self.dtype)
        N, K = W.shape
        y = torch.empty(N, dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.gemv(self.h, _ptr(x), _ptr(W), _ptr(b), _ptr(y), N, K, int(silu_in), int(silu_out)), "gemv")
        return y

    def resize_pad(self, video, target_hw, padding=(0, 0, 0, 0), pad_value=1.0):
        """F.interpolate(video, target_hw, mode='bilinear') + F.pad(video, padding, 'constant', pad_value)
        (video_to_video_model.py:81-87).  video: fp32 [F, C, h, w]; padding = (left, right, top, bottom)."""
        self._chk_tensor(video, torch.float32)
        video = video.contiguous()
        F_, C, h, w = video.shape
        th, tw = int(target_hw[0]), int(target_hw[1])
        pl, pr, pt, pb = (int(v) for v in padding)
        out = torch.empty(F_, C, th + pt + pb, tw + pl + pr, dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.resize_pad(self.h, _ptr(video), _ptr(out), F_ * C, h, w, th, tw, pl, pr, pt, pb, float(pad_value)),
                    "resize_pad")
        return out

    def plane_stats(self, x, scale=1.0, shift=0.0, clamp01=False, eps=1e-5):
        """calc_mean_std (color_fix.py:62-74) of x*scale+shift over the last two dims -> [..., 2] = (mean, std)."""
        self._chk_tensor(x, torch.float32)
        x = x.contiguous()
        lead = x.shape[:-2]
        planes = 1
        for d in lead:
            planes *= int(d)
        n = int(x.shape[-2]) * int(x.shape[-1])
        out = torch.empty(*lead, 2, dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.plane_stats(self.h, _ptr(x), _ptr(out), planes, n, float(scale), float(shift), int(clamp01), float(eps)),
                    "plane_stats")
        return out

    def color_fix(self, video, source, as_uint8=False):
        """tensor2vid + adain_color_fix (inference_utils.py:16-23, color_fix.py:15-29).  video: fp32 [1, C, F, H, W];
        source: fp32 [F, C, h, w] in [-1, 1]  ->  fp32 [F, H, W, C] in [0, 255], or (as_uint8) the uint8 frames save_video's
        `.astype('uint8')` makes of them (inference_utils.py:92), truncated on the GPU."""
        self._chk_tensor(video, torch.float32); self._chk_tensor(source, torch.float32)
        video, source = video.contiguous(), source.contiguous()
        assert video.dim() == 5 and video.shape[0] == 1 and source.dim() == 4
        _, C, F_, H, W = video.shape
        assert source.shape[0] == F_ and source.shape[1] == C, "video and source disagree on frames / channels"
        out = torch.empty(F_, H, W, C, dtype=torch.uint8 if as_uint8 else torch.float32, device=self.torch_device)
        fn = self.lib.color_fix_u8 if as_uint8 else self.lib.color_fix
        self._check(fn(self.h, _ptr(video), _ptr(source), _ptr(out), F_, C, H, W, source.shape[2], source.shape[3]), "color_fix")
        return out

    def adain_color_fix(self, target, source):
        """adain_color_fix on its own (color_fix.py:15-29): target fp32 [F, H, W, C] in [0, 255] (tensor2vid result),
        source fp32 [F, C, h, w] in [-1, 1]  ->  fp32 [F, H, W, C] in [0, 255]."""
        self._chk_tensor(target, torch.float32); self._chk_tensor(source, torch.float32)
        target, source = target.contiguous(), source.contiguous()
        assert target.dim() == 4 and source.dim() == 4
        F_, H, W, C = target.shape
        assert source.shape[0] == F_ and source.shape[1] == C, "target and source disagree on frames / channels"
        out = torch.empty(F_, H, W, C, dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.adain_color_fix(self.h, _ptr(target), _ptr(source), _ptr(out), F_, C, H, W, source.shape[2], source.shape[3]),
                    "adain_color_fix")
        return out

    def cast(self, x):
        self._chk_tensor(x, torch.float32)
        y = torch.empty(x.shape, dtype=self.dtype, device=self.torch_device)
        self._check(self.lib.cast(self.h, _ptr(x.contiguous()), _ptr(y), x.numel()), "cast")
        return y
