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
EPI_BIAS, EPI_RES, EPI_GEGLU, EPI_OUT_F32 = 1, 2, 4, 8

_TORCH2STAR = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}
_STAR2TORCH = {v: k for k, v in _TORCH2STAR.items()}


class StarError(RuntimeError):
    pass


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
        ("HW", ctypes.c_int32), ("F", ctypes.c_int32),
        ("epi", ctypes.c_int32), ("force_tile", ctypes.c_int32),
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
        self.ctx_create = _sig(c, "star_ctx_create", i32, i32, i32, ctypes.POINTER(vp))
        self.ctx_destroy = _sig(c, "star_ctx_destroy", None, vp)
        self.last_error = _sig(c, "star_last_error", ctypes.c_char_p, vp)
        self.set_stream = _sig(c, "star_set_stream", i32, vp, vp)
        self.sync = _sig(c, "star_sync", i32, vp)
        self.pool_bytes = _sig(c, "star_pool_bytes", sz, vp)
        self.pool_peak_bytes = _sig(c, "star_pool_peak_bytes", sz, vp)
        self.gemm = _sig(c, "star_gemm", i32, vp, ctypes.POINTER(GemmDesc))


_default_library = None


def default_library():
    global _default_library
    if _default_library is None:
        _default_library = Library()
    return _default_library


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class Context:
    """One star_ctx: a device, a compute dtype (fp16/bf16), a stream, a workspace pool."""

    def __init__(self, device=0, dtype=torch.float16, library=None):
        self.lib = library or default_library()
        if isinstance(device, torch.device):
            device = device.index or 0
        if not self.lib.is_hostemu and not torch.cuda.is_available():
            raise StarError("star_amd needs a ROCm GPU (gfx950); no CPU fallback exists for the HIP hot path")
        self.device_index = int(device)
        self.torch_device = torch.device("cpu") if self.lib.is_hostemu else torch.device("cuda", self.device_index)
        self.dtype = dtype
        h = ctypes.c_void_p()
        rc = self.lib.ctx_create(self.device_index, _TORCH2STAR[dtype], ctypes.byref(h))
        if rc:
            raise StarError(f"star_ctx_create failed (rc={rc})")
        self.h = h
        if not self.lib.is_hostemu:
            self.lib.set_stream(self.h, ctypes.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream))

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

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.torch_device)

    # ------------------------------------------------------------------ kernels
    def gemm(self, A, W, bias=None, res=None, out=None, *, mode=A_PLAIN, M=None, conv=None, temporal=None,
             geglu=False, out_f32=False, force_tile=0):
        """out[M, N] = epilogue(A' @ W^T).  A: [rows, lda] activations (channels-last tokens);
        W: [N, K]; conv=(NB,H,Wd,Cin,Ho,Wo,stride,pad_t,pad_l); temporal=(F,HW,Cin)."""
        self._chk_tensor(A, self.dtype); self._chk_tensor(W, self.dtype)
        self._chk_tensor(bias, torch.float32); self._chk_tensor(res, self.dtype)
        N, K = W.shape
        d = GemmDesc()
        d.mode = mode
        d.stride, d.pad_t, d.pad_l = 1, 1, 1
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
                (EPI_GEGLU if geglu else 0) | (EPI_OUT_F32 if out_f32 else 0)
        d.force_tile = force_tile
        self._check(self.lib.gemm(self.h, ctypes.byref(d)), "gemm")
        return out
