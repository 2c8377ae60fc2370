/* star_hip.h -- C ABI of libstar_hip.so: the MI355X (gfx950) implementation of
 * STAR's per-chunk denoising hot path.
 *
 * The reference (NJU-PCALab/STAR) has no FFI: its seam is the Python object
 * VideoToVideo_sr and the callables it wires together.
 * Each entry point below names the reference interface it replaces; the
 * reference-side binding is the ctypes stub shown in INTEGRATION.md
 * (star_amd/lib.py is that stub, shipped).
 *
 * Conventions: plain pointers and sizes only; all device pointers are
 * caller-owned (e.g. PyTorch-ROCm tensors) unless stated; every call enqueues
 * on the context's stream and returns 0 on success, non-zero on error with a
 * message available from star_last_error().  One host thread per context.
 * Activation layout is channels-last: [frames, H, W, C] == [tokens, C].
 */
#ifndef STAR_HIP_H_
#define STAR_HIP_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct star_ctx star_ctx;

enum { STAR_F16 = 0, STAR_BF16 = 1, STAR_F32 = 2 };

/* A-operand gather modes of star_gemm (see star_amd/csrc/gemm.h) */
enum { STAR_A_PLAIN = 0, STAR_A_CONV3X3 = 1, STAR_A_CONV3X3_UP = 2, STAR_A_TCONV3 = 3 };
/* epilogue flags of star_gemm */
enum { STAR_EPI_BIAS = 1, STAR_EPI_RES = 2, STAR_EPI_GEGLU = 4, STAR_EPI_OUT_F32 = 8 };

/* ---- context ------------------------------------------------------------ */
/* replaces: VideoToVideo_sr.__init__ device selection (video_to_video_model.py:21-34,42) */
int star_ctx_create(int device_id, int dtype, star_ctx** out);
void star_ctx_destroy(star_ctx* ctx);
const char* star_last_error(star_ctx* ctx);
int star_set_stream(star_ctx* ctx, void* hip_stream);   /* use the caller's hipStream_t */
int star_sync(star_ctx* ctx);
int star_is_hostemu(void);                               /* 1 only in the test-tooling emulator build */
size_t star_pool_bytes(star_ctx* ctx);
size_t star_pool_peak_bytes(star_ctx* ctx);

/* ---- kernel-level entry points (unit parity; each is one HIP kernel family) */
typedef struct star_gemm_desc {
  const void* A; const void* W; void* C; const float* bias; const void* res;
  int32_t M, N, K, lda, ldc, ldr;
  int32_t mode;                      /* STAR_A_* */
  int32_t H, Wd, Cin, Ho, Wo, stride, pad_t, pad_l;   /* conv geometry, NHWC */
  int32_t HW, F;                     /* temporal-conv geometry */
  int32_t epi;                       /* STAR_EPI_* */
  int32_t force_tile;                /* 0 = auto */
} star_gemm_desc;
/* replaces: nn.Linear / nn.Conv2d / nn.Conv3d(3,1,1) / nn.Conv1d(k=1) call sites
 * (unet_v2v.py:151-155,274,294,500,526,553,612,639,648,717,1005,1025,1209-1220) */
int star_gemm(star_ctx* ctx, const star_gemm_desc* d);

#ifdef __cplusplus
}
#endif
#endif /* STAR_HIP_H_ */
