// mfma_valu_overlap -- does a wave's VALU work run in the shadow of ITS OWN MFMAs on gfx950 (one wave per SIMD)?
// One workgroup of 4 waves (one per SIMD, a single CU: no power throttling) loops over {1 MFMA 32x32x16 f16, NV independent VALU
// instructions} x 4 accumulators; time per MFMA slot against NV says whether the VALU issues ride in the MFMA's 8 passes (flat up to
// ~7 x 4-cycle issues) or serialise behind it (linear from NV = 1).  Accumulators in AGPRs ("a" form) or architectural VGPRs ("v" form,
// what -mllvm -amdgpu-mfma-vgpr-form produces for gemm_as.h / gemm_tq.h); VALU = v_fma_f32, v_pk_fma_f32 or v_exp_f32.
//   hipcc -O2 --offload-arch=gfx950 tools/probe/mfma_valu_overlap.hip -o tools/probe/mfma_valu_overlap && tools/probe/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define HCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int FORM> __device__ __forceinline__ void mfma(f32x16& acc, f16x8 a, f16x8 b) {
  if constexpr (FORM == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int KIND> __device__ __forceinline__ void valu(float& x, f32x2& p, float y, float z) {
  if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
  else if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
  else asm volatile("v_exp_f32 %0, %0" : "+v"(x));
}

template <int FORM, int KIND, int NV>
__global__ void __launch_bounds__(256, 1) probe(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
  float x[8]; f32x2 p[8];
  for (int i = 0; i < 8; ++i) { x[i] = 0.5f + 0.01f * i; p[i][0] = 0.25f; p[i][1] = 0.125f * i; }
  const float y = 0.999f, z = 0.001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma<FORM>(acc[i], a, b);
#pragma unroll
      for (int k = 0; k < NV; ++k) valu<KIND>(x[k & 7], p[k & 7], y, z);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += x[i] + p[i][0] + p[i][1];
  out[threadIdx.x] = s;
}

template <int FORM, int KIND, int NV>
static void run(float* out, int iters, int nblocks) {
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0)); HCHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<FORM, KIND, NV>), dim3(nblocks), dim3(256), 0, 0, out, iters / 10);
  HCHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    HCHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<FORM, KIND, NV>), dim3(nblocks), dim3(256), 0, 0, out, iters);
    HCHECK(hipEventRecord(e1, 0));
    HCHECK(hipEventSynchronize(e1));
    float ms = 0; HCHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double ns_per_slot = best * 1e6 / ((double)iters * 4);
  printf("  NV=%2d: %7.2f ns per MFMA slot\n", NV, ns_per_slot);
}
template <int FORM, int KIND>
static void sweep(float* out, int iters, int nblocks, const char* what) {
  printf("%s, %d workgroup(s)\n", what, nblocks);
  run<FORM, KIND, 0>(out, iters, nblocks);
  run<FORM, KIND, 2>(out, iters, nblocks);
  run<FORM, KIND, 4>(out, iters, nblocks);
  run<FORM, KIND, 6>(out, iters, nblocks);
  run<FORM, KIND, 8>(out, iters, nblocks);
  run<FORM, KIND, 10>(out, iters, nblocks);
  run<FORM, KIND, 12>(out, iters, nblocks);
  run<FORM, KIND, 16>(out, iters, nblocks);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out;
  HCHECK(hipMalloc(&out, 256 * sizeof(float)));
  for (int nb : {1, 256}) {
    sweep<0, 0>(out, iters, nb, "accumulators in AGPRs, VALU = v_fma_f32");
    sweep<1, 0>(out, iters, nb, "accumulators in VGPRs, VALU = v_fma_f32");
    sweep<0, 1>(out, iters, nb, "accumulators in AGPRs, VALU = v_pk_fma_f32");
    sweep<1, 1>(out, iters, nb, "accumulators in VGPRs, VALU = v_pk_fma_f32");
    sweep<0, 2>(out, iters, nb, "accumulators in AGPRs, VALU = v_exp_f32");
    sweep<1, 2>(out, iters, nb, "accumulators in VGPRs, VALU = v_exp_f32");
  }
  return 0;
}
