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
  else if constexpr (FORM == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));   // FORM 2: gemm_as.h's form -- B operand (the A panel) from an AGPR, accumulator in VGPRs
}
// KIND: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_pk_add_f16, 4 v_pk_mul_f32, 5 v_cvt_pk_f16_f32 (e64), 6 v_max3_f32, 7 ds_read_b128
// (LDS), 8 v_accvgpr_read_b32, 9 v_permlane32_swap
template <int KIND> __device__ __forceinline__ void valu(float& x, f32x2& p, float y, float z, const char* lds, f32x16& spare) {
  if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
  else if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p));
  else if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  else if constexpr (KIND == 3) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(y));
  else if constexpr (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(p));
  else if constexpr (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));
  else if constexpr (KIND == 6) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
  else if constexpr (KIND == 7) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((unsigned)(size_t)lds) : "memory");
    asm volatile("" :: "v"(r));
  }
  else if constexpr (KIND == 8) { float s0 = spare[0]; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(s0)); }
  else if constexpr (KIND == 9) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(p[0]));
  else if constexpr (KIND == 11) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));   // round 6: the round-toward-zero pack (one pass?)
  else {   // KIND 10: the epilogue's read pattern -- v_fma_f32 whose operand is an element of ANOTHER (idle) 16-register accumulator block in VGPRs
    float s0 = spare[3];
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(s0), "v"(y));
  }
}

template <int FORM, int KIND, int NV, int NT = 256>   // NT = 512: two waves per SIMD (does wave B's VALU overlap wave A's MFMA?)
__global__ void __launch_bounds__(NT, 1) probe(float* out, int iters) {
  __shared__ char lds_buf[65536];
  const char* lds = lds_buf + (threadIdx.x & 255) * 16;
  f32x16 spare;
  for (int r = 0; r < 16; ++r) spare[r] = 1.f;
  if (KIND == 8) { float s0 = spare[0]; asm volatile("" : "+a"(s0)); spare[0] = s0; }
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
      for (int k = 0; k < NV; ++k) valu<KIND>(x[k & 7], p[k & 7], y, z, lds, spare);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  for (int i = 0; i < 8; ++i) s += x[i] + p[i][0] + p[i][1];
  if (KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[threadIdx.x & 255] = s;
}

template <int FORM, int KIND, int NV, int NT = 256>
static void run(float* out, int iters, int nblocks) {
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0)); HCHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<FORM, KIND, NV, NT>), dim3(nblocks), dim3(NT), 0, 0, out, iters / 10);
  HCHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    HCHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<FORM, KIND, NV, NT>), dim3(nblocks), dim3(NT), 0, 0, out, iters);
    HCHECK(hipEventRecord(e1, 0));
    HCHECK(hipEventSynchronize(e1));
    float ms = 0; HCHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double ns_per_slot = best * 1e6 / ((double)iters * 4);
  printf("  NV=%2d: %7.2f ns per MFMA slot\n", NV, ns_per_slot);
}
template <int FORM, int KIND, int NT = 256>
static void sweep(float* out, int iters, int nblocks, const char* what) {
  printf("%s, %d workgroup(s) of %d threads\n", what, nblocks, NT);
  run<FORM, KIND, 0, NT>(out, iters, nblocks);
  run<FORM, KIND, 2, NT>(out, iters, nblocks);
  run<FORM, KIND, 4, NT>(out, iters, nblocks);
  run<FORM, KIND, 6, NT>(out, iters, nblocks);
  run<FORM, KIND, 8, NT>(out, iters, nblocks);
  run<FORM, KIND, 10, NT>(out, iters, nblocks);
  run<FORM, KIND, 12, NT>(out, iters, nblocks);
  run<FORM, KIND, 16, NT>(out, iters, nblocks);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const bool more = argc > 2 && atoi(argv[2]) != 0;    // second argument 1: the round-5 extension (more instruction kinds, two waves per SIMD)
  float* out;
  HCHECK(hipMalloc(&out, 256 * sizeof(float)));
  if (argc > 2 && atoi(argv[2]) == 2) {   // round 6: the two f32 -> packed f16 converts side by side (one wave and two waves per SIMD)
    sweep<0, 5>(out, iters, 1, "v_cvt_pk_f16_f32");
    sweep<0, 11>(out, iters, 1, "v_cvt_pkrtz_f16_f32");
    sweep<0, 5, 512>(out, iters, 1, "two waves per SIMD, v_cvt_pk_f16_f32");
    sweep<0, 11, 512>(out, iters, 1, "two waves per SIMD, v_cvt_pkrtz_f16_f32");
    return 0;
  }
  if (!more) {
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
  // one workgroup (boost clock), accumulators in AGPRs: the other instruction kinds a K / key loop carries
  sweep<0, 3>(out, iters, 1, "v_pk_add_f16");
  sweep<0, 4>(out, iters, 1, "v_pk_mul_f32");
  sweep<0, 5>(out, iters, 1, "v_cvt_pk_f16_f32");
  sweep<0, 6>(out, iters, 1, "v_max3_f32");
  sweep<0, 7>(out, iters, 1, "ds_read_b128");
  sweep<0, 8>(out, iters, 1, "v_accvgpr_read_b32");
  sweep<0, 9>(out, iters, 1, "v_permlane32_swap_b32");
  // two waves per SIMD: per-SIMD time per MFMA slot PAIR (both waves run the same loop); if wave B's VALU overlaps wave A's MFMA the
  // time stays at 2 x 34 cycles until the VALU of both waves fills the gaps
  sweep<0, 0, 512>(out, iters, 1, "two waves per SIMD, v_fma_f32");
  sweep<0, 1, 512>(out, iters, 1, "two waves per SIMD, v_pk_fma_f32");
  sweep<0, 2, 512>(out, iters, 1, "two waves per SIMD, v_exp_f32");
  // gemm_as.h's operand forms: B operand from an AGPR, accumulators in VGPRs; the epilogue's reads of an idle accumulator block
  sweep<2, 0>(out, iters, 1, "B operand from AGPR + accumulators in VGPRs, v_fma_f32");
  sweep<1, 10>(out, iters, 1, "accumulators in VGPRs, v_fma_f32 reading an idle accumulator block");
  sweep<2, 10>(out, iters, 1, "B operand from AGPR + accumulators in VGPRs, v_fma_f32 reading an idle accumulator block");
  sweep<0, 3, 512>(out, iters, 1, "two waves per SIMD, v_pk_add_f16");
  return 0;
}
