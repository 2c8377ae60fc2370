// Issue-overlap probe for gfx950: how much VALU work hides behind v_mfma_f32_32x32x16_f16,
//   (a) inside one wave (k independent fillers after every MFMA), and
//   (b) across the two waves that share a SIMD (one MFMA-only, one VALU-only).
// Build: hipcc --offload-arch=gfx950 -O2 overlap.hip -o overlap ; run on the GPU box.  Prints cycles per loop iteration
// (s_memtime, 100 MHz-independent: wall clock of the shader clock domain) for one wave per SIMD unless stated.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define FMA1 "v_fma_f32 %[f0], %[f0], %[c], %[c]\n"
#define FMA2 FMA1 "v_fma_f32 %[f1], %[f1], %[c], %[c]\n"
#define FMA4 FMA2 "v_fma_f32 %[f2], %[f2], %[c], %[c]\n" "v_fma_f32 %[f3], %[f3], %[c], %[c]\n"
#define FMA6 FMA4 "v_fma_f32 %[f4], %[f4], %[c], %[c]\n" "v_fma_f32 %[f5], %[f5], %[c], %[c]\n"
#define FMA8 FMA6 "v_fma_f32 %[f6], %[f6], %[c], %[c]\n" "v_fma_f32 %[f7], %[f7], %[c], %[c]\n"
#define EXP1 "v_exp_f32 %[f0], %[f0]\n"
#define EXP2 EXP1 "v_exp_f32 %[f1], %[f1]\n"
#define EXP4 EXP2 "v_exp_f32 %[f2], %[f2]\n" "v_exp_f32 %[f3], %[f3]\n"
#define EXP6 EXP4 "v_exp_f32 %[f4], %[f4]\n" "v_exp_f32 %[f5], %[f5]\n"
#define CVT4 "v_cvt_pk_f16_f32 %[f0], %[f1], %[f2]\n" "v_cvt_pk_f16_f32 %[f3], %[f4], %[f5]\n" "v_cvt_pk_f16_f32 %[f6], %[f1], %[f2]\n" "v_cvt_pk_f16_f32 %[f7], %[f4], %[f5]\n"
#define NONE ""
// two independent accumulators so that consecutive MFMAs never wait on each other's result
#define MF "v_mfma_f32_32x32x16_f16 %[a0], %[x], %[y], %[a0]\n"
#define MG "v_mfma_f32_32x32x16_f16 %[a1], %[x], %[y], %[a1]\n"

#define BODY(ASM)                                                                                          \
  asm volatile(ASM                                                                                         \
               : [a0] "+v"(acc0), [a1] "+v"(acc1), [f0] "+v"(f[0]), [f1] "+v"(f[1]), [f2] "+v"(f[2]),      \
                 [f3] "+v"(f[3]), [f4] "+v"(f[4]), [f5] "+v"(f[5]), [f6] "+v"(f[6]), [f7] "+v"(f[7])       \
               : [x] "v"(x), [y] "v"(y), [c] "v"(c))

template <int MODE>
__global__ void __launch_bounds__(512) k_overlap(long long* out, int iters, int role_split) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  h8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(0.001f * (lane + e)); y[e] = (_Float16)(0.002f * (lane - e)); }
  f16v acc0 = {0}, acc1 = {0};
  float f[8];
  for (int e = 0; e < 8; ++e) f[e] = 0.5f + 0.01f * e;
  const float c = 0.999f;
  // role_split: waves >= 4 (the second wave of every SIMD) run the VALU-only body, waves < 4 the MFMA-only body
  const bool valu_role = role_split && wave >= 4;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) BODY(MF MG MF MG);                                  // 4 bare MFMAs
    if (MODE == 1) BODY(MF FMA2 MG FMA2 MF FMA2 MG FMA2);              // + 2 fma per gap
    if (MODE == 2) BODY(MF FMA4 MG FMA4 MF FMA4 MG FMA4);
    if (MODE == 3) BODY(MF FMA6 MG FMA6 MF FMA6 MG FMA6);
    if (MODE == 4) BODY(MF FMA8 MG FMA8 MF FMA8 MG FMA8);
    if (MODE == 5) BODY(MF EXP2 MG EXP2 MF EXP2 MG EXP2);
    if (MODE == 6) BODY(MF EXP4 MG EXP4 MF EXP4 MG EXP4);
    if (MODE == 7) BODY(MF EXP6 MG EXP6 MF EXP6 MG EXP6);
    if (MODE == 8) BODY(MF CVT4 MG CVT4 MF CVT4 MG CVT4);
    if (MODE == 9) BODY(FMA8 FMA8 FMA8 FMA8);                          // 32 bare fma
    if (MODE == 10) BODY(EXP6 EXP6 EXP6 EXP6);                         // 24 bare exp
    if (MODE == 11) {                                                  // cross-wave: MFMA-only beside fma-only
      if (valu_role) BODY(FMA8 FMA8 FMA8 FMA8); else BODY(MF MG MF MG);
    }
    if (MODE == 12) {                                                  // cross-wave: MFMA-only beside exp-only
      if (valu_role) BODY(EXP6 EXP6 EXP6 EXP6); else BODY(MF MG MF MG);
    }
    if (MODE == 13) BODY(MF MG MF MG FMA8 FMA8 FMA8 FMA8);             // same work as mode 4, phases not interleaved
    if (MODE == 14) {                                                  // cross-wave with the MFMA wave at raised priority
      if (valu_role) BODY(FMA8 FMA8 FMA8 FMA8); else BODY("s_setprio 3\n" MF MG MF MG);
    }
    if (MODE == 15) BODY(MF MG MF MG MF MG MF MG MF MG MF MG MF MG MF MG FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8);   // 16 MFMA then 64 fma
    if (MODE == 16) BODY(MF MG MF MG MF MG MF MG MF MG MF MG MF MG MF MG EXP6 FMA8 EXP6 FMA8 EXP6 FMA8 EXP6 FMA8 FMA8);  // 16 MFMA then 24 exp + 40 fma
    if (MODE == 17) {                                                  // phase streams with priority raised for the MFMA burst only
      BODY("s_setprio 3\n" MF MG MF MG MF MG MF MG MF MG MF MG MF MG MF MG "s_setprio 0\n" FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float sink = f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7] + acc0[0] + acc1[0];
  if (sink == 12345.678f) out[1023] = 1;
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int threads, int role_split) {
  long long* d;
  CK(hipMalloc(&d, 1024 * sizeof(long long)));
  const int iters = 20000;
  hipLaunchKernelGGL(k_overlap<MODE>, dim3(256), dim3(threads), 0, 0, d, 100, role_split);
  hipLaunchKernelGGL(k_overlap<MODE>, dim3(256), dim3(threads), 0, 0, d, iters, role_split);
  CK(hipDeviceSynchronize());
  std::vector<long long> h(8);
  CK(hipMemcpy(h.data(), d, 8 * sizeof(long long), hipMemcpyDeviceToHost));
  const int nw = threads / 64;
  printf("%-58s", name);
  for (int w = 0; w < nw; w += (nw > 4 ? 4 : 3)) printf("  wave%d %7.1f", w, (double)h[w] / iters);
  printf("   (s_memtime ticks per iteration; 4 MFMAs = 128 matrix-pipe cycles)\n");
  CK(hipFree(d));
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device %s, %d CUs, clock %d kHz, wall-clock rate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.clockInstructionRate);
  printf("---- one wave per SIMD (256 threads)\n");
  run<0>("4 MFMA", 256, 0);
  run<1>("4 x (MFMA + 2 fma)", 256, 0);
  run<2>("4 x (MFMA + 4 fma)", 256, 0);
  run<3>("4 x (MFMA + 6 fma)", 256, 0);
  run<4>("4 x (MFMA + 8 fma)", 256, 0);
  run<13>("4 MFMA then 32 fma (not interleaved)", 256, 0);
  run<5>("4 x (MFMA + 2 exp)", 256, 0);
  run<6>("4 x (MFMA + 4 exp)", 256, 0);
  run<7>("4 x (MFMA + 6 exp)", 256, 0);
  run<8>("4 x (MFMA + 4 cvt_pk)", 256, 0);
  run<9>("32 fma", 256, 0);
  run<10>("24 exp", 256, 0);
  printf("---- two waves per SIMD (512 threads): both waves run the same body\n");
  run<0>("4 MFMA | 4 MFMA", 512, 0);
  run<4>("4 x (MFMA + 8 fma) | same", 512, 0);
  run<9>("32 fma | 32 fma", 512, 0);
  printf("---- two waves per SIMD: waves 0-3 MFMA-only, waves 4-7 VALU-only\n");
  run<11>("4 MFMA | 32 fma", 512, 1);
  run<12>("4 MFMA | 24 exp", 512, 1);
  run<14>("4 MFMA at s_setprio 3 | 32 fma", 512, 1);
  printf("---- attention-like phase streams: 16 MFMA (512 pipe cycles) then 64 VALU, every wave the same program\n");
  run<15>("1 wave/SIMD : 16 MFMA then 64 fma", 256, 0);
  run<15>("2 waves/SIMD: 16 MFMA then 64 fma", 512, 0);
  run<16>("1 wave/SIMD : 16 MFMA then 24 exp + 40 fma", 256, 0);
  run<16>("2 waves/SIMD: 16 MFMA then 24 exp + 40 fma", 512, 0);
  run<17>("2 waves/SIMD: 16 MFMA (prio 3) then 64 fma (prio 0)", 512, 0);
  return 0;
}
