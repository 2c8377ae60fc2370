// Issue-overlap probe 2 (round 3) for gfx950, one wave per SIMD: what an LDS fragment read, a transpose read and a mixed
// softmax filler set cost beside v_mfma_f32_32x32x16_f16, with the MFMA accumulators in architectural VGPRs or in AGPRs and the
// A operand in a VGPR or an AGPR.  Build: hipcc --offload-arch=gfx950 -O2 overlap2.hip -o overlap2.  Prints shader cycles per
// loop iteration (4 MFMAs = 128 matrix-pipe cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define FMA2 "v_fma_f32 %[f0], %[f0], %[c], %[c]\n" "v_fma_f32 %[f1], %[f1], %[c], %[c]\n"
#define EXP2 "v_exp_f32 %[f2], %[f2]\n" "v_exp_f32 %[f3], %[f3]\n"
#define CVPK "v_cvt_pk_f16_f32 %[f4], %[f2], %[f3]\n" "v_pk_add_f16 %[f5], %[f5], %[f6]\n"
#define MF "v_mfma_f32_32x32x16_f16 %[a0], %[x], %[y], %[a0]\n"
#define MG "v_mfma_f32_32x32x16_f16 %[a1], %[x], %[y], %[a1]\n"
#define MFA "v_mfma_f32_32x32x16_f16 %[a0], %[xa], %[y], %[a0]\n"   // A operand from an AGPR
#define MGA "v_mfma_f32_32x32x16_f16 %[a1], %[xa], %[y], %[a1]\n"
#define LD0 "ds_read_b128 %[l0], %[ad]\n"
#define LD1 "ds_read_b128 %[l1], %[ad] offset:4096\n"
#define LD2 "ds_read_b128 %[l2], %[ad] offset:8192\n"
#define LD3 "ds_read_b128 %[l3], %[ad] offset:12288\n"
#define TR0 "ds_read_b64_tr_b16 %[t0], %[ad]\n" "ds_read_b64_tr_b16 %[t1], %[ad] offset:2048\n"
#define TR1 "ds_read_b64_tr_b16 %[t2], %[ad] offset:4096\n" "ds_read_b64_tr_b16 %[t3], %[ad] offset:6144\n"
#define WAITL "s_waitcnt lgkmcnt(0)\n"

#define BODYV(ASM)                                                                                           \
  asm volatile(ASM                                                                                           \
               : [a0] "+v"(acc0), [a1] "+v"(acc1), [f0] "+v"(f[0]), [f1] "+v"(f[1]), [f2] "+v"(f[2]),        \
                 [f3] "+v"(f[3]), [f4] "+v"(f[4]), [f5] "+v"(f[5]), [f6] "+v"(f[6]), [l0] "=&v"(l0),         \
                 [l1] "=&v"(l1), [l2] "=&v"(l2), [l3] "=&v"(l3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3) \
               : [x] "v"(x), [y] "v"(y), [c] "v"(c), [ad] "v"(ad), [xa] "a"(xa) : "memory")
#define BODYA(ASM)                                                                                           \
  asm volatile(ASM                                                                                           \
               : [a0] "+a"(acc0), [a1] "+a"(acc1), [f0] "+v"(f[0]), [f1] "+v"(f[1]), [f2] "+v"(f[2]),        \
                 [f3] "+v"(f[3]), [f4] "+v"(f[4]), [f5] "+v"(f[5]), [f6] "+v"(f[6]), [l0] "=&v"(l0),         \
                 [l1] "=&v"(l1), [l2] "=&v"(l2), [l3] "=&v"(l3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3) \
               : [x] "v"(x), [y] "v"(y), [c] "v"(c), [ad] "v"(ad), [xa] "a"(xa) : "memory")
#define BODYLA(ASM)   /* LDS read destinations in AGPRs */                                                   \
  asm volatile(ASM                                                                                           \
               : [a0] "+v"(acc0), [a1] "+v"(acc1), [f0] "+v"(f[0]), [f1] "+v"(f[1]), [f2] "+v"(f[2]),        \
                 [f3] "+v"(f[3]), [f4] "+v"(f[4]), [f5] "+v"(f[5]), [f6] "+v"(f[6]), [l0] "=&a"(l0),         \
                 [l1] "=&a"(l1), [l2] "=&a"(l2), [l3] "=&a"(l3), [t0] "=&a"(t0), [t1] "=&a"(t1), [t2] "=&a"(t2), [t3] "=&a"(t3) \
               : [x] "v"(x), [y] "v"(y), [c] "v"(c), [ad] "v"(ad), [xa] "a"(xa) : "memory")

template <int MODE>
__global__ void __launch_bounds__(256) k_overlap(long long* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((float*)lds)[i] = 0.001f * i;
  h8 x, y, xa;
  for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(0.001f * (lane + e)); y[e] = (_Float16)(0.002f * (lane - e)); xa[e] = x[e]; }
  f16v acc0 = {0}, acc1 = {0};
  float f[8];
  for (int e = 0; e < 8; ++e) f[e] = 0.5f + 0.01f * e;
  const float c = 0.999f;
  f4v l0, l1, l2, l3;
  f2v t0, t1, t2, t3;
  const unsigned ad = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + lane * 16 + wave * 1024;
  __syncthreads();
  const long long t_0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) BODYV(MF MG MF MG);
    if (MODE == 1) BODYA(MF MG MF MG);
    if (MODE == 2) BODYV(MFA MGA MFA MGA);
    if (MODE == 3) BODYV(MF LD0 MG LD1 MF LD2 MG LD3 WAITL);
    if (MODE == 4) BODYA(MF LD0 MG LD1 MF LD2 MG LD3 WAITL);
    if (MODE == 5) BODYLA(MF LD0 MG LD1 MF LD2 MG LD3 WAITL);
    if (MODE == 6) BODYV(MF TR0 MG TR1 MF TR0 MG TR1 WAITL);
    if (MODE == 7) BODYA(MF TR0 MG TR1 MF TR0 MG TR1 WAITL);
    if (MODE == 8) BODYV(MF EXP2 FMA2 MG EXP2 FMA2 MF EXP2 FMA2 MG EXP2 FMA2);
    if (MODE == 9) BODYA(MF EXP2 FMA2 MG EXP2 FMA2 MF EXP2 FMA2 MG EXP2 FMA2);
    if (MODE == 10) BODYV(MF EXP2 CVPK MG EXP2 CVPK MF EXP2 CVPK MG EXP2 CVPK);
    if (MODE == 11) BODYV(MF EXP2 FMA2 LD0 MG EXP2 FMA2 LD1 MF EXP2 FMA2 LD2 MG EXP2 FMA2 LD3 WAITL);
    if (MODE == 12) BODYA(MF EXP2 FMA2 LD0 MG EXP2 FMA2 LD1 MF EXP2 FMA2 LD2 MG EXP2 FMA2 LD3 WAITL);
    if (MODE == 13) BODYLA(MF EXP2 FMA2 LD0 MG EXP2 FMA2 LD1 MF EXP2 FMA2 LD2 MG EXP2 FMA2 LD3 WAITL);
    if (MODE == 14) BODYV(LD0 LD1 LD2 LD3 WAITL);
    if (MODE == 15) BODYV(MF LD0 LD1 MG LD2 LD3 MF MG WAITL);
    if (MODE == 16) BODYV(MF MG MF MG LD0 LD1 LD2 LD3 WAITL);
    if (MODE == 17) BODYV(MF EXP2 FMA2 "s_nop 0\n" MG EXP2 FMA2 "s_nop 0\n" MF EXP2 FMA2 "s_nop 0\n" MG EXP2 FMA2 "s_nop 0\n");
    if (MODE == 18) BODYV(MF "s_barrier\n" MG MF MG);
  }
  const long long t_1 = __builtin_readcyclecounter();
  float sink = f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + acc0[0] + acc1[0] + l0[0] + l1[0] + l2[0] + l3[0] + t0[0] + t1[0] + t2[0] + t3[0];
  if (sink == 12345.678f) out[1023] = 1;
  if (lane == 0 && blockIdx.x == 0) out[wave] = t_1 - t_0;
}

template <int MODE>
static void run(const char* name) {
  long long* d;
  CK(hipMalloc(&d, 1024 * sizeof(long long)));
  const int iters = 20000;
  hipLaunchKernelGGL(k_overlap<MODE>, dim3(256), dim3(256), 0, 0, d, 100);
  hipLaunchKernelGGL(k_overlap<MODE>, dim3(256), dim3(256), 0, 0, d, iters);
  CK(hipDeviceSynchronize());
  std::vector<long long> h(8);
  CK(hipMemcpy(h.data(), d, 8 * sizeof(long long), hipMemcpyDeviceToHost));
  printf("%-72s wave0 %7.1f  wave3 %7.1f\n", name, (double)h[0] / iters, (double)h[3] / iters);
  CK(hipFree(d));
}

int main() {
  printf("one wave per SIMD; shader cycles per iteration (4 MFMAs = 128 matrix-pipe cycles)\n");
  run<0>("4 MFMA, C/D in VGPRs");
  run<1>("4 MFMA, C/D in AGPRs");
  run<2>("4 MFMA, A operand from an AGPR");
  run<3>("4 x (MFMA + ds_read_b128), C/D VGPR, read -> VGPR");
  run<4>("4 x (MFMA + ds_read_b128), C/D AGPR, read -> VGPR");
  run<5>("4 x (MFMA + ds_read_b128), C/D VGPR, read -> AGPR");
  run<6>("4 x (MFMA + 2 ds_read_b64_tr_b16), C/D VGPR");
  run<7>("4 x (MFMA + 2 ds_read_b64_tr_b16), C/D AGPR");
  run<8>("4 x (MFMA + 2 exp + 2 fma), C/D VGPR");
  run<9>("4 x (MFMA + 2 exp + 2 fma), C/D AGPR");
  run<10>("4 x (MFMA + 2 exp + cvt_pk + pk_add), C/D VGPR");
  run<11>("4 x (MFMA + 2 exp + 2 fma + ds_read_b128), C/D VGPR");
  run<12>("4 x (MFMA + 2 exp + 2 fma + ds_read_b128), C/D AGPR");
  run<13>("4 x (MFMA + 2 exp + 2 fma + ds_read_b128 -> AGPR), C/D VGPR");
  run<14>("4 ds_read_b128 alone + wait");
  run<15>("MFMA 2 reads MFMA 2 reads MFMA MFMA + wait");
  run<16>("4 MFMA then 4 ds_read_b128 + wait");
  run<17>("4 x (MFMA + 2 exp + 2 fma + s_nop 0), C/D VGPR");
  run<18>("4 MFMA with one s_barrier");
  return 0;
}
