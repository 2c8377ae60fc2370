// Per-CU fill bandwidth probe (round 3): how fast can ONE workgroup per CU pull L2- / MALL-resident data, by path:
//   mode 0  global_load_lds_dwordx4 (LDS-DMA), rolling window of outstanding pieces per wave
//   mode 1  global_load_dwordx4 into VGPRs, same window
//   mode 2  both at once, half of the waves each
// 256 workgroups (one per CU) x W waves; every workgroup streams REPS times over the same `span` bytes (span = 1 MiB fits every
// XCD's L2; 64 MiB fits the MALL only; 1 GiB is HBM).  Prints aggregate TB/s and bytes per clock per CU at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int MODE, int WIN>
__global__ void __launch_bounds__(512) k_fill(const char* __restrict__ src, size_t span, int reps, unsigned* sink, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  // workgroup b starts at a different offset so that the 32 CUs of an XCD do not walk in lock step over the same lines
  const size_t per_wave = 1024;                       // bytes per wave-instruction
  const size_t pieces = span / per_wave;              // pieces in the span
  // the WORKGROUP walks one fixed piece sequence (start + 7 k) mod pieces, k = 0, 1, ...; wave w takes k = w, w + nw, ... : the
  // address stream a CU emits is the same for every wave count
  size_t p = ((size_t)blockIdx.x * 977 + (size_t)wave * 7) % pieces;
  u4 acc = {0, 0, 0, 0};
  const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + wave * (WIN * 1024);
  const long long t0 = __builtin_readcyclecounter();
  const long long total = (long long)reps * (long long)(pieces / nw);
  const bool dma = MODE == 0 || (MODE == 2 && (wave & 1) == 0);
  for (long long it = 0; it < total; it += WIN) {
#pragma unroll
    for (int w = 0; w < WIN; ++w) {
      const char* g = src + p * per_wave + lane * 16;
      if (dma) {
        const unsigned dst = ldsbase + w * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(g) : "memory", "m0");
      } else {
        u4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(g) : "memory");
        asm volatile("s_waitcnt vmcnt(%1)\n\tv_xor_b32 %0, %0, %2" : "+v"(acc[0]) : "n"(WIN - 1), "v"(v[0]) : "memory");
      }
      p += nw * 7; if (p >= pieces) p -= pieces;
    }
    if (dma) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WIN / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  if (acc[0] == 0x12345678u) sink[0] = acc[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int WIN>
static void run(const char* name, const char* src, size_t span, int waves, int reps) {
  unsigned* sink; long long* cyc;
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&cyc, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = (size_t)waves * WIN * 1024;
  CK(hipFuncSetAttribute((const void*)k_fill<MODE, WIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL((k_fill<MODE, WIN>), dim3(256), dim3(waves * 64), lds, 0, src, span, 1, sink, cyc);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_fill<MODE, WIN>), dim3(256), dim3(waves * 64), lds, 0, src, span, reps, sink, cyc);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const size_t pieces = span / 1024;
  const double bytes = 256.0 * waves * ((double)reps * (pieces / waves) / WIN) * WIN * 1024.0;
  printf("%-46s span %7.1f MiB  %6.2f TB/s  %6.1f B/clk/CU  (clock %.2f GHz)\n", name, span / 1048576.0, bytes / ms / 1e9,
         bytes / 256.0 / (double)c, (double)c / ms / 1e6);
  CK(hipFree(sink)); CK(hipFree(cyc));
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  const size_t big = (size_t)1 << 30;
  char* src; CK(hipMalloc(&src, big)); CK(hipMemset(src, 1, big));
  for (size_t span : {(size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)192 << 20}) {
    const int reps = (int)(((size_t)4 << 30) / span);
    run<0, 8>("LDS-DMA, 8 waves, window 8", src, span, 8, reps);
    run<0, 4>("LDS-DMA, 8 waves, window 4", src, span, 8, reps);
    run<0, 2>("LDS-DMA, 8 waves, window 2", src, span, 8, reps);
    run<0, 16>("LDS-DMA, 4 waves, window 16", src, span, 4, reps);
    run<0, 8>("LDS-DMA, 4 waves, window 8", src, span, 4, reps);
    run<0, 4>("LDS-DMA, 4 waves, window 4", src, span, 4, reps);
    run<0, 16>("LDS-DMA, 2 waves, window 16", src, span, 2, reps);
    run<0, 8>("LDS-DMA, 2 waves, window 8", src, span, 2, reps);
    run<0, 16>("LDS-DMA, 1 wave, window 16", src, span, 1, reps);
  }
  return 0;
}
