// Hardware-semantics probe for gfx950 primitives used by star_amd kernels.
// Verifies (on a real MI355X) the lane layouts the kernels and the host
// emulator (tools/hostemu) assume: MFMA 32x32x16 / 16x16x32 (f16, bf16),
// global_load_lds 16B, ds_read_b64_tr_b16, v_permlane32_swap, f64 atomics.
// Build: hipcc --offload-arch=gfx950 -O2 probe.hip -o probe ; run on GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
#define LDS3 __attribute__((address_space(3)))
#define GLB1 __attribute__((address_space(1)))

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <typename T8, int BF>
__global__ void k_mfma32(const T8* a, const T8* b, f16v* c) {
  int l = threadIdx.x;
  f16v acc = {0};
  if constexpr (BF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[l], b[l], acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[l], b[l], acc, 0, 0, 0);
  c[l] = acc;
}
template <typename T8, int BF>
__global__ void k_mfma16(const T8* a, const T8* b, f4v* c) {
  int l = threadIdx.x;
  f4v acc = {0};
  if constexpr (BF) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[l], b[l], acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[l], b[l], acc, 0, 0, 0);
  c[l] = acc;
}

// glds: 4 waves, each wave copies 1 KiB chunk with a per-lane permuted source.
__global__ void k_glds(const char* g, const int* srcperm, char* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int t = threadIdx.x, w = t >> 6, l = t & 63;
  // wave w writes LDS [w*2048 + 1024, +1024) (non-trivial base); lane source = chunk srcperm[t]
  __builtin_amdgcn_global_load_lds((const GLB1 void*)(g + (size_t)srcperm[t] * 16),
                                   (LDS3 void*)(smem + w * 2048 + 1024), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) etc
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = t; i < 8192 / 16; i += blockDim.x)
    ((uint4*)out)[i] = ((uint4*)smem)[i];
}

// tr-read: lane supplies arbitrary 8B-aligned LDS address addr[l].
__global__ void k_tr(const unsigned short* fill, const int* addr, s4* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) ((unsigned short*)smem)[i] = fill[i];
  __syncthreads();
  s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS3 s4*)(smem + addr[l]));
  out[l] = r;
}

__global__ void k_perm(const int* a, const int* b, int* oa, int* ob) {
  int l = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a[l], b[l], false, false);
  oa[l] = r[0]; ob[l] = r[1];
}

__global__ void k_atomic(double* d, float* f) {
  atomicAdd(d, 1.0 + threadIdx.x * 1e-9);
  atomicAdd(f, 1.0f);
}

// bandwidth: float4 copy
__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

template <int BF>
__global__ void __launch_bounds__(256) k_peak(float* out, int iters) {
  h8 a; b8 ab;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); ab[j] = (__bf16)(0.001f * (threadIdx.x + j)); }
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
    if constexpr (BF) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
  if (s == 12345.678f) out[0] = s;
}

static float bf16_to_f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f_to_bf16(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

template <typename T>
static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s arch=%s CUs=%d clock=%d kHz mem=%.1f GB lds/block=%zu\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, prop.clockRate, prop.totalGlobalMem / 1e9, prop.sharedMemPerBlock);
  int fails = 0;
  // ---------------- MFMA 32x32x16 ----------------
  for (int bf = 0; bf < 2; ++bf) {
    // logical A[32][16], B[16][32] small integers; assumed layout:
    //   A: lane l holds A[l&31][8*(l>>5)+j];  B: lane l holds B[8*(l>>5)+j][l&31]
    //   C: lane l reg r holds C[(r&3)+8*(r>>2)+4*(l>>5)][l&31]
    float A[32][16], B[16][32], C[32][32];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i][k] = (float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) B[k][n] = (float)((k * 5 + n * 13) % 9 - 4);
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i][k] * B[k][n]; C[i][n] = s; }
    std::vector<unsigned short> ha(64 * 8), hb(64 * 8);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
      float av = A[l & 31][8 * (l >> 5) + j], bv = B[8 * (l >> 5) + j][l & 31];
      if (bf) { ha[l * 8 + j] = f_to_bf16(av); hb[l * 8 + j] = f_to_bf16(bv); }
      else { _Float16 x = (_Float16)av, y = (_Float16)bv; memcpy(&ha[l * 8 + j], &x, 2); memcpy(&hb[l * 8 + j], &y, 2); }
    }
    unsigned short *da = dalloc<unsigned short>(512), *db = dalloc<unsigned short>(512); float* dc = dalloc<float>(64 * 16);
    CK(hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice));
    if (bf) k_mfma32<b8, 1><<<1, 64>>>((b8*)da, (b8*)db, (f16v*)dc); else k_mfma32<h8, 0><<<1, 64>>>((h8*)da, (h8*)db, (f16v*)dc);
    CK(hipDeviceSynchronize());
    std::vector<float> hc(64 * 16); CK(hipMemcpy(hc.data(), dc, 64 * 16 * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31; if (hc[l * 16 + r] != C[row][col]) ++bad; }
    printf("mfma_32x32x16_%s assumed layout: %s (%d mismatches)\n", bf ? "bf16" : "f16", bad ? "FAIL" : "OK", bad); fails += bad != 0;
    if (bad) { for (int l = 0; l < 64; l += 9) { printf(" lane %d:", l); for (int r = 0; r < 16; ++r) printf(" %g", hc[l * 16 + r]); printf("\n"); } }
  }
  // ---------------- MFMA 16x16x32 ----------------
  for (int bf = 0; bf < 2; ++bf) {
    // assumed: A: lane l holds A[l&15][8*(l>>4)+j]; B: B[8*(l>>4)+j][l&15]; C: lane l reg r: C[4*(l>>4)+r][l&15]
    float A[16][32], B[32][16], C[16][16];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i][k] = (float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) B[k][n] = (float)((k * 5 + n * 13) % 9 - 4);
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i][k] * B[k][n]; C[i][n] = s; }
    std::vector<unsigned short> ha(512), hb(512);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
      float av = A[l & 15][8 * (l >> 4) + j], bv = B[8 * (l >> 4) + j][l & 15];
      if (bf) { ha[l * 8 + j] = f_to_bf16(av); hb[l * 8 + j] = f_to_bf16(bv); }
      else { _Float16 x = (_Float16)av, y = (_Float16)bv; memcpy(&ha[l * 8 + j], &x, 2); memcpy(&hb[l * 8 + j], &y, 2); }
    }
    unsigned short *da = dalloc<unsigned short>(512), *db = dalloc<unsigned short>(512); float* dc = dalloc<float>(64 * 4);
    CK(hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice));
    if (bf) k_mfma16<b8, 1><<<1, 64>>>((b8*)da, (b8*)db, (f4v*)dc); else k_mfma16<h8, 0><<<1, 64>>>((h8*)da, (h8*)db, (f4v*)dc);
    CK(hipDeviceSynchronize());
    std::vector<float> hc(256); CK(hipMemcpy(hc.data(), dc, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (hc[l * 4 + r] != C[4 * (l >> 4) + r][l & 15]) ++bad;
    printf("mfma_16x16x32_%s assumed layout: %s (%d mismatches)\n", bf ? "bf16" : "f16", bad ? "FAIL" : "OK", bad); fails += bad != 0;
  }
  // ---------------- global_load_lds ----------------
  {
    std::vector<unsigned> hg(256 * 4 * 2); for (size_t i = 0; i < hg.size(); ++i) hg[i] = (unsigned)i;  // 512 chunks of 16B
    std::vector<int> perm(256); for (int t = 0; t < 256; ++t) perm[t] = (t * 37 + 11) % 512;
    unsigned* dg = dalloc<unsigned>(hg.size()); int* dp = dalloc<int>(256); char* dout = dalloc<char>(8192);
    CK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, perm.data(), 1024, hipMemcpyHostToDevice));
    k_glds<<<1, 256, 8192>>>((const char*)dg, dp, dout); CK(hipDeviceSynchronize());
    std::vector<unsigned> ho(2048); CK(hipMemcpy(ho.data(), dout, 8192, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int t = 0; t < 256; ++t) { int w = t >> 6, l = t & 63; for (int j = 0; j < 4; ++j) { unsigned got = ho[(w * 2048 + 1024 + l * 16) / 4 + j], exp = (unsigned)(perm[t] * 4 + j); if (got != exp) ++bad; } }
    printf("global_load_lds 16B (dst = wave base + lane*16, per-lane src): %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad); fails += bad != 0;
  }
  // ---------------- ds_read_b64_tr_b16 ----------------
  {
    std::vector<unsigned short> fill(4096); for (int i = 0; i < 4096; ++i) fill[i] = (unsigned short)i;
    for (int variant = 0; variant < 2; ++variant) {
      std::vector<int> addr(64);
      for (int l = 0; l < 64; ++l) addr[l] = variant == 0 ? l * 8 : (((l * 29 + 7) % 64) * 72 + ((l * 5) % 8) * 8);  // 8B aligned
      unsigned short* df = dalloc<unsigned short>(4096); int* da = dalloc<int>(64); s4* dout = dalloc<s4>(64);
      CK(hipMemcpy(df, fill.data(), 8192, hipMemcpyHostToDevice)); CK(hipMemcpy(da, addr.data(), 256, hipMemcpyHostToDevice));
      k_tr<<<1, 64, 8192>>>(df, da, dout); CK(hipDeviceSynchronize());
      std::vector<short> ho(256); CK(hipMemcpy(ho.data(), dout, 512, hipMemcpyDeviceToHost));
      // hypothesis: within each 16-lane group, lane i elem j = P[4*j + i/4][i%4], P[x] = the 4 halves at lane x's address
      int bad = 0;
      for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        int g = l & ~15, i = l & 15; int src_lane = g + 4 * j + i / 4; int e = i % 4;
        unsigned short exp = fill[addr[src_lane] / 2 + e];
        if ((unsigned short)ho[l * 4 + j] != exp) ++bad;
      }
      printf("ds_read_b64_tr_b16 variant %d hypothesis R[i][j]=P[4j+i/4][i%%4]: %s (%d mismatches)\n", variant, bad ? "FAIL" : "OK", bad); fails += bad != 0;
      if (bad || variant == 0) { printf("  raw (canonical addr l*8): lane: 4 values (as 16-bit element indices)\n"); for (int l = 0; l < 64; ++l) { printf("  l%02d a=%4d: %4d %4d %4d %4d\n", l, addr[l], (unsigned short)ho[l * 4], (unsigned short)ho[l * 4 + 1], (unsigned short)ho[l * 4 + 2], (unsigned short)ho[l * 4 + 3]); } }
    }
  }
  // ---------------- permlane32_swap ----------------
  {
    std::vector<int> a(64), b(64); for (int l = 0; l < 64; ++l) { a[l] = 100 + l; b[l] = 200 + l; }
    int *da = dalloc<int>(64), *db = dalloc<int>(64), *oa = dalloc<int>(64), *ob = dalloc<int>(64);
    CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    k_perm<<<1, 64>>>(da, db, oa, ob); CK(hipDeviceSynchronize());
    std::vector<int> ra(64), rb(64); CK(hipMemcpy(ra.data(), oa, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), ob, 256, hipMemcpyDeviceToHost));
    // hypothesis: r0[l<32]=a[l], r0[l>=32]=b[l-32]; r1[l<32]=a[l+32], r1[l>=32]=b[l]
    int bad = 0;
    for (int l = 0; l < 64; ++l) { int e0 = l < 32 ? a[l] : b[l - 32], e1 = l < 32 ? a[l + 32] : b[l]; if (ra[l] != e0 || rb[l] != e1) ++bad; }
    printf("permlane32_swap hypothesis: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad); fails += bad != 0;
    if (bad) { for (int l = 0; l < 64; ++l) printf("  l%02d r0=%d r1=%d\n", l, ra[l], rb[l]); }
  }
  // ---------------- atomics ----------------
  {
    double* dd = dalloc<double>(1); float* df = dalloc<float>(1);
    k_atomic<<<64, 256>>>(dd, df); CK(hipDeviceSynchronize());
    double hd; float hf; CK(hipMemcpy(&hd, dd, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hf, df, 4, hipMemcpyDeviceToHost));
    printf("atomicAdd f64 sum=%.6f (expect ~16384.002) f32 sum=%.1f\n", hd, hf);
  }
  // ---------------- HBM copy bandwidth ----------------
  {
    size_t n = (size_t)1 << 28;  // 256M float4 = 4 GiB in, 4 GiB out
    float4 *in, *out; CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMemset(in, 1, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0)); k_copy<<<256 * 8, 256>>>(in, out, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("copy 4GiB+4GiB: %.3f ms -> %.2f TB/s\n", ms, 2.0 * n * 16 / ms / 1e9);
    }
    CK(hipFree(in)); CK(hipFree(out));
  }
  // ---------------- MFMA peak ----------------
  {
    float* d = dalloc<float>(4); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int bf = 0; bf < 2; ++bf) for (int rep = 0; rep < 2; ++rep) {
      int iters = 20000; CK(hipEventRecord(e0));
      if (bf) k_peak<1><<<256 * 8, 256>>>(d, iters); else k_peak<0><<<256 * 8, 256>>>(d, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double flops = 2.0 * 32 * 32 * 16 * 4.0 * iters * (256.0 * 8 * 4);
      printf("mfma_32x32x16 %s peak loop: %.3f ms -> %.1f TFLOP/s\n", bf ? "bf16" : "f16", ms, flops / ms / 1e9);
    }
  }
  printf("PROBE %s (%d failing checks)\n", fails ? "HAS FAILURES" : "ALL OK", fails);
  return 0;
}
