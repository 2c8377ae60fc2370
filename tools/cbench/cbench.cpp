// cbench -- a torch-free timing harness over the C ABI (include/star_hip.h): dlopen()s a build of the library, fills device
// buffers with N(0,1)-like 16-bit operands and times kernel launches with HIP events on the context's stream.  A fresh GPU box
// spends 1-2 minutes paging in `import torch`; this binary starts in a second, so a kernel A/B costs seconds of the GPU budget.
//
//   hipcc -O2 --offload-arch=gfx950 tools/cbench/cbench.cpp -o tools/cbench/cbench -ldl          (the GPU build)
//   g++ -O2 -DCBENCH_EMU tools/cbench/cbench.cpp -o tools/cbench/cbench_emu -ldl                 (against libstar_emu.so: plumbing check)
//   cbench <lib.so> <f16|bf16> <spec file | -> [reps]
// spec lines (# comments):
//   gemm M N K epi tile[,tile..]   plain-A star_gemm (one line per tile, same operands); epi = STAR_EPI_* bits (1 bias, 2 residual, 4 GEGLU, 32 folded LayerNorm), tile = force_tile
//   conv NB H W Cin Cout tile      3x3 conv, stride 1, pad 1, bias
//   tconv F HW C tile              temporal conv (3,1,1), bias + residual
//   attn B heads Nq Nk [v,v..]     star_attn_fwd (d = 64), K / V per batch; optional list of variant ids (default 9 = product), every
//                                  variant after the first compared bit for bit with the first one's output
//   tq F HW                        star_temporal_qkv_attn (C = 320, 5 heads)
// Output: one line per (spec, tile): min / mean ms per launch over `reps` batches of back-to-back launches (~10 ms each, after
// >= 150 ms of warm-up launches) and the TFLOP/s of both.
// Environment: CBENCH_BATCH_MS (default 10) = length of one timed batch; CBENCH_POWER=1 samples the socket's hwmon (power1_average,
// freq1_input of card 0) every 50 ms during the timed batches and prints mean W / MHz beside the line (use batches >= 500 ms).
// Test tooling; numbers quoted from it are labelled "cbench" in profiles/.
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <dirent.h>
#include <unistd.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/star_hip.h"

#ifndef CBENCH_EMU
#include <hip/hip_runtime.h>
#define HCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#endif

struct Api {
  void* h = nullptr;
  template <class F> F sym(const char* n) {
    void* p = dlsym(h, n);
    if (!p) { fprintf(stderr, "missing symbol %s\n", n); exit(2); }
    return reinterpret_cast<F>(p);
  }
  decltype(&star_ctx_create) ctx_create;
  decltype(&star_last_error) last_error;
  decltype(&star_set_stream) set_stream;
  decltype(&star_sync) sync;
  decltype(&star_is_hostemu) is_hostemu;
  decltype(&star_gemm) gemm;
  decltype(&star_attn_fwd) attn_fwd;
  decltype(&star_temporal_qkv_attn) tq;
  decltype(&star_layer_norm_rowab) rowab;
};

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static float urand() {
  g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17;
  return (float)((g_rng >> 40) & 0xFFFFFF) / 16777216.0f;
}
static float nrand() { return (urand() + urand() + urand() + urand() - 2.0f) * 1.7320508f; }   // variance 1
static uint16_t to_f16(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t s = (x >> 16) & 0x8000u;
  int e = (int)((x >> 23) & 0xFF) - 127 + 15;
  uint32_t m = x & 0x7FFFFFu;
  if (e <= 0) return (uint16_t)s;                       // flush (operands here are O(1))
  if (e >= 31) return (uint16_t)(s | 0x7BFFu);
  uint32_t r = (m >> 13) + ((m >> 12) & 1u);
  uint32_t v = ((uint32_t)e << 10) + r;
  return (uint16_t)(s | v);
}
static uint16_t to_bf16(float f) { uint32_t x; memcpy(&x, &f, 4); return (uint16_t)((x + 0x7FFFu + ((x >> 16) & 1u)) >> 16); }

static bool g_bf16 = false;
static void* dev_alloc(size_t bytes) {
  void* p = nullptr;
#ifdef CBENCH_EMU
  if (posix_memalign(&p, 256, bytes ? bytes : 256)) exit(2);
#else
  HCHECK(hipMalloc(&p, bytes ? bytes : 256));
#endif
  return p;
}
static void dev_free(void* p) {
#ifdef CBENCH_EMU
  free(p);
#else
  HCHECK(hipFree(p));
#endif
}
static void upload(void* d, const void* h, size_t n) {
#ifdef CBENCH_EMU
  memcpy(d, h, n);
#else
  HCHECK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
#endif
}
// n 16-bit values ~ N(0, scale^2); a 4 M-element random block is tiled over larger buffers (operand statistics, not identity, matter)
static void* rand16(size_t n, float scale) {
  const size_t blk = n < ((size_t)1 << 22) ? n : ((size_t)1 << 22);
  std::vector<uint16_t> h(blk);
  for (size_t i = 0; i < blk; ++i) { const float v = nrand() * scale; h[i] = g_bf16 ? to_bf16(v) : to_f16(v); }
  char* d = (char*)dev_alloc(n * 2);
  for (size_t off = 0; off < n; off += blk) upload(d + off * 2, h.data(), ((n - off < blk) ? n - off : blk) * 2);
  return d;
}
static void download(void* h, const void* d, size_t n) {
#ifdef CBENCH_EMU
  memcpy(h, d, n);
#else
  HCHECK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
#endif
}
static float* rand32(size_t n, float scale, float shift = 0.f) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = nrand() * scale + shift;
  float* d = (float*)dev_alloc(n * 4);
  upload(d, h.data(), n * 4);
  return d;
}

// socket power / shader clock from sysfs hwmon (never rocm-smi beside a kernel on this pool)
struct PowerSampler {
  std::string dir;
  std::atomic<bool> stop{false};
  std::thread th;
  double sw = 0, sf = 0; int n = 0;
  static bool read_ll(const std::string& path, long long& v) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    const bool ok = fscanf(f, "%lld", &v) == 1;
    fclose(f);
    return ok;
  }
  // the hwmon directory of the card whose PCI address is HIP device 0's (a box shows every card of the node in sysfs, not only the visible one)
  PowerSampler() {
    char want[64] = "";
#ifndef CBENCH_EMU
    if (hipDeviceGetPCIBusId(want, sizeof want, 0) != hipSuccess) want[0] = 0;
    for (char* c = want; *c; ++c) *c = (char)tolower((unsigned char)*c);
#endif
    for (int card = 0; card < 128 && dir.empty(); ++card) {
      const std::string dev = "/sys/class/drm/card" + std::to_string(card) + "/device";
      char link[512];
      const ssize_t n = readlink(dev.c_str(), link, sizeof link - 1);
      if (n <= 0) continue;
      link[n] = 0;
      if (want[0] && !strstr(link, want)) continue;
      const std::string base = dev + "/hwmon";
      if (DIR* d = opendir(base.c_str())) {
        while (dirent* e = readdir(d)) {
          if (strncmp(e->d_name, "hwmon", 5)) continue;
          long long v;
          const std::string h = base + "/" + e->d_name;
          if (read_ll(h + "/power1_average", v) || read_ll(h + "/power1_input", v)) { dir = h; break; }
        }
        closedir(d);
      }
    }
    if (dir.empty()) fprintf(stderr, "cbench: no hwmon found for device %s\n", want);
  }
  void start() {
    if (dir.empty()) return;
    stop = false; sw = sf = 0; n = 0;
    th = std::thread([this] {
      while (!stop) {
        long long pw = 0, fq = 0;
        if (!read_ll(dir + "/power1_average", pw)) read_ll(dir + "/power1_input", pw);
        read_ll(dir + "/freq1_input", fq);
        sw += pw * 1e-6; sf += fq * 1e-6; ++n;
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
      }
    });
  }
  void finish(double& w, double& mhz) {
    w = mhz = 0;
    if (dir.empty()) return;
    stop = true; th.join();
    if (n) { w = sw / n; mhz = sf / n; }
  }
};

struct Timer {
#ifndef CBENCH_EMU
  hipStream_t stream;
  hipEvent_t e0, e1;
#endif
  Api* api; star_ctx* ctx;
  // warm-up: launches back to back for >= 150 ms (clocks and caches settle); measurement: `reps` batches of back-to-back launches
  // (a batch ~ 10 ms) bracketed by two events on the launch stream; min / mean are per launch
  template <class F> void run(const char* label, double flops, int reps, F&& launch) {
    if (int rc = launch()) { printf("%-44s FAILED rc=%d: %s\n", label, rc, api->last_error(ctx)); fflush(stdout); return; }
    api->sync(ctx);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
#ifdef CBENCH_EMU
    const double warm_ms = 0.0, batch_ms = 0.0;
#else
    const double warm_ms = 150.0, batch_ms = getenv("CBENCH_BATCH_MS") ? atof(getenv("CBENCH_BATCH_MS")) : 10.0;
    const bool power = getenv("CBENCH_POWER") != nullptr;
    PowerSampler ps;
#endif
    auto t0 = now();
    int nwarm = 0;
    do { for (int i = 0; i < 4; ++i) launch(); api->sync(ctx); nwarm += 4; } while (ms_since(t0) < warm_ms);
    const double est = ms_since(t0) / nwarm;
    int inner = est > 0 ? (int)(batch_ms / est + 0.5) : 1;
    if (inner < 1) inner = 1;
    double mn = 1e30, sum = 0;
#ifndef CBENCH_EMU
    if (power) ps.start();
#endif
    for (int i = 0; i < reps; ++i) {
#ifdef CBENCH_EMU
      auto t1 = now();
      for (int k = 0; k < inner; ++k) launch();
      api->sync(ctx);
      const double ms = ms_since(t1) / inner;
#else
      HCHECK(hipEventRecord(e0, stream));
      for (int k = 0; k < inner; ++k) launch();
      HCHECK(hipEventRecord(e1, stream));
      HCHECK(hipEventSynchronize(e1));
      float msf = 0; HCHECK(hipEventElapsedTime(&msf, e0, e1));
      const double ms = msf / inner;
#endif
      mn = ms < mn ? ms : mn; sum += ms;
    }
    const double mean = sum / reps;
    printf("%-44s min %8.4f ms  mean %8.4f ms  %8.1f TFLOP/s (mean)  %8.1f (min)  [%d x %d]", label, mn, mean, flops / (mean * 1e-3) * 1e-12,
           flops / (mn * 1e-3) * 1e-12, reps, inner);
#ifndef CBENCH_EMU
    if (power) { double w, mhz; ps.finish(w, mhz); printf("  %6.0f W %5.0f MHz", w, mhz); }
#endif
    printf("\n");
    fflush(stdout);
  }
};

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: cbench <lib.so> <f16|bf16> <spec file | -> [reps]\n"); return 2; }
  g_bf16 = !strcmp(argv[2], "bf16");
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  Api api;
  api.h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!api.h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  api.ctx_create = api.sym<decltype(&star_ctx_create)>("star_ctx_create");
  api.last_error = api.sym<decltype(&star_last_error)>("star_last_error");
  api.set_stream = api.sym<decltype(&star_set_stream)>("star_set_stream");
  api.sync = api.sym<decltype(&star_sync)>("star_sync");
  api.is_hostemu = api.sym<decltype(&star_is_hostemu)>("star_is_hostemu");
  api.gemm = api.sym<decltype(&star_gemm)>("star_gemm");
  api.attn_fwd = api.sym<decltype(&star_attn_fwd)>("star_attn_fwd");
  api.tq = api.sym<decltype(&star_temporal_qkv_attn)>("star_temporal_qkv_attn");
  api.rowab = api.sym<decltype(&star_layer_norm_rowab)>("star_layer_norm_rowab");
#ifdef CBENCH_EMU
  if (!api.is_hostemu()) { fprintf(stderr, "cbench_emu drives the emulator build only\n"); return 2; }
#else
  if (api.is_hostemu()) { fprintf(stderr, "this is the emulator build: use cbench_emu\n"); return 2; }
#endif
  star_ctx* ctx = nullptr;
  if (int rc = api.ctx_create(0, g_bf16 ? STAR_BF16 : STAR_F16, &ctx)) { fprintf(stderr, "star_ctx_create: %d\n", rc); return 2; }
  Timer T;
  T.api = &api; T.ctx = ctx;
#ifndef CBENCH_EMU
  HCHECK(hipStreamCreate(&T.stream));
  HCHECK(hipEventCreate(&T.e0)); HCHECK(hipEventCreate(&T.e1));
  if (api.set_stream(ctx, T.stream)) { fprintf(stderr, "star_set_stream: %s\n", api.last_error(ctx)); return 2; }
#endif
  FILE* f = strcmp(argv[3], "-") ? fopen(argv[3], "r") : stdin;
  if (!f) { fprintf(stderr, "cannot open %s\n", argv[3]); return 2; }
  char line[512];
  while (fgets(line, sizeof line, f)) {
    if (char* c = strchr(line, '#')) *c = 0;
    std::istringstream in(line);
    std::string kind;
    if (!(in >> kind)) continue;
    char label[256];
    if (kind == "gemm" || kind == "conv" || kind == "tconv") {
      // the last field may be a comma-separated list of tiles: one line per tile, same operands
      std::string rest((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
      while (!rest.empty() && isspace((unsigned char)rest.back())) rest.pop_back();
      const size_t sp = rest.find_last_of(" \t");
      std::string tiles_s = sp == std::string::npos ? rest : rest.substr(sp + 1);
      std::vector<int> tiles;
      { std::istringstream ts(tiles_s); std::string tk; while (std::getline(ts, tk, ',')) if (!tk.empty()) tiles.push_back(atoi(tk.c_str())); }
      if (tiles.empty()) { printf("bad spec: %s", line); continue; }
      std::istringstream in(rest);
      star_gemm_desc d{};
      long long M = 0, N = 0, K = 0; int epi = 0;
      size_t a_elems = 0;
      char stem[200];
      if (kind == "gemm") {
        in >> M >> N >> K >> epi;
        d.mode = STAR_A_PLAIN; d.lda = (int)K; a_elems = (size_t)M * K;
        snprintf(stem, sizeof stem, "gemm %lldx%lldx%lld epi=%d", M, N, K, epi);
      } else if (kind == "conv") {
        long long NB, H, W, Cin, Cout; in >> NB >> H >> W >> Cin >> Cout;
        M = NB * H * W; N = Cout; K = 9 * Cin; epi = STAR_EPI_BIAS;
        d.mode = STAR_A_CONV3X3; d.H = (int)H; d.Wd = (int)W; d.Cin = (int)Cin; d.Ho = (int)H; d.Wo = (int)W; d.stride = 1; d.pad_t = 1; d.pad_l = 1;
        d.lda = (int)Cin; a_elems = (size_t)M * Cin;
        snprintf(stem, sizeof stem, "conv3x3 %lldx%lldx%lld %lld->%lld", NB, H, W, Cin, Cout);
      } else {
        long long Fr, HW, C; in >> Fr >> HW >> C;
        M = Fr * HW; N = C; K = 3 * C; epi = STAR_EPI_BIAS | STAR_EPI_RES;
        d.mode = STAR_A_TCONV3; d.F = (int)Fr; d.HW = (int)HW; d.Cin = (int)C; d.lda = (int)C; a_elems = (size_t)M * C;
        snprintf(stem, sizeof stem, "tconv F=%lld HW=%lld C=%lld", Fr, HW, C);
      }
      if (M <= 0 || N <= 0 || K <= 0) { printf("bad spec: %s", line); continue; }
      const long long n_out = (epi & STAR_EPI_GEGLU) ? N / 2 : N;
      void* A = rand16(a_elems, 1.0f);
      void* W = rand16((size_t)N * K, 1.0f / sqrtf((float)K));
      void* C = dev_alloc((size_t)M * n_out * 2);
      void* R = (epi & STAR_EPI_RES) ? rand16((size_t)M * n_out, 1.0f) : nullptr;
      float* bias = rand32((size_t)N, 1.0f);
      float* colsum = rand32((size_t)N, 0.1f);
      float* rowab = (epi & STAR_EPI_ROWAFF) ? rand32((size_t)M * 2, 0.1f, 1.0f) : nullptr;
      d.A = A; d.W = W; d.C = C; d.res = R; d.bias = (epi & STAR_EPI_BIAS) ? bias : nullptr;
      d.M = (int)M; d.N = (int)N; d.K = (int)K; d.ldc = (int)n_out; d.ldr = (int)n_out; d.epi = epi;
      d.rowab = rowab; d.colsum = (epi & STAR_EPI_ROWAFF) ? colsum : nullptr;
      // every tile after the first is also compared bit for bit with the first one's output (hardware check of a variant against
      // the kernel it would replace)
      std::vector<uint16_t> first, cur;
      for (size_t ti = 0; ti < tiles.size(); ++ti) {
        const int tile = tiles[ti];
        d.force_tile = tile;
        snprintf(label, sizeof label, "%s tile=%d", stem, tile);
        T.run(label, 2.0 * M * N * K, reps, [&] { return api.gemm(ctx, &d); });
        if (tiles.size() > 1) {
          std::vector<uint16_t>& dst = ti == 0 ? first : cur;
          dst.resize((size_t)M * n_out);
          api.sync(ctx);
          download(dst.data(), C, dst.size() * 2);
          if (ti > 0) {
            size_t bad = 0;
            for (size_t i = 0; i < cur.size(); ++i) bad += cur[i] != first[i];
            printf("    tile %d vs tile %d: %zu of %zu outputs differ%s\n", tile, tiles[0], bad, cur.size(), bad ? "" : " (bit-identical)");
          }
#ifndef CBENCH_EMU
          HCHECK(hipMemset(C, 0xFF, (size_t)M * n_out * 2));     // the next tile must write every output itself
#else
          memset(C, 0xFF, (size_t)M * n_out * 2);
#endif
        }
      }
      dev_free(A); dev_free(W); dev_free(C); if (R) dev_free(R); dev_free(bias); dev_free(colsum); if (rowab) dev_free(rowab);
    } else if (kind == "attn") {
      long long B, heads, Nq, Nk; in >> B >> heads >> Nq >> Nk;
      std::vector<int> variants;
      { std::string vs; if (in >> vs) { std::istringstream ts(vs); std::string tk; while (std::getline(ts, tk, ',')) if (!tk.empty()) variants.push_back(atoi(tk.c_str())); } }
      if (variants.empty()) variants.push_back(9);
      const long long Cw = heads * 64;
      void* Q = rand16((size_t)B * Nq * Cw, 1.0f);
      void* Kp = rand16((size_t)B * Nk * Cw, 1.0f);
      void* V = rand16((size_t)B * Nk * Cw, 1.0f);
      void* O = dev_alloc((size_t)B * Nq * Cw * 2);
      star_attn_desc d{};
      d.Q = Q; d.K = Kp; d.V = V; d.O = O; d.ldq = d.ldk = d.ldv = d.ldo = (int)Cw;
      d.bsq = d.bso = Nq * Cw; d.bsk = d.bsv = Nk * Cw;
      d.Nq = (int)Nq; d.Nk = (int)Nk; d.heads = (int)heads; d.batch = (int)B; d.scale = 0.125f; d.variant = 9;
      std::vector<uint16_t> first, cur;
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        d.variant = variants[vi];
        snprintf(label, sizeof label, "attn B=%lld heads=%lld Nq=%lld Nk=%lld v=%d", B, heads, Nq, Nk, d.variant);
        T.run(label, 4.0 * B * heads * (double)Nq * Nk * 64.0, reps, [&] { return api.attn_fwd(ctx, &d); });
        if (variants.size() > 1) {
          std::vector<uint16_t>& dst = vi == 0 ? first : cur;
          dst.resize((size_t)B * Nq * Cw);
          api.sync(ctx);
          download(dst.data(), O, dst.size() * 2);
          if (vi > 0) {
            size_t bad = 0;
            for (size_t i = 0; i < cur.size(); ++i) bad += cur[i] != first[i];
            printf("    variant %d vs variant %d: %zu of %zu outputs differ%s\n", d.variant, variants[0], bad, cur.size(), bad ? "" : " (bit-identical)");
          }
#ifndef CBENCH_EMU
          HCHECK(hipMemset(O, 0xFF, dst.size() * 2));
#else
          memset(O, 0xFF, dst.size() * 2);
#endif
        }
      }
      dev_free(Q); dev_free(Kp); dev_free(V); dev_free(O);
    } else if (kind == "tq") {
      long long Fr, HW; in >> Fr >> HW;
      const long long M = Fr * HW;
      void* A = rand16((size_t)M * 320, 1.0f);
      void* W = rand16((size_t)960 * 320, 1.0f / sqrtf(320.f));
      void* O = dev_alloc((size_t)M * 320 * 2);
      float* bias = rand32(960, 0.1f);
      float* colsum = rand32(960, 0.1f);
      float* rowab = (float*)dev_alloc((size_t)M * 2 * 4);
      if (int rc = api.rowab(ctx, A, 320, rowab, (int)M, 320, 1e-5f, 0, nullptr, nullptr, 0, 0)) { printf("layer_norm_rowab failed rc=%d: %s\n", rc, api.last_error(ctx)); continue; }
      star_tq_desc d{};
      d.A = A; d.W = W; d.O = O; d.bias = bias; d.colsum = colsum; d.rowab = rowab;
      d.lda = 320; d.ldo = 320; d.HW = (int)HW; d.F = (int)Fr; d.C = 320; d.heads = 5; d.scale = 0.125f;
      snprintf(label, sizeof label, "tq F=%lld HW=%lld", Fr, HW);
      T.run(label, 2.0 * M * 960.0 * 320.0 + 4.0 * M * Fr * 320.0, reps, [&] { return api.tq(ctx, &d); });
      dev_free(A); dev_free(W); dev_free(O); dev_free(bias); dev_free(colsum); dev_free(rowab);
    } else {
      printf("unknown spec: %s", line);
    }
  }
  return 0;
}
