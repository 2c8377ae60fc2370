// ctx.h -- per-device context behind the C ABI (include/star_hip.h).
#pragma once
#include "rt.h"
#include <map>
#include <string>
#include <unordered_map>
#include <vector>
#include <memory>

namespace star {

enum DType : int { DT_F16 = 0, DT_BF16 = 1, DT_F32 = 2 };

// size-bucketed caching allocator: every kernel runs on ctx->stream, so a freed
// block may be handed out again immediately (stream order protects it).
// INVARIANT the hipGraph replay of the UNet forward relies on (unet.cpp: unet_forward_graph): a captured graph bakes in the
// addresses of blocks that are back on the free list once the capture ends; nothing pins them.  That is safe only while every
// Buf is scoped to one C-ABI call on the context's single host thread -- a replay then finds the pool exactly as the capture
// left it.  A Buf that outlives a call (a cached VAE / control buffer, a second thread on the context) could be handed one of
// those blocks and would be overwritten by a replay: the replay path therefore checks in_use() == 0 and runs the eager
// forward instead when anything is live; trim() bumps the generation, which invalidates the graphs.
class Pool {
 public:
  ~Pool() { release(); }
  void* alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    auto it = free_.lower_bound(bytes);
    if (it != free_.end() && it->first <= bytes + (bytes >> 2)) {
      void* p = it->second;
      free_.erase(it);
      live_[p] = bytes_of_[p];
      in_use_ += bytes_of_[p];
      if (in_use_ > peak_) peak_ = in_use_;
      return p;
    }
    void* p = nullptr;
    if (rt::dev_malloc(&p, bytes)) {
      // try again after dropping the cache
      trim();
      if (rt::dev_malloc(&p, bytes)) return nullptr;
    }
    bytes_of_[p] = bytes;
    live_[p] = bytes;
    total_ += bytes;
    in_use_ += bytes;
    if (in_use_ > peak_) peak_ = in_use_;
    return p;
  }
  void free(void* p) {
    if (!p) return;
    auto it = live_.find(p);
    if (it == live_.end()) return;
    in_use_ -= it->second;
    free_.emplace(it->second, p);
    live_.erase(it);
  }
  void trim() {
    ++gen_;   // cached blocks go back to the driver: captured graphs that hold their addresses are stale
    for (auto& kv : free_) { rt::dev_free(kv.second); total_ -= kv.first; bytes_of_.erase(kv.second); }
    free_.clear();
  }
  void release() {
    trim();
    for (auto& kv : live_) { rt::dev_free(kv.first); }
    live_.clear();
    bytes_of_.clear();
    total_ = in_use_ = 0;
  }
  size_t total() const { return total_; }
  size_t peak() const { return peak_; }
  size_t in_use() const { return in_use_; }
  uint64_t generation() const { return gen_; }
 private:
  uint64_t gen_ = 0;
  std::multimap<size_t, void*> free_;
  std::unordered_map<void*, size_t> live_, bytes_of_;
  size_t total_ = 0, in_use_ = 0, peak_ = 0;
};

struct Ctx;

// RAII device buffer from the ctx pool
struct Buf {
  Ctx* ctx = nullptr;
  void* p = nullptr;
  size_t bytes = 0;
  Buf() = default;
  Buf(Ctx* c, size_t n);
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  Buf(Buf&& o) noexcept : ctx(o.ctx), p(o.p), bytes(o.bytes) { o.p = nullptr; }
  Buf& operator=(Buf&& o) noexcept;
  ~Buf();
  void reset();
  template <class U> U* as() const { return reinterpret_cast<U*>(p); }
};

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct UNetModel;
struct VaeModel;

// per-kernel-family timing with HIP events on the launch stream (bench.py roofline leg)
enum ProfKind : int { PK_ATTN_SELF = 0, PK_ATTN_CROSS, PK_TATTN, PK_GEMM, PK_CONV, PK_TCONV, PK_GN, PK_LN, PK_MISC, PK_COUNT };
struct ProfRec { int kind; double flops; double bytes; void* e0; void* e1; int d0, d1, d2, d3; };

struct Ctx {
  int device = 0;
  int num_cus = 256;          // compute units of the device (the GEMM launcher balances its last round of tiles against it)
  long long ln_fused = 0;     // LayerNorm row-coefficient / map passes served from their producer's row statistics (diagnostic)
  long long gn_fused = 0;     // GroupNorms finalized from their producer's partial statistics instead of a statistics pass (diagnostic)
  long long gemm_splits = 0;  // launches the GEMM launcher split into full rounds of big tiles + a remainder of small ones (diagnostic)
  int dtype = DT_F16;
  hipStream_t stream = nullptr;
  std::string err;
  Pool pool;
  void* zero_page = nullptr;  // 256 B of zeros (conv padding source)
  std::unordered_map<std::string, HostTensor> host_tensors;  // staged by star_load_tensor
  std::shared_ptr<UNetModel> unet;   // shared_ptr: deleter bound where the type is complete
  std::shared_ptr<VaeModel> vae;
  bool profiling = false;
  unsigned prof_mask = ~0u;   // kernel families (bit = ProfKind) that get HIP events while profiling: two event records per launch cost
                              // ~2-4 us of stream time each -- 1.4 s of a 61 s clip with all ~340 000 launches bracketed (bench.py times only
                              // the dominant kernel inside its timed region)
  std::vector<ProfRec> prof;
  bool unet_graph = false;   // star_unet_graph(): replay the UNet forward from a captured hipGraph (unet.cpp)
  int fail(const std::string& m) { err = m; return 1; }
  size_t esize() const { return dtype == DT_F32 ? 4 : 2; }
};

inline Buf::Buf(Ctx* c, size_t n) : ctx(c), bytes(n) { p = c->pool.alloc(n); }
inline Buf::~Buf() { reset(); }
inline void Buf::reset() { if (p && ctx) ctx->pool.free(p); p = nullptr; }
inline Buf& Buf::operator=(Buf&& o) noexcept { if (this != &o) { reset(); ctx = o.ctx; p = o.p; bytes = o.bytes; o.p = nullptr; } return *this; }

}  // namespace star
