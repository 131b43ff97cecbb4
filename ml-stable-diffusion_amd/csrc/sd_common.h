// Common host/device definitions for libsdmi355 (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <atomic>
#include <vector>

namespace sd {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// Output stores of the step's kernels.  -DSD_SC1_STORES (experiment, round 6): write-through (sc1) instead of plain stores, so that
// a kernel's results leave the XCD's L2 while it runs instead of at its end-of-kernel release (MI355X_MICROARCH.md "boundary":
// + bytes / 6 TB/s when the predecessor leaves dirty lines).  16- and 8-byte forms; the s_nop keeps the data registers alive until
// the store has read them (cdna guide 5.7).
template <typename V>
__device__ __forceinline__ void out_store(V* p, const V& v) {
#if defined(SD_SC1_STORES) && defined(__HIP_DEVICE_COMPILE__)
  if constexpr (sizeof(V) == 16) {
    typedef unsigned u4_ __attribute__((ext_vector_type(4)));
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(u4_, v)) : "memory");
  } else if constexpr (sizeof(V) == 8) {
    typedef unsigned u2_ __attribute__((ext_vector_type(2)));
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(__builtin_bit_cast(u2_, v)) : "memory");
  } else {
    *p = v;
  }
#else
  *p = v;
#endif
}

// ---- errors: C++ exceptions inside, int status + thread-local string at the C ABI ------------
enum Status : int {
  kOk = 0,
  kInvalidArgument = -1,   // Python wrapper raises ValueError / TypeError
  kNotFound = -2,          // FileNotFoundError / KeyError (missing weight)
  kHipError = -3,          // RuntimeError
  kUnsupported = -4,       // NotImplementedError
  kInternal = -5,
};

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] void fail(int code, const char* fmt, ...);

#define SD_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      ::sd::fail(::sd::kHipError, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                 __FILE__, __LINE__);                                                        \
  } while (0)

#define SD_REQUIRE(cond, code, ...)            \
  do {                                         \
    if (!(cond)) ::sd::fail((code), __VA_ARGS__); \
  } while (0)

// ---- device memory ---------------------------------------------------------------------------
// One arena per model handle: a few large hipMallocs carved by a bump pointer (256-B aligned).
// 288 GB of HBM per GPU means we never recycle activation buffers inside a forward: every
// tensor of the static graph owns its bytes, which is what makes HIP-graph replay trivially safe.
class Arena {
 public:
  Arena() = default;
  ~Arena();
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;
  void* alloc(size_t bytes);          // 256-B aligned, zero-initialised
  template <typename T>
  T* alloc_n(size_t n) { return reinterpret_cast<T*>(alloc(n * sizeof(T))); }
  size_t bytes() const { return total_; }
  const std::vector<void*>& chunks() const { return chunks_; }       // (debug scans: SD_NAN_TRACE)
  const std::vector<size_t>& chunk_bytes() const { return sizes_; }

 private:
  std::vector<void*> chunks_;
  std::vector<size_t> sizes_;
  size_t cap_ = 0, cur_ = 0, total_ = 0;
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
};

// A/B switches of the measurement tools: read from the environment ONLY when SD_TUNE is set, so a production process cannot be
// steered by a stray variable.  (cached per call site by the callers' function-local statics)
inline int tune_env_int(const char* name, int dflt) {
  static const bool on = getenv("SD_TUNE") != nullptr;
  if (!on) {   // (callers cache the result in function-local statics: one line per switch and call site at most)
    if (getenv(name)) fprintf(stderr, "libsdmi355: %s is set but ignored: A/B switches are only read when SD_TUNE=1 is set too\n", name);
    return dflt;
  }
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
inline bool tune_env_set(const char* name) {
  static const bool on = getenv("SD_TUNE") != nullptr;
  if (!on && getenv(name)) fprintf(stderr, "libsdmi355: %s is set but ignored: A/B switches are only read when SD_TUNE=1 is set too\n", name);
  return on && getenv(name) != nullptr;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }


// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per (kernel, device): remember it per device so a
// process that drives several GPUs (one handle per device) raises the limit on each of them.
// Handles may be driven from different host threads: the flags are atomics; two threads racing on the
// first launch both set the (idempotent) attribute.
struct DynLdsOnce {
  std::atomic<bool> done[32] = {};
  template <class K>
  void set(K kernel, size_t bytes) {
    int dev = 0;
    SD_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 32 || !done[dev].load(std::memory_order_acquire)) {
      SD_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      if (dev >= 0 && dev < 32) done[dev].store(true, std::memory_order_release);
    }
  }
};

}  // namespace sd
