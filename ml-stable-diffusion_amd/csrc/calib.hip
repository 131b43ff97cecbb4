// Box calibration for bench.py: boxes of the GPU pool run ONE binary of this library 20 % apart at identical
// reported clocks, so a single ms-per-step figure says nothing about a code change unless it comes with what the box itself
// can do.  Four fixed micro-measurements, none of which touches the UNet code:
//   copy_gbs        float4 copy of 1 GiB (read + written bytes / time): the HBM stream
//   mfma_tflops     dense v_mfma_f32_32x32x16_f16 register loop, every SIMD busy: the matrix pipe at the box's power state
//   empty_launch_us a captured graph of 323 empty 256-workgroup launches (the step's launch count): the launch floor
//   chain_us        the same graph shape, each launch reading 8 MB it has never touched and writing 0.5 MB the next one
//                   reads back: a dependent chain of short kernels on cold operands - what the batch-2 step actually is
// Round 5 found the four figures above IDENTICAL (to 1 %) on boxes that run the step 4.40 and 5.52 ms apart, so two more:
//   handover_us     323 launches, each reading ALL 8 MB the previous one wrote - every workgroup the 16 KB of a workgroup that
//                   ran on ANOTHER XCD - and writing 8 MB: the producer -> consumer hand-over of activations between the L2s
//   latency_ns      one wave chasing 512 dependent 64-byte-strided pointers through a 1-GiB table (never-touched lines: HBM
//                   latency) and latency_l2_ns the same through a 2-MB table after a warm-up pass (L2 / Infinity-Cache latency)
#include <utility>
#include <vector>

#include "kernels.h"

namespace sd {

namespace {

__global__ __launch_bounds__(256) void calib_copy_kernel(const floatx4* __restrict__ src, floatx4* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void calib_mfma_kernel(float* out, int iters) {
  half8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = (half_t)(0.001f * (float)((threadIdx.x + e) & 7));
    b[e] = (half_t)(0.002f * (float)((threadIdx.x * 3 + e) & 7));
  }
  floatx16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;   // keeps the loop alive, never true
}

__global__ __launch_bounds__(256) void calib_empty_kernel(float* out) {
  if (out == nullptr) return;   // (always taken apart from the first thread's never-true test below)
  if (threadIdx.x == 1023) out[0] = 0.f;
}

// 512 workgroups: each thread reads 4 x 16 B of `cold` it has not seen (8 MB per launch at a rotating offset) plus 16 B of what
// the previous launch wrote, and writes 16 B for the next one.
__global__ __launch_bounds__(256) void calib_chain_kernel(const floatx4* __restrict__ cold, size_t off4, const floatx4* __restrict__ prev,
                                                           floatx4* __restrict__ next) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  floatx4 s = prev[t];
#pragma unroll
  for (int k = 0; k < 4; ++k) s += cold[off4 + (size_t)k * 512 * 256 + t];
  next[t] = s;
}

// 512 workgroups x 256 threads x 4 x 16 B = 8 MB in, 8 MB out.  Workgroup i reads what workgroup (37 i + 11) mod 512 wrote:
// XCDs are assigned round-robin (id mod 8) and 37 i + 11 = 5 i + 3 (mod 8) != i for every i, so every line comes from another L2.
__global__ __launch_bounds__(256) void calib_handover_kernel(const floatx4* __restrict__ prev, floatx4* __restrict__ next) {
  const unsigned src = (blockIdx.x * 37u + 11u) & 511u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const floatx4 v = prev[((size_t)src * 4 + k) * 256 + threadIdx.x];
    next[((size_t)blockIdx.x * 4 + k) * 256 + threadIdx.x] = v + floatx4{1.f, 1.f, 1.f, 1.f};
  }
}

// A SMALL grid with a reduction in the middle (the shape of the single-launch GroupNorm, of the 16x16-level attention, of conv_in /
// conv_out: 64 workgroups of 256 threads): every thread reads 16 x 16 B the previous launch wrote, the workgroup reduces through
// LDS behind one barrier, every thread writes 16 x 16 B.  Session E of round 5 (profiles/r05_ffn_proj_ab_slow_box.log) met a box that
// runs the step 20 % slower with all seven figures above unchanged: on it every launch of MANY workgroups took its usual time and
// the launches of 64-160 workgroups 1.3-2.0 x as long.
__global__ __launch_bounds__(256) void calib_small_grid_kernel(const floatx4* __restrict__ prev, floatx4* __restrict__ next) {
  __shared__ float red[4];
  const size_t base = (size_t)blockIdx.x * 16 * 256 + threadIdx.x;
  floatx4 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = prev[base + (size_t)k * 256];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float t = ((red[0] + red[1]) + (red[2] + red[3])) * 1.0e-9f;
#pragma unroll
  for (int k = 0; k < 16; ++k) next[base + (size_t)k * 256] = v[k] * 0.5f + floatx4{t, t, t, t};
}

// COLD CODE (Finding 14, tools/ubench/icache.hip): 32 instantiations of one kernel with ~30 KB of straight-line code each.  A chain
// that repeats ONE of them keeps its code in the instruction caches; a chain that walks all 32 round-robin fetches 30 KB of cold
// code per launch - the UNet step's situation (~40 different kernels one after the other).  The difference per launch is what a
// box's instruction-fetch path costs: 0.8 us on the fast boxes of the pool, 11 us on the slow ones - the ONE figure of this file
// that separates them.
template <int ID>
__global__ __launch_bounds__(256) void calib_code_kernel(float* buf) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float a = buf[t], b = a + 1.f, c = a + 2.f, d = a + 3.f;
#pragma unroll
  for (int i = 0; i < 448; ++i) {   // 4 independent chains x 448 FMAs with 32-bit literals
    a = __builtin_fmaf(a, 1.0f + 1e-6f * (float)(ID * 4096 + i * 4 + 0), 1e-7f * (float)(i + ID));
    b = __builtin_fmaf(b, 1.0f + 1e-6f * (float)(ID * 4096 + i * 4 + 1), 2e-7f * (float)(i + ID));
    c = __builtin_fmaf(c, 1.0f + 1e-6f * (float)(ID * 4096 + i * 4 + 2), 3e-7f * (float)(i + ID));
    d = __builtin_fmaf(d, 1.0f + 1e-6f * (float)(ID * 4096 + i * 4 + 3), 4e-7f * (float)(i + ID));
  }
  buf[t] = (a + b) + (c + d);
}
typedef void (*calib_code_fn)(float*);
template <int... I>
std::vector<calib_code_fn> calib_code_table(std::integer_sequence<int, I...>) { return {calib_code_kernel<I>...}; }

// the table holds, at element i * stride, the index of the next element; one lane walks it
__global__ void calib_chase_init_kernel(unsigned* tab, unsigned n, unsigned stride) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tab[(size_t)i * stride] = ((i * 40503u + 12345u) % n) * stride;   // (a fixed pseudo-random successor)
}
__global__ void calib_chase_kernel(const unsigned* __restrict__ tab, int hops, unsigned start, unsigned* out, long long* cycles) {
  if (threadIdx.x != 0) return;
  unsigned p = start;
  const long long t0 = wall_clock64();
  for (int h = 0; h < hops; ++h) p = __builtin_nontemporal_load(tab + p);
  const long long t1 = wall_clock64();
  out[0] = p;
  cycles[0] = t1 - t0;   // ticks of the 100-MHz constant clock
}

float time_graph(hipStream_t st, hipGraphExec_t g, int reps) {
  hipEvent_t e0, e1;
  SD_HIP(hipEventCreate(&e0));
  SD_HIP(hipEventCreate(&e1));
  SD_HIP(hipGraphLaunch(g, st));
  SD_HIP(hipStreamSynchronize(st));
  SD_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) SD_HIP(hipGraphLaunch(g, st));
  SD_HIP(hipEventRecord(e1, st));
  SD_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  SD_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ms / (float)reps;
}

}  // namespace

// out[0..8] = copy_gbs, mfma_tflops, empty_launch_us, chain_us, handover_us, latency_ns (HBM), latency_l2_ns, small_grid_us, cold_code_us
void run_calibration(int device, float* out) {
  // everything this function creates is released on every exit path (a throwing SD_HIP included), and the caller's current
  // device is put back
  struct Guard {
    int prev = -1;
    hipStream_t st = nullptr;
    floatx4 *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Guard() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (src) (void)hipFree(src);
      if (dst) (void)hipFree(dst);
      if (st) (void)hipStreamDestroy(st);
      if (prev >= 0) (void)hipSetDevice(prev);
    }
  } g;
  SD_HIP(hipGetDevice(&g.prev));
  SD_HIP(hipSetDevice(device));
  SD_HIP(hipStreamCreateWithFlags(&g.st, hipStreamNonBlocking));
  const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
  SD_HIP(hipMalloc(reinterpret_cast<void**>(&g.src), bytes));
  SD_HIP(hipMalloc(reinterpret_cast<void**>(&g.dst), bytes));
  SD_HIP(hipEventCreate(&g.e0));
  SD_HIP(hipEventCreate(&g.e1));
  const hipStream_t st = g.st;
  floatx4 *const src = g.src, *const dst = g.dst;
  const hipEvent_t e0 = g.e0, e1 = g.e1;
  SD_HIP(hipMemsetAsync(src, 1, bytes, st));
  SD_HIP(hipMemsetAsync(dst, 0, bytes, st));
  auto timed = [&](int reps, auto&& launch) {
    launch();
    SD_HIP(hipStreamSynchronize(st));
    SD_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch();
    SD_HIP(hipEventRecord(e1, st));
    SD_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SD_HIP(hipEventElapsedTime(&ms, e0, e1));
    return ms / (float)reps;
  };
  // (a) copy
  {
    const float ms = timed(6, [&] { hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 16), dim3(256), 0, st, src, dst, n4); });
    out[0] = (float)(2.0 * (double)bytes / (ms * 1e-3) / 1e9);
  }
  // (b) dense MFMA loop: 2048 workgroups x 4 waves x iters x 4 MFMAs x 32768 FLOP
  {
    const int iters = 1024, wgs = 2048;
    float* sink = reinterpret_cast<float*>(dst);
    const float ms = timed(4, [&] { hipLaunchKernelGGL(calib_mfma_kernel, dim3(wgs), dim3(256), 0, st, sink, iters); });
    out[1] = (float)((double)wgs * 4 * iters * 4 * 32768.0 / (ms * 1e-3) / 1e12);
  }
  // (c) / (d): captured graphs of 323 launches
  const int kLaunches = 323;
  auto capture = [&](auto&& body) {
    hipGraph_t g;
    hipGraphExec_t ge;
    SD_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    body();
    SD_HIP(hipStreamEndCapture(st, &g));
    SD_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    return ge;
  };
  {
    float* sink = reinterpret_cast<float*>(dst);
    hipGraphExec_t ge = capture([&] {
      for (int i = 0; i < kLaunches; ++i) hipLaunchKernelGGL(calib_empty_kernel, dim3(256), dim3(256), 0, st, sink);
    });
    out[2] = time_graph(st, ge, 10) * 1e3f / (float)kLaunches;
    (void)hipGraphExecDestroy(ge);
  }
  {
    // ping-pong hand-over buffers at the start of dst; the cold reads walk src in 8-MB steps with a 3-MB skew (never the same lines
    // twice inside one replay: 323 x 8 MB > 1 GiB wraps once, onto lines evicted 1 GiB of traffic earlier)
    floatx4* pp0 = dst;
    floatx4* pp1 = dst + (size_t)512 * 256;
    const size_t span4 = (size_t)4 * 512 * 256;   // float4s read per launch
    hipGraphExec_t ge = capture([&] {
      size_t off = 0;
      for (int i = 0; i < kLaunches; ++i) {
        hipLaunchKernelGGL(calib_chain_kernel, dim3(512), dim3(256), 0, st, src, off, (i & 1) ? pp1 : pp0, (i & 1) ? pp0 : pp1);
        off += span4 + 196608;
        if (off + span4 > n4) off = (off + span4) % (n4 - span4);
      }
    });
    out[3] = time_graph(st, ge, 10) * 1e3f / (float)kLaunches;
    (void)hipGraphExecDestroy(ge);
  }
  {   // (e) hand-over chain: ping-pong between two 8-MB buffers at the start of src / dst
    floatx4* b0 = src;
    floatx4* b1 = dst;
    hipGraphExec_t ge = capture([&] {
      for (int i = 0; i < kLaunches; ++i)
        hipLaunchKernelGGL(calib_handover_kernel, dim3(512), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1);
    });
    out[4] = time_graph(st, ge, 10) * 1e3f / (float)kLaunches;
    (void)hipGraphExecDestroy(ge);
  }
  {   // (g) small-grid chain: 64 workgroups, 4 MB in / 4 MB out per launch, ping-pong between the starts of src and dst
    floatx4* b0 = src;
    floatx4* b1 = dst;
    hipGraphExec_t ge = capture([&] {
      for (int i = 0; i < kLaunches; ++i)
        hipLaunchKernelGGL(calib_small_grid_kernel, dim3(64), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1);
    });
    out[7] = time_graph(st, ge, 10) * 1e3f / (float)kLaunches;
    (void)hipGraphExecDestroy(ge);
  }
  {   // (h) cold code: 320-launch chains of 64-workgroup launches, 32 kernels round-robin minus one kernel repeated
    const std::vector<calib_code_fn> ks = calib_code_table(std::make_integer_sequence<int, 32>{});
    float* buf = reinterpret_cast<float*>(dst);
    float us[2];
    for (int mode = 0; mode < 2; ++mode) {
      hipGraphExec_t ge = capture([&] {
        for (int i = 0; i < 320; ++i) hipLaunchKernelGGL(ks[mode == 0 ? 0 : i % 32], dim3(64), dim3(256), 0, st, buf);
      });
      const float a = time_graph(st, ge, 3), b = time_graph(st, ge, 3);
      us[mode] = (a < b ? a : b) * 1e3f / 320.f;
      (void)hipGraphExecDestroy(ge);
    }
    out[8] = us[1] - us[0];
  }
  {   // (f) dependent-load latency: never-touched lines of the 1-GiB table, then a 2-MB table that was just walked
    unsigned* tab = reinterpret_cast<unsigned*>(src);
    unsigned* sink = reinterpret_cast<unsigned*>(dst);
    long long* cyc = reinterpret_cast<long long*>(dst) + 8;
    auto chase = [&](unsigned n, unsigned stride, int hops, int passes) {
      hipLaunchKernelGGL(calib_chase_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tab, n, stride);
      if (passes == 1) {   // big table: evict what the initialisation left in the caches
        SD_HIP(hipMemsetAsync(dst + (size_t)4096, 0, bytes - 65536, st));
      }
      long long ticks = 0;
      for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(calib_chase_kernel, dim3(1), dim3(64), 0, st, tab, hops, 0u, sink, cyc);
        SD_HIP(hipStreamSynchronize(st));
        SD_HIP(hipMemcpy(&ticks, cyc, sizeof(ticks), hipMemcpyDeviceToHost));
      }
      return (float)ticks * 10.0f / (float)hops;   // ns per hop (last pass)
    };
    out[5] = chase((unsigned)(bytes / 4096), 1024u, 512, 1);          // 262 144 entries, 4 KB apart: 512 hops on cold lines
    out[6] = chase(32768u, 16u, 4096, 2);                              // 2 MB: the second pass walks what the first pulled in
  }
  SD_HIP(hipStreamSynchronize(st));
}

}  // namespace sd
