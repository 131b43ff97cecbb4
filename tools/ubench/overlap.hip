// can launch k+1's prologue (code fetch, launch, its WHOLE weight slice) run under launch k?
//
// The chain is the 8x8 level of the UNet step at CFG batch 2 (unet.py:731-795 mid block): GroupNorm (G: 64 workgroups, 328 KB) ->
// 3x3 conv 1280 -> 1280 as a weight stream (W: 200 workgroups of 8 waves, 29.5 MB of cold fp16 weights straight into VGPRs in
// MFMA fragment order, 72 MFMAs per wave, K slices summed through LDS, five fp32 slabs) -> slab combine (R: 160 workgroups),
// 100 times = 300 dependent launches; every W reads a different 29.5-MB slice of a 1.5-GB pool (cold like in the step).
//
//   serial   : one stream, captured graph - what the library does today
//   serial+hs: the same with the signal / poll code active (the price of the instrumentation alone)
//   S streams: launch k goes to stream k % S, so the graph has S chains and NO edge between neighbours (k -> k+S only); kernel k
//              adds 1 to cnt[k] per workgroup after its last store (release), kernel k+1 does everything that does not depend on
//              k FIRST - W: all 18 weight fragments of every wave requested - then ONE lane polls cnt[k] (relaxed, s_sleep,
//              bounded), one agent-scope acquire, barrier, activation loads.
//     prefetch=0: the poll comes first (isolates hand-off cost against the kernel boundary it replaces)
//     sc1=1     : outputs stored write-through (sc1) + vmcnt(0) instead of plain stores + release fence
//   2 graphs  : ROCm 7.2 runs the branches of ONE captured graph one after the other (first run of this probe: every second
//               launch of the "S streams" rows above hit its 2-ms spin bound - chain 0 ran to its end before chain 1 started;
//               the `branches` line below measures that directly), so the same S chains are also captured as S separate graphs
//               and launched on S streams.
//   flag kernels: no fence anywhere.  Stream B carries ONLY the weight-streaming convs, stream A everything else; a one-thread
//               kernel behind the producer sets a flag (it starts after the producer's end-of-kernel release), the consumer -
//               launched earlier on the other stream, weights already requested - polls it relaxed and reads with plain loads.
// Every mode must reproduce the serial chain's final tensor BIT FOR BIT (the buffers are re-used every iteration, so a consumer
// that skips its acquire reads the previous iteration's lines from its L1 / L2).
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/overlap.hip -o /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int M = 128, C = 1280, N = 1280, TAPS = 9, NSL = C / 32, NW = 8, SPLITS = NSL / NW, NF = TAPS * 2;
constexpr int G_WGS = 64, W_WGS = (N / 32) * SPLITS, R_WGS = M * N / 4 / 256;
constexpr size_t W_HALVES = (size_t)N * C * TAPS;   // 14.7 M halves = 29.5 MB

struct Dep {
  unsigned* wait;     // counter of the producer (null: stream order is the dependency)
  unsigned expect;    // its workgroup count
  int nofence;        // 1: poll only (flag-kernel protocol), no acquire
  unsigned* sig;      // this launch's counter (null: none)
  unsigned* err;      // set when a spin gave up
};

__device__ __forceinline__ void dep_wait(const Dep& d) {
  if (d.wait) {   // kernel-uniform
    if (threadIdx.x == 0) {
      const long long t0 = wall_clock64();   // 100 MHz
      while (__hip_atomic_load((gu32*)d.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < d.expect) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 200000) {   // 2 ms: give up loudly, never hang the box
          __hip_atomic_store((gu32*)d.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      if (!d.nofence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

// fence-free consumer side of the flag-kernel protocol: the producer's data reached memory through ITS end-of-kernel release
// (the flag kernel runs behind it in stream order); this CU cannot hold stale lines of it (nobody read them since the
// producer's kernel-start invalidate).  d.expect < 0 selects this form in dep_wait.
__global__ void k_flag(unsigned* f) {
  if (threadIdx.x == 0) __hip_atomic_store((gu32*)f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool SC1>
__device__ __forceinline__ void dep_signal(const Dep& d) {
  if (d.sig) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave
    __syncthreads();
    if (threadIdx.x == 0) {
      if (!SC1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the compiler may drop the wait behind buffer_wbl2 (guide, G16 pitfall 12)
      }
      __hip_atomic_fetch_add((gu32*)d.sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <bool SC1, typename V>
__device__ __forceinline__ void store16(V* p, V v) {
  if constexpr (SC1) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 16, 0x00020000);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rs, 0, 0, 16);   // aux 16 = sc1: write-through
  } else {
    *p = v;
  }
}
template <bool SC1>
__device__ __forceinline__ void store8(half4* p, half4 v) {
  if constexpr (SC1) {
    __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// G: GroupNorm + SiLU of x [2][64][1280] (32 groups of 40 channels per sample) -> y.  grid 64, 256 threads, 320 half8 items.
template <bool SC1>
__global__ __launch_bounds__(256) void k_gn(const half_t* __restrict__ x, half_t* __restrict__ y, const float* __restrict__ gamma, Dep d) {
  __shared__ float red[4][2];
  const int t = threadIdx.x, b = blockIdx.x >> 5, g = blockIdx.x & 31;
  floatx4 ga[2], gb[2];   // does not depend on the producer
  int px[2], ch[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = t + 256 * i;
    px[i] = id < 320 ? id / 5 : -1;
    ch[i] = g * 40 + (id % 5) * 8;
    ga[i] = *reinterpret_cast<const floatx4*>(gamma + ch[i]);
    gb[i] = *reinterpret_cast<const floatx4*>(gamma + ch[i] + 4);
  }
  dep_wait(d);
  half8 v[2];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    v[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
    if (px[i] >= 0) v[i] = *reinterpret_cast<const half8*>(x + ((size_t)b * 64 + px[i]) * C + ch[i]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[i][e];
      s += f;
      q = fmaf(f, f, q);
    }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((t & 63) == 0) {
    red[t >> 6][0] = s;
    red[t >> 6][1] = q;
  }
  __syncthreads();
  s = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
  q = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
  const float mean = s * (1.f / 2560.f), rstd = rsqrtf(fmaxf(q * (1.f / 2560.f) - mean * mean, 0.f) + 1e-5f);
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (px[i] >= 0) {
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gm = e < 4 ? ga[i][e] : gb[i][e - 4];
        float z = ((float)v[i][e] - mean) * rstd * gm;
        z = z * __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        o[e] = (half_t)z;
      }
      store16<SC1>(reinterpret_cast<half8*>(y + ((size_t)b * 64 + px[i]) * C + ch[i]), o);
    }
  dep_signal<SC1>(d);
}

// W: slabs[split][m][n] = sum over the split's 8 channel slices of x[m][slice] . wt[n][slice][taps]  (the tap shift of the real
// conv is left out: the same activation fragment meets all nine taps - the memory side and the MFMA count are the product's).
// grid (40 strips, 5 splits), 512 threads, 8 x 16 KB of LDS for the K-slice sum.
template <bool SC1, bool PREFETCH>
__global__ __launch_bounds__(512) void k_ws(const half_t* __restrict__ x, const half_t* __restrict__ wt, float* __restrict__ partial, Dep d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = blockIdx.x, split = blockIdx.y;
  const int slice = split * NW + wave;
  char* const region = smem + wave * 16384;
  floatx16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const half_t* wp = wt + (((size_t)strip * NSL + slice) * NF * 64 + lane) * 8;
  half8 bf[NF];
  if (PREFETCH) {
#pragma unroll
    for (int j = 0; j < NF; ++j) bf[j] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(wp + (size_t)j * 512));
  }
  dep_wait(d);
  half8 xf[8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xf[h * 4 + i] = *reinterpret_cast<const half8*>(x + (size_t)(i * 32 + (lane & 31)) * C + slice * 32 + h * 16 + (lane >> 5) * 8);
  if (!PREFETCH) {
#pragma unroll
    for (int j = 0; j < NF; ++j) bf[j] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(wp + (size_t)j * 512));
  }
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[tap * 2 + h], xf[h * 4 + i], acc[i], 0, 0, 0);
  {
    floatx4* rg = reinterpret_cast<floatx4*>(region);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) rg[(i * 4 + q) * 64 + lane] = floatx4{acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = it * 512 + tid;
    const int ml = e >> 3, pc = e & 7;
    const int id = (((ml >> 5) * 4 + (pc >> 1)) * 64) + (pc & 1) * 32 + (ml & 31);
    floatx4 s = reinterpret_cast<const floatx4*>(smem)[id];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += reinterpret_cast<const floatx4*>(smem + w * 16384)[id];
    store16<SC1>(reinterpret_cast<floatx4*>(partial + ((size_t)split * M + ml) * N + strip * 32 + 4 * pc), s);
  }
  dep_signal<SC1>(d);
}

// R: out = fp16(sum of the five slabs + bias + res).  grid 160, 256 threads, one float4 item per thread.
template <bool SC1>
__global__ __launch_bounds__(256) void k_red(const float* __restrict__ partial, const float* __restrict__ bias, const half_t* __restrict__ res,
                                             half_t* __restrict__ out, Dep d) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  const int m = id / (N / 4), n = (id - m * (N / 4)) * 4;
  const floatx4 bb = *reinterpret_cast<const floatx4*>(bias + n);
  dep_wait(d);
  floatx4 v[SPLITS];
#pragma unroll
  for (int z = 0; z < SPLITS; ++z) v[z] = *reinterpret_cast<const floatx4*>(partial + ((size_t)z * M + m) * N + n);
  const half4 rr = *reinterpret_cast<const half4*>(res + (size_t)m * N + n);
  floatx4 s = bb;
#pragma unroll
  for (int z = 0; z < SPLITS; ++z) s += v[z];
  half4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (half_t)(s[e] + (float)rr[e]);
  store8<SC1>(reinterpret_cast<half4*>(out + (size_t)m * N + n), o);
  dep_signal<SC1>(d);
}

struct Bufs {
  half_t *xo, *xa, *pool;
  float *slab, *gamma, *bias;
  unsigned *cnt, *err;
  size_t pool_slices;
};

// how the 3 * iters launches are laid out
enum Layout {
  kOneGraph,      // launch k on stream k % S, all captured into ONE graph (fork at the start, join at the end)
  kGraphPerChain, // the same chains, one captured graph per stream, launched together
  kFlagKernels,   // stream 1: the W kernels (+ a flag kernel behind each); stream 0: G, flag kernel, R ...; one graph per stream
};

// launches of chain `only` (or of every chain when only < 0) in launch order
template <bool SC1, bool PREFETCH>
static void enqueue(const Bufs& b, int iters, int nstreams, bool handshake, hipStream_t* st, int only) {
  const int L = iters * 3;
  for (int k = 0; k < L; ++k) {
    if (only >= 0 && k % nstreams != only) continue;
    hipStream_t s = st[k % nstreams];
    Dep d{nullptr, 0, 0, nullptr, b.err};
    const int kind = k % 3;   // 0 G, 1 W, 2 R
    if (handshake) {
      d.sig = b.cnt + k;
      if (k > 0) {
        d.wait = b.cnt + k - 1;
        d.expect = kind == 0 ? R_WGS : kind == 1 ? G_WGS : W_WGS;
      }
    }
    if (kind == 0) hipLaunchKernelGGL(k_gn<SC1>, dim3(G_WGS), dim3(256), 0, s, b.xo, b.xa, b.gamma, d);
    else if (kind == 1)
      hipLaunchKernelGGL((k_ws<SC1, PREFETCH>), dim3(N / 32, SPLITS), dim3(512), NW * 16384, s, b.xa,
                         b.pool + (size_t)((k / 3) % b.pool_slices) * W_HALVES, b.slab, d);
    else hipLaunchKernelGGL(k_red<SC1>, dim3(R_WGS), dim3(256), 0, s, b.slab, b.bias, b.xa, b.xo, d);
  }
}

// flag-kernel protocol: chain 0 = G_i, flag(cnt[3i]), R_i polling cnt[3i+1]; chain 1 = W_i polling cnt[3i], flag(cnt[3i+1])
template <bool PREFETCH>
static void enqueue_flags(const Bufs& b, int iters, hipStream_t* st, int only) {
  for (int i = 0; i < iters; ++i) {
    Dep none{nullptr, 0, 0, nullptr, b.err};
    if (only != 1) {
      hipLaunchKernelGGL(k_gn<false>, dim3(G_WGS), dim3(256), 0, st[0], b.xo, b.xa, b.gamma, none);
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st[0], b.cnt + 3 * i);
      Dep dr{b.cnt + 3 * i + 1, 1u, 1, nullptr, b.err};
      hipLaunchKernelGGL(k_red<false>, dim3(R_WGS), dim3(256), 0, st[0], b.slab, b.bias, b.xa, b.xo, dr);
    }
    if (only != 0) {
      Dep dw{b.cnt + 3 * i, 1u, 1, nullptr, b.err};
      hipLaunchKernelGGL((k_ws<false, PREFETCH>), dim3(N / 32, SPLITS), dim3(512), NW * 16384, st[1], b.xa,
                         b.pool + (size_t)(i % b.pool_slices) * W_HALVES, b.slab, dw);
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st[1], b.cnt + 3 * i + 1);
    }
  }
}

struct Result { float us_per_launch; unsigned long long checksum; unsigned err; };

template <bool SC1, bool PREFETCH>
static Result run(const Bufs& b, const std::vector<half_t>& x0, int iters, int nstreams, bool handshake, hipStream_t* st, int reps,
                  Layout layout = kOneGraph) {
  const int L = iters * 3;
  hipEvent_t fork, join[8], e0, e1;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  for (int i = 0; i < 8; ++i) CK(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<hipGraph_t> g;
  std::vector<hipGraphExec_t> ge;
  auto end_capture = [&](hipStream_t s) {
    hipGraph_t gg;
    hipGraphExec_t gx;
    CK(hipStreamEndCapture(s, &gg));
    CK(hipGraphInstantiate(&gx, gg, nullptr, nullptr, 0));
    g.push_back(gg);
    ge.push_back(gx);
  };
  if (layout == kOneGraph) {
    CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(b.cnt, 0, (size_t)L * 4, st[0]));
    CK(hipEventRecord(fork, st[0]));
    for (int i = 1; i < nstreams; ++i) CK(hipStreamWaitEvent(st[i], fork, 0));
    enqueue<SC1, PREFETCH>(b, iters, nstreams, handshake, st, -1);
    for (int i = 1; i < nstreams; ++i) {
      CK(hipEventRecord(join[i], st[i]));
      CK(hipStreamWaitEvent(st[0], join[i], 0));
    }
    end_capture(st[0]);
  } else {
    if (layout == kFlagKernels) nstreams = 2;
    for (int c = 0; c < nstreams; ++c) {
      CK(hipStreamBeginCapture(st[c], hipStreamCaptureModeThreadLocal));
      if (layout == kFlagKernels) enqueue_flags<PREFETCH>(b, iters, st, c);
      else enqueue<SC1, PREFETCH>(b, iters, nstreams, handshake, st, c);
      end_capture(st[c]);
    }
  }
  float best = 1e30f;
  Result r{};
  for (int rep = 0; rep < reps + 1; ++rep) {
    CK(hipMemcpy(b.xo, x0.data(), x0.size() * 2, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, st[0]));
    if (layout == kOneGraph) {
      CK(hipGraphLaunch(ge[0], st[0]));
    } else {   // counters cleared in front of every chain; the chains start together and are joined on stream 0
      CK(hipMemsetAsync(b.cnt, 0, (size_t)L * 4, st[0]));
      CK(hipEventRecord(fork, st[0]));
      for (int c = 1; c < nstreams; ++c) CK(hipStreamWaitEvent(st[c], fork, 0));
      for (int c = 0; c < nstreams; ++c) CK(hipGraphLaunch(ge[c], st[c]));
      for (int c = 1; c < nstreams; ++c) {
        CK(hipEventRecord(join[c], st[c]));
        CK(hipStreamWaitEvent(st[0], join[c], 0));
      }
    }
    CK(hipEventRecord(e1, st[0]));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  std::vector<half_t> out(x0.size());
  CK(hipMemcpy(out.data(), b.xo, out.size() * 2, hipMemcpyDeviceToHost));
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < out.size(); ++i) {
    unsigned short u;
    memcpy(&u, &out[i], 2);
    h = (h ^ u) * 1099511628211ull;
  }
  CK(hipMemcpy(&r.err, b.err, 4, hipMemcpyDeviceToHost));
  CK(hipMemset(b.err, 0, 4));
  r.us_per_launch = best * 1e3f / L;   // per launch of the 3 * iters WORK kernels (flag kernels are overhead, not work)
  r.checksum = h;
  for (auto x : ge) (void)hipGraphExecDestroy(x);
  for (auto x : g) (void)hipGraphDestroy(x);
  return r;
}

// Do two independent branches of ONE captured graph overlap?  Two chains of 20 launches of a 16-workgroup kernel that spins
// ~20 us each: 400 us when they run one after the other, 800 us when serialised.
__global__ void k_spin(long long ticks, unsigned* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks < 0) sink[0] = 1;
}
static void branch_probe(hipStream_t* st, unsigned* sink) {
  hipEvent_t fork, join, e0, e1;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float res[3];
  for (int mode = 0; mode < 3; ++mode) {   // 0: one chain of 40; 1: two chains of 20 in ONE graph; 2: two chains as two graphs
    hipGraph_t g[2];
    hipGraphExec_t ge[2];
    const int ng = mode == 2 ? 2 : 1;
    if (mode < 2) {
      CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
      if (mode == 1) {
        CK(hipEventRecord(fork, st[0]));
        CK(hipStreamWaitEvent(st[1], fork, 0));
      }
      for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_spin, dim3(16), dim3(64), 0, st[mode == 1 ? (i & 1) : 0], 2000LL, sink);
      if (mode == 1) {
        CK(hipEventRecord(join, st[1]));
        CK(hipStreamWaitEvent(st[0], join, 0));
      }
      CK(hipStreamEndCapture(st[0], &g[0]));
      CK(hipGraphInstantiate(&ge[0], g[0], nullptr, nullptr, 0));
    } else {
      for (int c = 0; c < 2; ++c) {
        CK(hipStreamBeginCapture(st[c], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_spin, dim3(16), dim3(64), 0, st[c], 2000LL, sink);
        CK(hipStreamEndCapture(st[c], &g[c]));
        CK(hipGraphInstantiate(&ge[c], g[c], nullptr, nullptr, 0));
      }
    }
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, st[0]));
      if (mode < 2) {
        CK(hipGraphLaunch(ge[0], st[0]));
      } else {
        CK(hipEventRecord(fork, st[0]));
        CK(hipStreamWaitEvent(st[1], fork, 0));
        CK(hipGraphLaunch(ge[0], st[0]));
        CK(hipGraphLaunch(ge[1], st[1]));
        CK(hipEventRecord(join, st[1]));
        CK(hipStreamWaitEvent(st[0], join, 0));
      }
      CK(hipEventRecord(e1, st[0]));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    res[mode] = best * 1e3f;
    for (int c = 0; c < ng; ++c) {
      (void)hipGraphExecDestroy(ge[c]);
      (void)hipGraphDestroy(g[c]);
    }
  }
  printf("branches: 40 spin kernels of 20 us: one chain %.0f us | two chains of 20 in ONE captured graph %.0f us | two chains as TWO graphs on two "
         "streams %.0f us   (overlap = half of the first figure)\n", res[0], res[1], res[2]);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 100;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  Bufs b{};
  b.pool_slices = 52;   // 52 x 29.5 MB = 1.53 GB: a slice is touched again 1.5 GB later (Infinity Cache: 256 MB)
  CK(hipMalloc(&b.pool, b.pool_slices * W_HALVES * 2));
  CK(hipMalloc(&b.xo, (size_t)M * C * 2));
  CK(hipMalloc(&b.xa, (size_t)M * C * 2));
  CK(hipMalloc(&b.slab, (size_t)SPLITS * M * N * 4));
  CK(hipMalloc(&b.gamma, C * 4));
  CK(hipMalloc(&b.bias, N * 4));
  CK(hipMalloc(&b.cnt, 4096 * 4));
  CK(hipMalloc(&b.err, 4));
  CK(hipMemset(b.err, 0, 4));
  {
    std::vector<half_t> w(W_HALVES);
    unsigned s = 12345u;
    for (size_t sl = 0; sl < b.pool_slices; ++sl) {
      for (size_t i = 0; i < W_HALVES; ++i) {
        s = s * 1664525u + 1013904223u;
        w[i] = (half_t)(((int)(s >> 9) % 2001 - 1000) * 1e-5f);   // +-0.01: the conv output stays O(1)
      }
      CK(hipMemcpy(b.pool + sl * W_HALVES, w.data(), W_HALVES * 2, hipMemcpyHostToDevice));
    }
    std::vector<float> gm(C), bs(N);
    for (int i = 0; i < C; ++i) {
      s = s * 1664525u + 1013904223u;
      gm[i] = 1.0f + ((int)(s >> 9) % 201 - 100) * 1e-3f;
      s = s * 1664525u + 1013904223u;
      bs[i] = ((int)(s >> 9) % 201 - 100) * 1e-3f;
    }
    CK(hipMemcpy(b.gamma, gm.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.bias, bs.data(), N * 4, hipMemcpyHostToDevice));
  }
  std::vector<half_t> x0((size_t)M * C);
  {
    unsigned s = 777u;
    for (auto& v : x0) {
      s = s * 1664525u + 1013904223u;
      v = (half_t)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
    }
  }
  hipStream_t st[4];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void*)k_ws<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 16384));
  CK(hipFuncSetAttribute((const void*)k_ws<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 16384));
  CK(hipFuncSetAttribute((const void*)k_ws<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 16384));
  CK(hipFuncSetAttribute((const void*)k_ws<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 16384));

  branch_probe(st, b.cnt);
  printf("overlap ubench: %d x (GroupNorm 64 wg -> weight-stream conv 200 wg x 8 waves, 29.5 MB cold -> combine 160 wg) = %d launches\n", iters,
         iters * 3);
  const Result base = run<false, false>(b, x0, iters, 1, false, st, reps);
  printf("%-66s %7.2f us per launch   checksum %016llx\n", "serial (one stream, stream order)", base.us_per_launch, base.checksum);
  auto report = [&](const char* name, const Result& r) {
    printf("%-66s %7.2f us per launch   %s%s   (%+.1f %%)\n", name, r.us_per_launch, r.checksum == base.checksum ? "bit-identical" : "MISMATCH",
           r.err ? "  SPIN TIMEOUT" : "", 100.0 * (r.us_per_launch / base.us_per_launch - 1.0));
    fflush(stdout);
  };
  report("serial, weights requested first (control)", run<false, true>(b, x0, iters, 1, false, st, reps));
  report("serial + signal/poll code, plain stores + release", run<false, false>(b, x0, iters, 1, true, st, reps));
  report("serial + signal/poll code, sc1 stores", run<true, false>(b, x0, iters, 1, true, st, reps));
  // ONE captured graph with S chains: kept short - ROCm 7.2 serialises the chains, every launch of chain 0 waits out its spin bound
  report("ONE graph, 2 chains, weights before the poll, plain + release", run<false, true>(b, x0, 8, 2, true, st, 1));
  for (int ns = 2; ns <= 3; ++ns) {
    char nm[96];
    snprintf(nm, sizeof nm, "%d graphs on %d streams, poll first, plain + release", ns, ns);
    report(nm, run<false, false>(b, x0, iters, ns, true, st, reps, kGraphPerChain));
    snprintf(nm, sizeof nm, "%d graphs on %d streams, weights before the poll, plain + release", ns, ns);
    report(nm, run<false, true>(b, x0, iters, ns, true, st, reps, kGraphPerChain));
    snprintf(nm, sizeof nm, "%d graphs on %d streams, weights before the poll, sc1 stores", ns, ns);
    report(nm, run<true, true>(b, x0, iters, ns, true, st, reps, kGraphPerChain));
  }
  report("flag kernels, convs on their own stream, poll first", run<false, false>(b, x0, iters, 2, false, st, reps, kFlagKernels));
  report("flag kernels, convs on their own stream, weights before the poll", run<false, true>(b, x0, iters, 2, false, st, reps, kFlagKernels));
  // control: the hand-shake switched off on two streams must NOT reproduce the chain (it proves the check can fail)
  const Result bad = run<false, true>(b, x0, iters, 2, false, st, 1, kGraphPerChain);
  printf("%-66s %7.2f us per launch   %s (expected: MISMATCH - no dependency at all)\n", "2 graphs on 2 streams WITHOUT hand-shake (negative control)",
         bad.us_per_launch, bad.checksum == base.checksum ? "bit-identical" : "MISMATCH");
  return 0;
}
