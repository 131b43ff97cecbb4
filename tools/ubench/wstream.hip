// Weight-streaming probe (round 4): how fast does the chip stream a COLD weight matrix when every workgroup walks its own
// 64-row x 64-k tiles (8 KiB per K step, 16 B per lane, 8 lanes per 128-B row segment) -
//   layout 0: rows of a [N][K] fp16 matrix, K*2 bytes apart (what the conv / GEMM kernels read today: 64 scattered 128-B
//             segments per step),
//   layout 1: the same bytes pre-tiled so that a step's 8 KiB are contiguous?
// hipcc --offload-arch=gfx950 -O3 tools/ubench/wstream.hip -o tools/ubench/wstream && tools/ubench/wstream
// SD_PRE=1 (round 6): between the flush and the timed stream another kernel READS the matrix once (what a prefetch issued by the
// previous launch of the step would leave behind: the bytes in the Infinity Cache and in whichever XCD's L2 touched them)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

// grid = n_tiles * splits; workgroup (nt, sp) streams K-steps [sp*steps, (sp+1)*steps) of n-tile nt; DEPTH steps in flight
template <int DEPTH>
__global__ __launch_bounds__(256) void k_stream(const char* __restrict__ w, int K2 /* row bytes */, int steps, int nsteps_total, int tiled,
                                                unsigned* sink) {
  const int tid = threadIdx.x;
  const int nt = blockIdx.x, sp = blockIdx.y;
  const int row = tid >> 3, seg = tid & 7;   // 32 rows x 8 segments per pass, two passes = 64 rows
  uint4v acc = {0, 0, 0, 0};
  uint4v buf[DEPTH][2];
  auto addr = [&](int step, int pass) -> const char* {
    const int r = row + 32 * pass;
    const size_t kt = (size_t)sp * steps + step;
    if (tiled) return w + (((size_t)nt * nsteps_total + kt) * 64 + r) * 128 + seg * 16;
    return w + ((size_t)nt * 64 + r) * K2 + kt * 128 + seg * 16;
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < steps) {
      buf[d][0] = *reinterpret_cast<const uint4v*>(addr(d, 0));
      buf[d][1] = *reinterpret_cast<const uint4v*>(addr(d, 1));
    }
  for (int s0 = 0; s0 < steps; s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int s = s0 + d;
      if (s < steps) {
        acc ^= buf[d][0] ^ buf[d][1];
        if (s + DEPTH < steps) {
          buf[d][0] = *reinterpret_cast<const uint4v*>(addr(s + DEPTH, 0));
          buf[d][1] = *reinterpret_cast<const uint4v*>(addr(s + DEPTH, 1));
        }
      }
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = acc[0];
}

// evict the L2s and the Infinity Cache with CLEAN lines (a read-only sweep: inside the UNet step a layer's weights are preceded by
// 1.7 GB of other layers' weights, not by dirty data; SD_FLUSH_DIRTY=1 sweeps read-modify-write instead)
__global__ void k_flush(unsigned* p, size_t n, int dirty, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (dirty) p[i] += 1;
    else acc ^= p[i];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

static size_t pre_bytes = 0;
static int pre_grid = 2048;
template <int DEPTH>
float run(const char* w, int K, int n_tiles, int splits, int steps, int nsteps, unsigned* flush, size_t flush_bytes, unsigned* sink,
          hipEvent_t e0, hipEvent_t e1) {
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    static const int dirty = getenv("SD_FLUSH_DIRTY") != nullptr;
    hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, flush, flush_bytes / 4, dirty, sink);
    static const int pre = getenv("SD_PRE") != nullptr;
    if (pre) hipLaunchKernelGGL(k_flush, dim3(pre_grid), dim3(256), 0, 0, (unsigned*)w, pre_bytes / 4, 0, sink);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_stream<DEPTH>, dim3(n_tiles, splits), dim3(256), 0, 0, w, K * 2, steps, nsteps, 0, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  const size_t flush_bytes = (size_t)768 << 20;
  unsigned *flush, *sink;
  CK(hipMalloc(&flush, flush_bytes));
  CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(flush, 0, flush_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // cold streaming of one weight matrix: workgroups (of 4 waves) x loads in flight per lane (2 per step)
  struct Case { const char* name; int N, K; };
  const Case cases[] = {{"conv3x3 1280->1280", 1280, 11520}, {"geglu 1280->10240", 10240, 1280}, {"conv3x3 2560->1280", 1280, 23040}};
  for (const Case& c : cases) {
    const size_t bytes = (size_t)c.N * c.K * 2;
    char* w;
    CK(hipMalloc(&w, bytes));
    CK(hipMemset(w, 1, bytes));
    const int n_tiles = c.N / 64, nsteps = c.K / 64;
    pre_bytes = bytes;
    pre_grid = getenv("SD_PRE_GRID") ? atoi(getenv("SD_PRE_GRID")) : 2048;
    for (int splits : {1, 2, 5, 10, 20, 30, 45, 60, 90, 180}) {
      if (nsteps % splits) continue;
      const int steps = nsteps / splits, wgs = n_tiles * splits;
      if (wgs < 100 || wgs > 8192 || steps < 1) continue;
      const double moved = (double)wgs * steps * 8192.0;
      const float t3 = run<3>(w, c.K, n_tiles, splits, steps, nsteps, flush, flush_bytes, sink, e0, e1);
      const float t6 = run<6>(w, c.K, n_tiles, splits, steps, nsteps, flush, flush_bytes, sink, e0, e1);
      const float t12 = run<12>(w, c.K, n_tiles, splits, steps, nsteps, flush, flush_bytes, sink, e0, e1);
      const float t24 = run<24>(w, c.K, n_tiles, splits, steps, nsteps, flush, flush_bytes, sink, e0, e1);
      printf("%-20s %5d WGs x %3d steps: depth 3 %6.1f us (%4.2f TB/s) | 6 %6.1f (%4.2f) | 12 %6.1f (%4.2f) | 24 %6.1f (%4.2f)\n", c.name, wgs,
             steps, t3 * 1e3, moved / t3 / 1e9, t6 * 1e3, moved / t6 / 1e9, t12 * 1e3, moved / t12 / 1e9, t24 * 1e3, moved / t24 / 1e9);
    }
    CK(hipFree(w));
  }
  return 0;
}
