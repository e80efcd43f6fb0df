// Does HBM care whether a tile's K per-cell streams come from K mesh-wide arrays (2 KB from each) or from one
// tile-blocked array (K x 2 KB contiguous)?  Persistent workgroups walk 256-cell tiles as the RHS kernel does.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_layout_probe tools/probes/stream_layout_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int TILE = 256;

template <int K, bool BLOCKED>
__global__ __launch_bounds__(TILE) void probe(const double *__restrict__ src, double *__restrict__ dst, int ntiles, size_t plane) {
  const int x = blockIdx.x & 7, per = (ntiles + 7) / 8, step = gridDim.x >> 3;
  const int hi = min((x + 1) * per, ntiles);
  for (int t = x * per + (blockIdx.x >> 3); t < hi; t += step) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const size_t off = BLOCKED ? ((size_t)t * K + k) * TILE + threadIdx.x : (size_t)k * plane + (size_t)t * TILE + threadIdx.x;
      acc += __builtin_nontemporal_load(src + off);
    }
    __builtin_nontemporal_store(acc, dst + (size_t)t * TILE + threadIdx.x);
  }
}

template <int K, bool BLOCKED>
int run(const double *src, double *dst, int ntiles, size_t plane, int grid) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<K, BLOCKED>), dim3(grid), dim3(TILE), 0, 0, src, dst, ntiles, plane);
  CK(hipEventRecord(a));
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<K, BLOCKED>), dim3(grid), dim3(TILE), 0, 0, src, dst, ntiles, plane);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)ntiles * TILE * 8 * (K + 1);
  printf("K=%2d %-8s grid=%4d  %.4f ms  %.0f GB/s\n", K, BLOCKED ? "blocked" : "planes", grid, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
  return 0;
}

int main() {
  const int ntiles = 39063;  // 10 M cells
  const int KMAX = 20;
  const size_t plane = (size_t)ntiles * TILE;
  double *src, *dst;
  CK(hipMalloc(&src, plane * KMAX * 8)); CK(hipMalloc(&dst, plane * 8));
  CK(hipMemset(src, 0, plane * KMAX * 8));
  for (int grid : {768, 1024, 2048}) {
    if (run<4, false>(src, dst, ntiles, plane, grid) || run<4, true>(src, dst, ntiles, plane, grid)) return 1;
    if (run<10, false>(src, dst, ntiles, plane, grid) || run<10, true>(src, dst, ntiles, plane, grid)) return 1;
    if (run<20, false>(src, dst, ntiles, plane, grid) || run<20, true>(src, dst, ntiles, plane, grid)) return 1;
  }
  return 0;
}
