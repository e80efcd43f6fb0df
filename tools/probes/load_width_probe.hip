// Does the achievable streaming rate depend on the width of the per-lane load?  Persistent workgroups walk 256-cell tiles as
// the RHS kernels do and sum K stream chunks of 2 KB per tile, loaded 8 bytes per lane (one chunk per instruction and
// workgroup: what the kernels do for their per-cell planes) or 16 bytes per lane (two chunks per instruction: the lower half
// of the workgroup takes chunk 2j, the upper half chunk 2j+1).
// build: hipcc --offload-arch=gfx950 -O3 -o load_width_probe tools/probes/load_width_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int TILE = 256;
typedef double d2 __attribute__((ext_vector_type(2)));

template <int K, int W, bool NT>
__global__ __launch_bounds__(TILE) void probe(const double *__restrict__ src, double *__restrict__ dst, int ntiles, size_t plane) {
  const int x = blockIdx.x & 7, per = (ntiles + 7) / 8, step = gridDim.x >> 3;
  const int hi = min((x + 1) * per, ntiles);
  for (int t = x * per + (blockIdx.x >> 3); t < hi; t += step) {
    double acc = 0.0;
    if (W == 8) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double *p = src + (size_t)k * plane + (size_t)t * TILE + threadIdx.x;
        acc += NT ? __builtin_nontemporal_load(p) : *p;
      }
    } else {
      const int half = threadIdx.x >> 7, l = threadIdx.x & 127;
#pragma unroll
      for (int k = 0; k < K; k += 2) {
        const d2 *p = reinterpret_cast<const d2 *>(src + (size_t)(k + half) * plane + (size_t)t * TILE) + l;
        const d2  v = NT ? __builtin_nontemporal_load(p) : *p;
        acc += v.x + v.y;
      }
    }
    double *q = dst + (size_t)t * TILE + threadIdx.x;
    if (NT) __builtin_nontemporal_store(acc, q); else *q = acc;
  }
}

template <int K, int W, bool NT>
int run(const double *src, double *dst, int ntiles, size_t plane, int grid) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<K, W, NT>), dim3(grid), dim3(TILE), 0, 0, src, dst, ntiles, plane);
  CK(hipEventRecord(a));
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<K, W, NT>), dim3(grid), dim3(TILE), 0, 0, src, dst, ntiles, plane);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)ntiles * TILE * 8 * (K + 1);
  printf("K=%2d  %2d B/lane  %s  grid=%4d  %.4f ms  %.0f GB/s\n", K, W, NT ? "nt   " : "plain", grid, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
  return 0;
}

int main() {
  const int ntiles = 39063;  // 10 M cells
  const int KMAX = 20;
  const size_t plane = (size_t)ntiles * TILE;
  double *src, *dst;
  CK(hipMalloc(&src, plane * KMAX * 8)); CK(hipMalloc(&dst, plane * 8));
  CK(hipMemset(src, 0, plane * KMAX * 8));
  for (int grid : {768, 1024, 2048, 4096}) {
    if (run<20, 8, true>(src, dst, ntiles, plane, grid) || run<20, 16, true>(src, dst, ntiles, plane, grid)) return 1;
    if (run<20, 8, false>(src, dst, ntiles, plane, grid) || run<20, 16, false>(src, dst, ntiles, plane, grid)) return 1;
  }
  return 0;
}
