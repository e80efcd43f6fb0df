// Device-side forcing ingestion: the per-step loops of RDyApplyForcing
// (src/forcing/rdyforcing.c:688-770) that fill the operator's external-source
// and Dirichlet-value arrays, kept on the GPU so that no host array crosses
// PCIe between RHS evaluations.  Pure copies / one multiply per value: the
// results are bit-identical to the reference's host loops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdyhip {

// RDyForcingSetConstantRainfall / RDyForcingSetHomogeneousData (src/forcing/rdyforcing_dataset.c:282-288, 320-344)
// followed by SetRegionalSourceComponent (src/rdydata.c:225-250): ext[ids[i]][comp] = value
__global__ void forcing_fill_kernel(int n, const int32_t *__restrict__ ids, double value, double *__restrict__ dst, int ncomp, int comp) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int o                    = ids ? ids[i] : i;
    dst[(int64_t)o * ncomp + comp] = value;
  }
}

// RDyForcingSetRasterData (rdyforcing_dataset.c:295-314; stride 1, scale mm/h -> m/s) and
// RDyForcingSetUnstructuredData (rdyforcing_dataset.c:350-373; scale 1):
// ext[ids[i]][comp] = data[map[i] * stride + offset] * scale
template <bool SCALE>
__global__ void forcing_gather_kernel(int n, const int32_t *__restrict__ ids, const double *__restrict__ data, const int32_t *__restrict__ map,
                                      int64_t stride, int64_t offset, double scale, double *__restrict__ dst, int ncomp, int comp) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int    o                 = ids ? ids[i] : i;
    const double v                 = data[(int64_t)map[i] * stride + offset];
    dst[(int64_t)o * ncomp + comp] = SCALE ? v * scale : v;
  }
}

// RDyForcingSetHomogeneousBoundary (rdyforcing_dataset.c:380-406): bvalues[e] = [h, 0, 0]
__global__ void forcing_fill_boundary_kernel(int n, double h, double *__restrict__ bvalues) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * (int64_t)n; i += (int64_t)gridDim.x * blockDim.x) bvalues[i] = (i % 3 == 0) ? h : 0.0;
}

// RDyForcingSetUnstructuredData on a boundary dataset (stride 3): bvalues[e][c] = data[map[e] * stride + c + offset]
__global__ void forcing_gather_boundary_kernel(int n, const double *__restrict__ data, const int32_t *__restrict__ map, int64_t stride,
                                               int64_t offset, double *__restrict__ bvalues) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * (int64_t)n; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i / 3), c = (int)(i - 3 * (int64_t)e);
    bvalues[i]  = data[(int64_t)map[e] * stride + c + offset];
  }
}

// RDyForcingCreateRasterDatasetMapping / RDyForcingCreateUnstructuredDatasetMap
// (src/forcing/rdyforcing_map.c:111-141, 77-104): brute-force nearest data point,
// first index wins ties (strict <).  One thread per mesh point; the data points are
// streamed through LDS in chunks so each is read from HBM once per workgroup.
// `min_dist0` < 0 selects the unstructured variant (point 0 is the initial minimum);
// otherwise a point must be strictly nearer than min_dist0 to be taken and the
// map entry is left untouched when none is (as the reference's loop does).
constexpr int NN_BLOCK = 256;
__global__ __launch_bounds__(NN_BLOCK) void forcing_nearest_kernel(int n, const double *__restrict__ xc, const double *__restrict__ yc, int ndata,
                                                                  const double *__restrict__ dx_, const double *__restrict__ dy_, double min_dist0,
                                                                  int32_t *__restrict__ map) {
  __shared__ double sx[NN_BLOCK], sy[NN_BLOCK];
  const int    i     = blockIdx.x * NN_BLOCK + threadIdx.x;
  const bool   live  = i < n;
  const double x     = live ? xc[i] : 0.0, y = live ? yc[i] : 0.0;
  double       best  = min_dist0;
  int          besti = -1;
  for (int k0 = 0; k0 < ndata; k0 += NN_BLOCK) {
    const int k = k0 + threadIdx.x;
    __syncthreads();
    if (k < ndata) {
      sx[threadIdx.x] = dx_[k];
      sy[threadIdx.x] = dy_[k];
    }
    __syncthreads();
    const int m = min(NN_BLOCK, ndata - k0);
    for (int j = 0; j < m; ++j) {
      const double ddx  = x - sx[j];
      const double ddy  = y - sy[j];
      const double dist = __dsqrt_rn(__dadd_rn(__dmul_rn(ddx, ddx), __dmul_rn(ddy, ddy)));  // PetscPowReal(dx*dx + dy*dy, 0.5), no FMA
      if ((besti < 0 && min_dist0 < 0.0) || dist < best) {
        best  = dist;
        besti = k0 + j;
      }
    }
  }
  if (live && besti >= 0) map[i] = besti;
}

}  // namespace rdyhip
