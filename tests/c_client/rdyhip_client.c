/* A plain-C11 client of include/rdyhip.h (no C++, no Python, no torch): reads a
 * case file written by tests/test_gpu_c_client.py, creates the operator,
 * applies it to device buffers it allocated itself with the HIP runtime API,
 * and compares F, the primitive variables and the Courant diagnostic with the
 * expected values in the file (computed by the CPU oracle).
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude rdyhip_client.c \
 *       -Lrdycore_amd/csrc -lrdyhip -L/opt/rocm/lib -lamdhip64 -lm -o rdyhip_client
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rdyhip.h"

#define CHECK(call)                                                                      \
  do {                                                                                   \
    int rc_ = (call);                                                                    \
    if (rc_ != 0) {                                                                      \
      fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, rdyhip_last_error());          \
      return 2;                                                                          \
    }                                                                                    \
  } while (0)
#define HIPCHECK(call)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));                  \
      return 3;                                                                          \
    }                                                                                    \
  } while (0)

static void *rd(FILE *f, size_t n, size_t sz) {
  void *p = malloc(n * sz > 0 ? n * sz : 1);
  if (n && fread(p, sz, n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(4);
  }
  return p;
}

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[8]; /* num_cells, num_owned, num_edges, num_internal, num_boundaries, source_method, overwrite, reserved */
  if (fread(hdr, sizeof(int32_t), 8, f) != 8) return 4;
  double scal[4]; /* tiny_h, h_anuga, xq2018_threshold, dt */
  if (fread(scal, sizeof(double), 4, f) != 4) return 4;
  const int32_t nc = hdr[0], no = hdr[1], ne = hdr[2], ni = hdr[3], nb = hdr[4];

  RDyHipMesh m = {0};
  m.num_cells = nc; m.num_owned_cells = no; m.num_edges = ne; m.num_internal_edges = ni;
  m.cell_is_owned       = rd(f, nc, 4);
  m.cell_local_to_owned = rd(f, nc, 4);
  m.cell_global_ids     = rd(f, nc, 8);
  m.cell_areas          = rd(f, nc, 8);
  m.cell_dz_dx          = rd(f, nc, 8);
  m.cell_dz_dy          = rd(f, nc, 8);
  m.edge_cell_ids       = rd(f, 2 * (size_t)ne, 4);
  m.edge_internal_ids   = rd(f, ni, 4);
  m.edge_global_ids     = rd(f, ne, 8);
  m.edge_lengths        = rd(f, ne, 8);
  m.edge_cn             = rd(f, ne, 8);
  m.edge_sn             = rd(f, ne, 8);

  RDyHipBoundary *b      = calloc(nb > 0 ? nb : 1, sizeof(*b));
  double        **bvals  = calloc(nb > 0 ? nb : 1, sizeof(double *));
  for (int i = 0; i < nb; ++i) {
    int32_t bh[2]; /* num_edges, condition type */
    if (fread(bh, 4, 2, f) != 2) return 4;
    b[i].num_edges      = bh[0];
    b[i].condition_type = bh[1];
    b[i].edge_ids       = rd(f, bh[0], 4);
    bvals[i]            = rd(f, 3 * (size_t)bh[0], 8);
  }
  double *mannings = rd(f, no, 8);
  double *extsrc   = rd(f, 3 * (size_t)no, 8); /* [comp][owned] */
  double *u        = rd(f, 3 * (size_t)nc, 8);
  double *f_in     = rd(f, 3 * (size_t)no, 8);
  double *f_exp    = rd(f, 3 * (size_t)no, 8);
  double *pv_exp   = rd(f, 3 * (size_t)no, 8);
  double  courant_exp;
  if (fread(&courant_exp, 8, 1, f) != 1) return 4;
  fclose(f);

  RDyHipConfig cfg = {scal[0], scal[1], scal[2], hdr[5], RDYHIP_RIEMANN_ROE};
  RDyHipOperator op = NULL;
  CHECK(rdyhip_create(&cfg, &m, nb, b, &op));
  CHECK(rdyhip_set_mannings(op, no, NULL, mannings));
  for (int c = 0; c < 3; ++c) CHECK(rdyhip_set_external_source(op, c, no, NULL, extsrc + (size_t)c * no));
  for (int i = 0; i < nb; ++i) CHECK(rdyhip_set_boundary_values(op, i, 0, 3, b[i].num_edges, bvals[i]));

  double *d_u = NULL, *d_f = NULL;
  HIPCHECK(hipMalloc((void **)&d_u, sizeof(double) * 3 * (size_t)nc));
  HIPCHECK(hipMalloc((void **)&d_f, sizeof(double) * 3 * (size_t)no));
  HIPCHECK(hipMemcpy(d_u, u, sizeof(double) * 3 * (size_t)nc, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_f, f_in, sizeof(double) * 3 * (size_t)no, hipMemcpyHostToDevice));
  hipStream_t st;
  HIPCHECK(hipStreamCreate(&st));
  if (hdr[6]) {
    CHECK(rdyhip_rhs_function(op, scal[3], d_u, d_f, st)); /* OperatorRHSFunction: zero + reset + apply */
  } else {
    CHECK(rdyhip_reset_diagnostics(op, st));
    CHECK(rdyhip_apply(op, scal[3], d_u, d_f, st)); /* ApplyOperator: f += F(u) */
  }
  CHECK(rdyhip_update_diagnostics(op, st));
  RDyHipCourant cd;
  CHECK(rdyhip_get_diagnostics(op, &cd));
  double *f_out = malloc(sizeof(double) * 3 * (size_t)no), *pv_out = malloc(sizeof(double) * 3 * (size_t)no);
  HIPCHECK(hipMemcpy(f_out, d_f, sizeof(double) * 3 * (size_t)no, hipMemcpyDeviceToHost));
  double *d_pv = NULL;
  int64_t npv  = 0;
  CHECK(rdyhip_field_ptr(op, RDYHIP_FIELD_PRIMITIVE_VARIABLES, &d_pv, &npv));
  HIPCHECK(hipMemcpy(pv_out, d_pv, sizeof(double) * (size_t)npv, hipMemcpyDeviceToHost));

  double ef = 0, sf = 1, ep = 0, sp = 1;
  for (size_t i = 0; i < 3 * (size_t)no; ++i) {
    ef = fmax(ef, fabs(f_out[i] - f_exp[i]));
    sf = fmax(sf, fabs(f_exp[i]));
    ep = fmax(ep, fabs(pv_out[i] - pv_exp[i]));
    sp = fmax(sp, fabs(pv_exp[i]));
  }
  printf("cells %d  rhs_linf %.3e  pv_linf %.3e  courant %.15g (expected %.15g)\n", no, ef / sf, ep / sp, cd.max_courant_num, courant_exp);

  /* the whole forward-Euler step in one call: u2[owned] = u[owned] + dt F, F not requested */
  double ee = 0.0;
  if (hdr[6]) {
    double *d_u2 = NULL, *u2 = malloc(sizeof(double) * 3 * (size_t)nc);
    HIPCHECK(hipMalloc((void **)&d_u2, sizeof(double) * 3 * (size_t)nc));
    HIPCHECK(hipMemset(d_u2, 0, sizeof(double) * 3 * (size_t)nc));
    CHECK(rdyhip_euler_step(op, RDYHIP_PHASE_ALL, RDYHIP_PHASE_RESET_DIAGNOSTICS, scal[3], d_u, d_u2, NULL, st));
    HIPCHECK(hipStreamSynchronize(st));
    HIPCHECK(hipMemcpy(u2, d_u2, sizeof(double) * 3 * (size_t)nc, hipMemcpyDeviceToHost));
    for (int32_t c = 0; c < nc; ++c) {
      if (!((const int32_t *)m.cell_is_owned)[c]) continue;
      const int32_t o = ((const int32_t *)m.cell_local_to_owned)[c];
      for (int k = 0; k < 3; ++k) ee = fmax(ee, fabs(u2[3 * c + k] - (u[3 * c + k] + scal[3] * f_exp[3 * o + k])));
    }
    printf("euler_step linf %.3e\n", ee / sf);
    ee /= sf;
    HIPCHECK(hipFree(d_u2));
    free(u2);
  }
  /* forcing ingestion on the device: a constant water source over the whole domain */
  CHECK(rdyhip_forcing_fill_source(op, 0, no, NULL, 2.5e-6, st));
  HIPCHECK(hipStreamSynchronize(st));
  double *d_src = NULL, *src_out = malloc(sizeof(double) * 3 * (size_t)no);
  int64_t nsrc  = 0;
  CHECK(rdyhip_field_ptr(op, RDYHIP_FIELD_EXTERNAL_SOURCES, &d_src, &nsrc));
  HIPCHECK(hipMemcpy(src_out, d_src, sizeof(double) * (size_t)nsrc, hipMemcpyDeviceToHost));
  int src_ok = nsrc == 3 * (int64_t)no;
  for (int32_t o = 0; o < no && src_ok; ++o)
    src_ok = src_out[3 * o] == 2.5e-6 && src_out[3 * o + 1] == extsrc[(size_t)no + o] && src_out[3 * o + 2] == extsrc[2 * (size_t)no + o];
  printf("forcing_fill_source %s\n", src_ok ? "ok" : "MISMATCH");
  free(src_out);
  CHECK(rdyhip_destroy(&op));
  HIPCHECK(hipFree(d_u));
  HIPCHECK(hipFree(d_f));
  const int ok = ef / sf <= 1e-10 && ep / sp <= 1e-10 && ee <= 1e-10 && src_ok &&
                 fabs(cd.max_courant_num - courant_exp) <= 1e-12 * fmax(1.0, courant_exp);
  return ok ? 0 : 5;
}
