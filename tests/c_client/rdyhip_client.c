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

#include "case_io.h"

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

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  CaseFile cf;
  if (case_read(argv[1], &cf)) return 1;
  const int32_t *hdr = cf.hdr;
  const double  *scal = cf.scal;
  const int32_t  nc = hdr[0], no = hdr[1], nb = hdr[4];
  RDyHipMesh      m = cf.mesh;
  RDyHipBoundary *b = cf.boundaries;
  double **bvals = cf.bvals, *mannings = cf.mannings, *extsrc = cf.extsrc, *u = cf.u, *f_in = cf.f_in, *f_exp = cf.f_exp, *pv_exp = cf.pv_exp;
  const double courant_exp = cf.courant_exp;

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
