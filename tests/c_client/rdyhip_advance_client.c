/* RDyAdvance in plain C11 on the native operator (no Python, no torch, no PETSc): what a TS host's custom step does with
 * rdyhip_euler_step -- the forward-Euler update fused into the RHS kernel's stores, two state arrays ping-pong -- and the
 * Courant -> dt rule of src/rdyadvance.c:303-343 when adaptive stepping is on.  The state never leaves the device inside an
 * interval; the host reads the 16-byte Courant struct once per interval (adaptive only).
 *
 *   rdyhip_advance_client case.bin out.bin num_intervals interval_seconds dt adaptive [target_courant max_increase]
 *
 * Writes to out.bin: int64 number of steps taken, double final dt, double final time, then the final state [num_cells][3].
 * tests/test_gpu_c_client.py compares it with the same loop driven by the CPU oracle. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <string.h>

#include "case_io.h"

#define CHECK(call)                                                             \
  do {                                                                          \
    int rc_ = (call);                                                           \
    if (rc_ != 0) {                                                             \
      fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, rdyhip_last_error()); \
      return 2;                                                                 \
    }                                                                           \
  } while (0)
#define HIPCHECK(call)                                                  \
  do {                                                                  \
    hipError_t e_ = (call);                                             \
    if (e_ != hipSuccess) {                                             \
      fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_)); \
      return 3;                                                         \
    }                                                                   \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 7) return 1;
  CaseFile c;
  if (case_read(argv[1], &c)) return 1;
  const int    num_intervals = atoi(argv[3]);
  const double interval      = atof(argv[4]);
  double       dt            = atof(argv[5]);
  const int    adaptive      = atoi(argv[6]);
  const double target        = argc > 7 ? atof(argv[7]) : 0.5;
  const double max_increase  = argc > 8 ? atof(argv[8]) : 2.0;
  const int32_t nc = c.hdr[0], no = c.hdr[1], nb = c.hdr[4];

  RDyHipConfig   cfg = {c.scal[0], c.scal[1], c.scal[2], c.hdr[5], RDYHIP_RIEMANN_ROE};
  RDyHipOperator op  = NULL;
  CHECK(rdyhip_create(&cfg, &c.mesh, nb, c.boundaries, &op));
  hipStream_t st;
  HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  /* operator inputs through the stream-ordered setters: nothing synchronises the device */
  CHECK(rdyhip_set_mannings_on(op, no, NULL, c.mannings, st));
  for (int k = 0; k < 3; ++k) CHECK(rdyhip_set_external_source_on(op, k, no, NULL, c.extsrc + (size_t)k * no, st));
  for (int i = 0; i < nb; ++i) CHECK(rdyhip_set_boundary_values_on(op, i, 0, 3, c.boundaries[i].num_edges, c.bvals[i], st));

  const size_t bytes = sizeof(double) * 3 * (size_t)nc;
  double      *d_u[2];
  HIPCHECK(hipMalloc((void **)&d_u[0], bytes));
  HIPCHECK(hipMalloc((void **)&d_u[1], bytes));
  HIPCHECK(hipMemcpyAsync(d_u[0], c.u, bytes, hipMemcpyHostToDevice, st));
  HIPCHECK(hipMemcpyAsync(d_u[1], c.u, bytes, hipMemcpyHostToDevice, st)); /* ghost rows of a one-rank run: none; keeps the array defined */

  int     cur = 0;
  int64_t steps = 0;
  double  time = 0.0, max_courant = -1.0; /* < 0: diagnostics not valid yet */
  for (int iv = 0; iv < num_intervals; ++iv) {
    if (adaptive && max_courant >= 0.0) { /* src/rdyadvance.c:308-330 */
      if (max_courant < target) {
        const double ratio = max_courant > 0.0 ? target / max_courant : INFINITY;
        dt *= fmin(ratio, max_increase);
        dt = fmin(dt, interval);
      } else {
        dt *= target / max_courant;
      }
    }
    const double t_end = time + interval;
    CHECK(rdyhip_reset_diagnostics(op, st));
    while (time < t_end * (1.0 - 1e-14)) {
      const double h = fmin(dt, t_end - time); /* TS_EXACTFINALTIME_MATCHSTEP */
      /* TSStep_Euler + OperatorRHSFunction (src/rdysetup.c:1120-1172) in one launch; F is never stored */
      CHECK(rdyhip_euler_step(op, RDYHIP_PHASE_ALL, RDYHIP_PHASE_RESET_DIAGNOSTICS, h, d_u[cur], d_u[1 - cur], NULL, st));
      cur = 1 - cur;
      time += h;
      ++steps;
    }
    if (adaptive) { /* UpdateOperatorDiagnostics: 16 bytes, the interval's only synchronisation */
      RDyHipCourant cd;
      CHECK(rdyhip_update_diagnostics(op, st));
      CHECK(rdyhip_get_diagnostics(op, &cd));
      max_courant = cd.max_courant_num;
    }
  }
  double *u_out = malloc(bytes);
  HIPCHECK(hipMemcpyAsync(u_out, d_u[cur], bytes, hipMemcpyDeviceToHost, st));
  HIPCHECK(hipStreamSynchronize(st));
  FILE *f = fopen(argv[2], "wb");
  if (!f) return 1;
  fwrite(&steps, sizeof(steps), 1, f);
  fwrite(&dt, sizeof(dt), 1, f);
  fwrite(&time, sizeof(time), 1, f);
  fwrite(u_out, 1, bytes, f);
  fclose(f);
  printf("steps %lld  final dt %.17g  time %.17g\n", (long long)steps, dt, time);
  CHECK(rdyhip_destroy(&op));
  HIPCHECK(hipFree(d_u[0]));
  HIPCHECK(hipFree(d_u[1]));
  HIPCHECK(hipStreamDestroy(st));
  return 0;
}
