/* The multi-rank path of include/rdyhip.h from a plain C host: N processes (forked here, before any HIP call), one rank each,
 * no Python, no torch, no MPI -- pipes stand in for the host's communicator:
 *
 *   rdyhip_halo_plan_create / _requests      what each rank knows locally: its ghost cells, their owner ranks, their global ids
 *   (pipes)                                  the ONE all-to-all of the plan (MPI_Alltoall + MPI_Alltoallv in RDycore)
 *   rdyhip_halo_plan_finish / _get           -> the arguments of rdyhip_halo_create
 *   rdyhip_halo_set_transport                the bytes of every exchange travel through the pipes too (several ranks share one
 *                                            GPU in the test, where RCCL cannot connect them; on a node it is ncclSend / ncclRecv)
 *   rdyhip_copy_owned_rows + rdyhip_rhs_overlapped     exactly the two calls of OperatorRHSFunctionHip (adapter/rdyhip_petsc.c)
 *   rdyhip_euler_step_overlapped (+ rdyhip_halo_fuse_pack)   a few whole Euler steps, against RHS + axpy
 *   rdyhip_update_diagnostics + a struct-max over the ranks   (src/operator.c:705-715, 879)
 *
 *   rdyhip_mr_client N prefix        reads prefix.<rank>.bin (tests/test_gpu_c_client.py: write_case with the ghosts' owners; the
 *                                    expected F rows and Courant number come from the single-rank oracle on the undivided mesh)
 * Exit code 0: every rank's RHS matches its rows to 1e-10 and the reduced Courant number matches. */
#define _POSIX_C_SOURCE 200809L
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include "case_io.h"

#define MAXR 8
static int g_rank, g_world;
static int g_rd[MAXR], g_wr[MAXR]; /* g_rd[q]: read end of the pipe q -> me; g_wr[q]: write end of me -> q */

#define CHECK(call)                                                                             \
  do {                                                                                          \
    int rc_ = (call);                                                                           \
    if (rc_ != 0) {                                                                             \
      fprintf(stderr, "rank %d: %s failed: %d (%s)\n", g_rank, #call, rc_, rdyhip_last_error()); \
      return 2;                                                                                 \
    }                                                                                           \
  } while (0)
#define HIPCHECK(call)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "rank %d: %s failed: %s\n", g_rank, #call, hipGetErrorString(e_)); \
      return 3;                                                                          \
    }                                                                                    \
  } while (0)

static int put(int q, const void *buf, size_t n) {
  const char *p = buf;
  while (n) {
    ssize_t k = write(g_wr[q], p, n);
    if (k <= 0) return 1;
    p += k;
    n -= (size_t)k;
  }
  return 0;
}
static int get(int q, void *buf, size_t n) {
  char *p = buf;
  while (n) {
    ssize_t k = read(g_rd[q], p, n);
    if (k <= 0) return 1;
    p += k;
    n -= (size_t)k;
  }
  return 0;
}
/* messages stay far below a pipe's 64 KB, so "everybody writes, then everybody reads" cannot block */
#define PIPE_LIMIT 32768

/* the exchange pattern, shared with the transport callback */
static int32_t        g_npeers;
static const int32_t *g_peers, *g_send_counts, *g_recv_counts;
static double        *g_hsend, *g_hrecv;
static int32_t        g_nsend, g_nrecv;

/* RDyHipTransportFn: d_send's per-peer slices -> the peers' d_recv slices, staged through the host and the pipes */
static int pipe_transport(void *ctx, const double *d_send, double *d_recv, int32_t ncomp, void *stream) {
  (void)ctx;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1; /* the pack launch has filled d_send */
  if (g_nsend && hipMemcpy(g_hsend, d_send, sizeof(double) * (size_t)g_nsend * ncomp, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  size_t so = 0, ro = 0;
  for (int32_t i = 0; i < g_npeers; ++i) {
    const size_t n = (size_t)g_send_counts[i] * ncomp;
    if (n * sizeof(double) > PIPE_LIMIT) return 3;
    if (n && put(g_peers[i], g_hsend + so, n * sizeof(double))) return 4;
    so += n;
  }
  for (int32_t i = 0; i < g_npeers; ++i) {
    const size_t n = (size_t)g_recv_counts[i] * ncomp;
    if (n && get(g_peers[i], g_hrecv + ro, n * sizeof(double))) return 5;
    ro += n;
  }
  if (g_nrecv && hipMemcpyAsync(d_recv, g_hrecv, sizeof(double) * (size_t)g_nrecv * ncomp, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return 6;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 7; /* g_hrecv is reused by the next exchange */
  return 0;
}

static int rank_main(const char *prefix) {
  char path[4096];
  snprintf(path, sizeof(path), "%s.%d.bin", prefix, g_rank);
  CaseFile c;
  if (case_read(path, &c)) {
    fprintf(stderr, "rank %d: cannot read %s\n", g_rank, path);
    return 1;
  }
  const int32_t nc = c.hdr[0], no = c.hdr[1], nb = c.hdr[4];
  if (!c.owner) {
    fprintf(stderr, "rank %d: the case file carries no owner ranks\n", g_rank);
    return 1;
  }
  HIPCHECK(hipSetDevice(0));
  RDyHipConfig cfg = {c.scal[0], c.scal[1], c.scal[2], c.hdr[5], RDYHIP_RIEMANN_ROE, c.hdr[7]};
  c.mesh.cell_zc   = c.zc;
  RDyHipOperator op = NULL;
  CHECK(rdyhip_create(&cfg, &c.mesh, nb, c.boundaries, &op));
  hipStream_t st;
  HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CHECK(rdyhip_set_mannings_on(op, no, NULL, c.mannings, st));
  for (int k = 0; k < 3; ++k) CHECK(rdyhip_set_external_source_on(op, k, no, NULL, c.extsrc + (size_t)k * no, st));
  for (int i = 0; i < nb; ++i) CHECK(rdyhip_set_boundary_values_on(op, i, 0, 3, c.boundaries[i].num_edges, c.bvals[i], st));

  /* ---- the plan: ghosts grouped by owner, ONE all-to-all, send lists resolved ---- */
  const int32_t *is_owned = c.mesh.cell_is_owned;
  int32_t        ng = 0;
  int32_t       *g_cell = malloc(sizeof(int32_t) * (size_t)(nc > 0 ? nc : 1)), *g_own = malloc(sizeof(int32_t) * (size_t)(nc > 0 ? nc : 1));
  int64_t       *g_key = malloc(sizeof(int64_t) * (size_t)(nc > 0 ? nc : 1));
  for (int32_t k = 0; k < nc; ++k)
    if (!is_owned[k]) {
      g_cell[ng] = k;
      g_own[ng]  = c.owner[k];
      g_key[ng]  = c.mesh.cell_global_ids[k];
      ++ng;
    }
  RDyHipHaloPlan plan;
  CHECK(rdyhip_halo_plan_create(g_world, g_rank, ng, g_cell, g_own, g_key, &plan));
  const int32_t *req_counts;
  const int64_t *req_keys;
  CHECK(rdyhip_halo_plan_requests(plan, &req_counts, &req_keys));
  int32_t in_counts[MAXR] = {0};
  size_t  off = 0;
  for (int q = 0; q < g_world; ++q) {
    if (q == g_rank) continue;
    if ((size_t)req_counts[q] * 8 + 4 > PIPE_LIMIT) return 6;
    if (put(q, &req_counts[q], 4)) return 6;
  }
  for (int q = 0; q < g_world; ++q) {
    if (q != g_rank && req_counts[q] && put(q, req_keys + off, sizeof(int64_t) * (size_t)req_counts[q])) return 6;
    off += (size_t)req_counts[q];
  }
  size_t tot = 0;
  for (int q = 0; q < g_world; ++q) {
    if (q != g_rank && get(q, &in_counts[q], 4)) return 6;
    tot += (size_t)in_counts[q];
  }
  int64_t *in_keys = malloc(sizeof(int64_t) * (tot ? tot : 1));
  off              = 0;
  for (int q = 0; q < g_world; ++q) {
    if (q != g_rank && in_counts[q] && get(q, in_keys + off, sizeof(int64_t) * (size_t)in_counts[q])) return 6;
    off += (size_t)in_counts[q];
  }
  CHECK(rdyhip_halo_plan_finish(plan, in_counts, in_keys, nc, is_owned, c.mesh.cell_global_ids));
  const int32_t *send_cells, *recv_cells;
  CHECK(rdyhip_halo_plan_get(plan, &g_npeers, &g_peers, &g_send_counts, &send_cells, &g_recv_counts, &recv_cells));
  for (int32_t i = 0; i < g_npeers; ++i) {
    g_nsend += g_send_counts[i];
    g_nrecv += g_recv_counts[i];
  }
  g_hsend = malloc(sizeof(double) * 6 * (size_t)(g_nsend ? g_nsend : 1));
  g_hrecv = malloc(sizeof(double) * 6 * (size_t)(g_nrecv ? g_nrecv : 1));
  RDyHipHalo halo = NULL;
  CHECK(rdyhip_halo_create(op, NULL, g_npeers, g_peers, g_send_counts, send_cells, g_recv_counts, recv_cells, &halo));
  CHECK(rdyhip_halo_set_transport(halo, pipe_transport, NULL));
  const int direct = rdyhip_halo_direct_receive(halo), overlaps = rdyhip_halo_overlaps(halo);

  /* ---- OperatorRHSFunctionHip: the global vector's rows into the local one, then the overlapped RHS ---- */
  double *u_glob = malloc(sizeof(double) * 3 * (size_t)(no ? no : 1)), *u_nan = malloc(sizeof(double) * 3 * (size_t)(nc ? nc : 1));
  for (int32_t k = 0; k < nc; ++k)
    for (int j = 0; j < 3; ++j) {
      u_nan[3 * (size_t)k + j] = NAN; /* neither owned nor ghost rows are known before the two copies */
      if (is_owned[k]) u_glob[3 * (size_t)c.mesh.cell_local_to_owned[k] + j] = c.u[3 * (size_t)k + j];
    }
  double *d_ug, *d_ul, *d_ul2, *d_f;
  HIPCHECK(hipMalloc((void **)&d_ug, sizeof(double) * 3 * (size_t)(no ? no : 1)));
  HIPCHECK(hipMalloc((void **)&d_ul, sizeof(double) * 3 * (size_t)(nc ? nc : 1)));
  HIPCHECK(hipMalloc((void **)&d_ul2, sizeof(double) * 3 * (size_t)(nc ? nc : 1)));
  HIPCHECK(hipMalloc((void **)&d_f, sizeof(double) * 3 * (size_t)(no ? no : 1)));
  HIPCHECK(hipMemcpy(d_ug, u_glob, sizeof(double) * 3 * (size_t)no, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_ul, u_nan, sizeof(double) * 3 * (size_t)nc, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_ul2, u_nan, sizeof(double) * 3 * (size_t)nc, hipMemcpyHostToDevice));
  const double dt = c.scal[3];
  for (int rep = 0; rep < 2; ++rep) { /* twice: buffers, events and streams are reused */
    CHECK(rdyhip_copy_owned_rows(op, d_ug, d_ul, st));
    CHECK(rdyhip_rhs_overlapped(op, halo, dt, d_ul, d_f, st));
  }
  HIPCHECK(hipStreamSynchronize(st));
  double *f = malloc(sizeof(double) * 3 * (size_t)(no ? no : 1)), *ul = malloc(sizeof(double) * 3 * (size_t)(nc ? nc : 1));
  HIPCHECK(hipMemcpy(f, d_f, sizeof(double) * 3 * (size_t)no, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(ul, d_ul, sizeof(double) * 3 * (size_t)nc, hipMemcpyDeviceToHost));
  double ef = 0.0, sf = 1.0;
  int    ghosts_ok = 1;
  for (size_t i = 0; i < 3 * (size_t)no; ++i) {
    ef = fmax(ef, fabs(f[i] - c.f_exp[i]));
    sf = fmax(sf, fabs(c.f_exp[i]));
  }
  for (size_t i = 0; i < 3 * (size_t)nc; ++i) ghosts_ok = ghosts_ok && ul[i] == c.u[i]; /* owned AND ghost rows, bit for bit */

  /* ---- the Courant struct-max over the ranks (MPI_Allreduce with MPI_MAX_COURANT_NUMBER), gathered on rank 0 ---- */
  RDyHipCourant cd;
  CHECK(rdyhip_update_diagnostics(op, st));
  CHECK(rdyhip_get_diagnostics(op, &cd));
  int courant_ok = 1;
  if (g_rank != 0) {
    if (put(0, &cd, sizeof(cd))) return 6;
  } else {
    for (int q = 1; q < g_world; ++q) {
      RDyHipCourant o;
      if (get(q, &o, sizeof(o))) return 6;
      if (o.max_courant_num > cd.max_courant_num) cd = o;
    }
    courant_ok = fabs(cd.max_courant_num - c.courant_exp) <= 1e-12 * fmax(1.0, c.courant_exp) && cd.global_cell_id == c.courant_cell_exp;
  }

  /* ---- three whole Euler steps with the pack riding on the kernels = the same steps as RHS + axpy on the host's side ---- */
  double es = 0.0;
  {
    const int fused = rdyhip_halo_fuse_pack(halo, 1) == 0 && rdyhip_halo_pack_fused(halo); /* second order keeps its pack launch */
    double   *a = d_ul, *b = d_ul2;
    for (int s = 0; s < 3; ++s) {
      CHECK(rdyhip_euler_step_overlapped(op, halo, 0.1 * dt, a, b, NULL, st));
      double *t = a; a = b; b = t;
    }
    HIPCHECK(hipStreamSynchronize(st));
    double *ua = malloc(sizeof(double) * 3 * (size_t)(nc ? nc : 1));
    HIPCHECK(hipMemcpy(ua, a, sizeof(double) * 3 * (size_t)nc, hipMemcpyDeviceToHost));
    /* reference: the same three steps as rdyhip_rhs_overlapped + rdyhip_axpy_owned, pack launches and all */
    CHECK(rdyhip_halo_fuse_pack(halo, 0));
    CHECK(rdyhip_copy_owned_rows(op, d_ug, d_ul, st));
    for (int s = 0; s < 3; ++s) {
      CHECK(rdyhip_rhs_overlapped(op, halo, 0.1 * dt, d_ul, d_f, st));
      CHECK(rdyhip_axpy_owned(op, 0.1 * dt, d_f, d_ul, st));
    }
    HIPCHECK(hipStreamSynchronize(st));
    HIPCHECK(hipMemcpy(ul, d_ul, sizeof(double) * 3 * (size_t)nc, hipMemcpyDeviceToHost));
    for (int32_t k = 0; k < nc; ++k)
      if (is_owned[k])
        for (int j = 0; j < 3; ++j) es = fmax(es, fabs(ua[3 * (size_t)k + j] - ul[3 * (size_t)k + j]));
    printf("rank %d: fused pack %d\n", g_rank, fused);
    free(ua);
  }
  printf("rank %d of %d: %d owned + %d ghost cells, %d peers, direct receive %d, overlapped form %d, rhs_linf %.3e, ghosts %s, euler steps %.3e%s\n",
         g_rank, g_world, no, nc - no, g_npeers, direct, overlaps, ef / sf, ghosts_ok ? "ok" : "WRONG", es,
         g_rank == 0 ? (courant_ok ? ", courant ok" : ", courant WRONG") : "");
  CHECK(rdyhip_halo_destroy(&halo));
  CHECK(rdyhip_halo_plan_destroy(&plan));
  CHECK(rdyhip_destroy(&op));
  return (ef / sf <= 1e-10 && ghosts_ok && courant_ok && es <= 1e-12) ? 0 : 5;
}

int main(int argc, char **argv) {
  if (argc < 3) return 1;
  g_world = atoi(argv[1]);
  if (g_world < 1 || g_world > MAXR) return 1;
  int fd[MAXR][MAXR][2]; /* fd[i][j]: i writes, j reads */
  for (int i = 0; i < g_world; ++i)
    for (int j = 0; j < g_world; ++j)
      if (i != j && pipe(fd[i][j])) return 1;
  pid_t pid[MAXR];
  for (int r = 0; r < g_world; ++r) {
    pid[r] = fork(); /* before any HIP call: every rank initialises the runtime itself */
    if (pid[r] < 0) return 1;
    if (pid[r] == 0) {
      g_rank = r;
      for (int i = 0; i < g_world; ++i)
        for (int j = 0; j < g_world; ++j) {
          if (i == j) continue;
          if (i == r) g_wr[j] = fd[i][j][1]; else close(fd[i][j][1]);
          if (j == r) g_rd[i] = fd[i][j][0]; else close(fd[i][j][0]);
        }
      const int rc = rank_main(argv[2]);
      fflush(stdout);
      _exit(rc);
    }
  }
  for (int i = 0; i < g_world; ++i)
    for (int j = 0; j < g_world; ++j)
      if (i != j) {
        close(fd[i][j][0]);
        close(fd[i][j][1]);
      }
  int worst = 0;
  for (int r = 0; r < g_world; ++r) {
    int status = 0;
    if (waitpid(pid[r], &status, 0) < 0 || !WIFEXITED(status)) worst = worst > 99 ? worst : 99;
    else if (WEXITSTATUS(status) > worst) worst = WEXITSTATUS(status);
  }
  return worst;
}
