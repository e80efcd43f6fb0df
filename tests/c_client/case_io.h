/* Reader of the case files tests/test_gpu_c_client.py writes (write_case): mesh arrays, boundaries, operator inputs,
 * state and expected results, all little-endian raw arrays.  Shared by the C clients of include/rdyhip.h. */
#ifndef RDYHIP_TEST_CASE_IO_H
#define RDYHIP_TEST_CASE_IO_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rdyhip.h"

typedef struct {
  int32_t         hdr[8]; /* num_cells, num_owned, num_edges, num_internal, num_boundaries, source_method, overwrite, well_balancing */
  double          scal[4]; /* tiny_h, h_anuga, xq2018_threshold, dt */
  RDyHipMesh      mesh;
  RDyHipBoundary *boundaries;
  double        **bvals; /* [boundary][edge][3] */
  double         *mannings, *extsrc /* [comp][owned] */, *u, *f_in, *f_exp, *pv_exp, courant_exp;
  /* optional trailer (multi-rank cases): the global cell id of the expected Courant maximum, the owner rank of every local cell,
   * the per-cell bed elevation of the hydrostatic reconstruction */
  int64_t  courant_cell_exp;
  int32_t *owner; /* [num_cells] or NULL */
  double  *zc;    /* [num_cells] or NULL */
} CaseFile;

static void *case_rd(FILE *f, size_t n, size_t sz) {
  void *p = malloc(n * sz > 0 ? n * sz : 1);
  if (n && fread(p, sz, n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(4);
  }
  return p;
}

/* 0 on success */
static int case_read(const char *path, CaseFile *c) {
  FILE *f = fopen(path, "rb");
  if (!f) return 1;
  if (fread(c->hdr, sizeof(int32_t), 8, f) != 8) return 4;
  if (fread(c->scal, sizeof(double), 4, f) != 4) return 4;
  const int32_t nc = c->hdr[0], no = c->hdr[1], ne = c->hdr[2], ni = c->hdr[3], nb = c->hdr[4];
  RDyHipMesh    m  = {0};
  m.num_cells = nc; m.num_owned_cells = no; m.num_edges = ne; m.num_internal_edges = ni;
  m.cell_is_owned       = case_rd(f, nc, 4);
  m.cell_local_to_owned = case_rd(f, nc, 4);
  m.cell_global_ids     = case_rd(f, nc, 8);
  m.cell_areas          = case_rd(f, nc, 8);
  m.cell_dz_dx          = case_rd(f, nc, 8);
  m.cell_dz_dy          = case_rd(f, nc, 8);
  m.edge_cell_ids       = case_rd(f, 2 * (size_t)ne, 4);
  m.edge_internal_ids   = case_rd(f, ni, 4);
  m.edge_global_ids     = case_rd(f, ne, 8);
  m.edge_lengths        = case_rd(f, ne, 8);
  m.edge_cn             = case_rd(f, ne, 8);
  m.edge_sn             = case_rd(f, ne, 8);
  c->mesh               = m;
  c->boundaries         = calloc(nb > 0 ? nb : 1, sizeof(RDyHipBoundary));
  c->bvals              = calloc(nb > 0 ? nb : 1, sizeof(double *));
  for (int i = 0; i < nb; ++i) {
    int32_t bh[2]; /* num_edges, condition type */
    if (fread(bh, 4, 2, f) != 2) return 4;
    c->boundaries[i].num_edges      = bh[0];
    c->boundaries[i].condition_type = bh[1];
    c->boundaries[i].edge_ids       = case_rd(f, bh[0], 4);
    c->bvals[i]                     = case_rd(f, 3 * (size_t)bh[0], 8);
  }
  c->mannings = case_rd(f, no, 8);
  c->extsrc   = case_rd(f, 3 * (size_t)no, 8);
  c->u        = case_rd(f, 3 * (size_t)nc, 8);
  c->f_in     = case_rd(f, 3 * (size_t)no, 8);
  c->f_exp    = case_rd(f, 3 * (size_t)no, 8);
  c->pv_exp   = case_rd(f, 3 * (size_t)no, 8);
  if (fread(&c->courant_exp, 8, 1, f) != 1) return 4;
  c->courant_cell_exp = -1;
  c->owner            = NULL;
  c->zc               = NULL;
  int32_t trailer[2]; /* has owners, has zc */
  if (fread(&c->courant_cell_exp, 8, 1, f) == 1 && fread(trailer, 4, 2, f) == 2) {
    if (trailer[0]) c->owner = case_rd(f, nc, 4);
    if (trailer[1]) c->zc = case_rd(f, nc, 8);
  }
  fclose(f);
  return 0;
}
#endif
