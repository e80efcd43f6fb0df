/*
 * TEST INFRASTRUCTURE -- see swe_oracle.h.  CPU restatement of RDycore's
 * first-order SWE right-hand side (PETSc backend).  The sweep structure of the
 * reference (gather -> velocities -> Roe -> accumulate; copy; source) is kept
 * so that timing this file is a fair stand-in for timing the reference's CPU
 * path, and every floating-point expression keeps the reference's operand
 * order so results agree to rounding.  Build with -ffp-contract=off (no FMA
 * fusion), as a default x86-64 gcc -O2 build of the reference would be.
 */
#include "swe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/swe/swe_types_petsc.h:7 */
static const double G = 9.806;

static double sq(double x) { return x * x; }

/* one side of a batch of Riemann problems (RiemannStateData, swe_types_petsc.h:14-18) */
typedef struct {
  int     n;
  double *h, *hu, *hv, *u, *v;
} Side;

/* per-edge data of a batch (RiemannEdgeData, swe_types_petsc.h:20-25) */
typedef struct {
  int     n;
  double *cn, *sn, *flux, *amax;
} Batch;

static void side_alloc(Side *s, int n) {
  s->n  = n;
  s->h  = calloc(n > 0 ? n : 1, sizeof(double));
  s->hu = calloc(n > 0 ? n : 1, sizeof(double));
  s->hv = calloc(n > 0 ? n : 1, sizeof(double));
  s->u  = calloc(n > 0 ? n : 1, sizeof(double));
  s->v  = calloc(n > 0 ? n : 1, sizeof(double));
}
static void side_free(Side *s) {
  free(s->h);
  free(s->hu);
  free(s->hv);
  free(s->u);
  free(s->v);
}
static void batch_alloc(Batch *b, int n) {
  b->n    = n;
  b->cn   = calloc(n > 0 ? n : 1, sizeof(double));
  b->sn   = calloc(n > 0 ? n : 1, sizeof(double));
  b->flux = calloc(n > 0 ? 3 * n : 1, sizeof(double));
  b->amax = calloc(n > 0 ? n : 1, sizeof(double));
}
static void batch_free(Batch *b) {
  free(b->cn);
  free(b->sn);
  free(b->flux);
  free(b->amax);
}

typedef struct {
  OracleBoundary desc;
  Side           left, right;
  Batch          edges;
  double        *values, *fluxes, *fluxes_accum; /* [num_edges][3] (src/operator.c:124-129) */
} BoundaryOp;

struct OracleOperator {
  OracleMesh    mesh;
  OracleConfig  config;
  OracleCourant courant;
  /* interior flux sub-operator (InteriorFluxOperator, src/swe/swe_petsc.c:79-94) */
  Side  left, right;
  Batch edges;
  /* boundary flux sub-operators */
  int         num_boundaries;
  BoundaryOp *boundaries;
  /* operator-owned vectors */
  double *external_sources, *material_properties, *flux_divergence, *primitive_variables;
  /* second order: the MUSCL members of InteriorFluxOperator (src/swe/swe_petsc.c:87-93) */
  int     num_owned_internal_edges, gradients_ready;
  double *ls_grad_coeffs;   /* [num_internal_edges][4] */
  double *grad[3];          /* grad_h, grad_hu, grad_hv: [num_cells][2] */
  double *q_reconstructed;  /* [num_owned_internal_edges][6] */
  double *rhs_local;        /* [num_cells][3] */
  Side    left2, right2;    /* Riemann batch of the owned internal edges */
  Batch   edges2;
  /* OpenMP build only (libswe_oracle_omp.so, the all-cores CPU line of bench.py): owned cell -> the positions e of its
   * internal edges in internal_edge_ids, ascending, so that a cell's contributions can be summed by ONE thread in the
   * serial loop's order (bitwise the serial result) */
  int *ce_off, *ce_idx;
};

/* ------------------------------------------------------------------------ */
/* ComputeRiemannVelocities, src/swe/swe_petsc.c:57-73                       */
static void velocities(double tiny_h, double h_anuga, Side *s) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int i = 0; i < s->n; ++i) {
    if (s->h[i] < tiny_h) {
      s->u[i] = 0.0;
      s->v[i] = 0.0;
    } else {
      double denom = sq(s->h[i]) + sq(h_anuga);
      s->u[i]      = s->hu[i] * s->h[i] / denom;
      s->v[i]      = s->hv[i] * s->h[i] / denom;
    }
  }
}

/* ComputeSWERoeEigenspectrum + one iteration of ComputeSWERoeFlux,
 * src/swe/swe_roe_flux_petsc.h:15-81, 103-128.  pow(x, 0.5), not sqrt, as there. */
void oracle_roe_flux(double hl, double ul, double vl, double hr, double ur, double vr, double sn, double cn, double fij[3], double *amax) {
  /* Roe averages (swe_roe_flux_petsc.h:21-29) */
  double sqhl  = pow(hl, 0.5);
  double sqhr  = pow(hr, 0.5);
  double cl    = pow(G * hl, 0.5);
  double cr    = pow(G * hr, 0.5);
  double hhat  = sqhl * sqhr;
  double uhat  = (sqhl * ul + sqhr * ur) / (sqhl + sqhr);
  double vhat  = (sqhl * vl + sqhr * vr) / (sqhl + sqhr);
  double chat  = pow(0.5 * G * (hl + hr), 0.5);
  double uperp = uhat * cn + vhat * sn;

  /* jumps (31-35) */
  double dh     = hr - hl;
  double du     = ur - ul;
  double dv     = vr - vl;
  double dupar  = -du * sn + dv * cn;
  double duperp = du * cn + dv * sn;

  /* right eigenvectors (38-46); R[0][*] = {1, 0, 1} */
  double r10 = uhat - chat * cn, r11 = -sn, r12 = uhat + chat * cn;
  double r20 = vhat - chat * sn, r21 = cn, r22 = vhat + chat * sn;

  /* |eigenvalues| with the critical-flow fix (49-67) */
  double uperpl = ul * cn + vl * sn;
  double uperpr = ur * cn + vr * sn;
  double a1     = fabs(uperp - chat);
  double a2     = fabs(uperp);
  double a3     = fabs(uperp + chat);
  double al1    = uperpl - cl;
  double ar1    = uperpr - cr;
  double da1    = fmax(0.0, 2.0 * (ar1 - al1));
  if (a1 < da1) a1 = 0.5 * (a1 * a1 / da1 + da1);
  double al3 = uperpl + cl;
  double ar3 = uperpr + cr;
  double da3 = fmax(0.0, 2.0 * (ar3 - al3));
  if (a3 < da3) a3 = 0.5 * (a3 * a3 / da3 + da3);

  /* characteristic jumps (73-75) */
  double dw0 = 0.5 * (dh - hhat * duperp / chat);
  double dw1 = hhat * dupar;
  double dw2 = 0.5 * (dh + hhat * duperp / chat);

  /* max wave speed (78) */
  *amax = chat + fabs(uperp);

  /* physical fluxes on both sides (111-122) */
  double fl0 = uperpl * hl;
  double fl1 = ul * uperpl * hl + 0.5 * G * hl * hl * cn;
  double fl2 = vl * uperpl * hl + 0.5 * G * hl * hl * sn;
  double fr0 = uperpr * hr;
  double fr1 = ur * uperpr * hr + 0.5 * G * hr * hr * cn;
  double fr2 = vr * uperpr * hr + 0.5 * G * hr * hr * sn;

  /* fij = 0.5 (FL + FR - R |Lambda| dW) (125-127) */
  fij[0] = 0.5 * (fl0 + fr0 - 1.0 * a1 * dw0 - 0.0 * a2 * dw1 - 1.0 * a3 * dw2);
  fij[1] = 0.5 * (fl1 + fr1 - r10 * a1 * dw0 - r11 * a2 * dw1 - r12 * a3 * dw2);
  fij[2] = 0.5 * (fl2 + fr2 - r20 * a1 * dw0 - r21 * a2 * dw1 - r22 * a3 * dw2);
}

static void roe_batch(const Side *l, const Side *r, Batch *b, double *flux_out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int i = 0; i < b->n; ++i) {
    oracle_roe_flux(l->h[i], l->u[i], l->v[i], r->h[i], r->u[i], r->v[i], b->sn[i], b->cn[i], &flux_out[3 * i], &b->amax[i]);
  }
}

/* ------------------------------------------------------------------------ */
/* ApplyInteriorFlux, src/swe/swe_petsc.c:215-316                            */
static void apply_interior_flux(OracleOperator *op, double dt, const double *u, double *f) {
  const OracleMesh *m      = &op->mesh;
  const double      tiny_h = op->config.tiny_h;
  Side             *L = &op->left, *R = &op->right;
  Batch            *E = &op->edges;

  /* gather (244-257) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int e = 0; e < m->num_internal_edges; ++e) {
    int edge = m->internal_edge_ids[e];
    int cl   = m->cell_ids[2 * edge];
    int cr   = m->cell_ids[2 * edge + 1];
    if (cr != -1) {
      L->h[e]  = u[3 * cl + 0];
      L->hu[e] = u[3 * cl + 1];
      L->hv[e] = u[3 * cl + 2];
      R->h[e]  = u[3 * cr + 0];
      R->hu[e] = u[3 * cr + 1];
      R->hv[e] = u[3 * cr + 2];
    }
  }
  /* velocities and Roe fluxes (259-270) */
  velocities(tiny_h, op->config.h_anuga_regular, L);
  velocities(tiny_h, op->config.h_anuga_regular, R);
  roe_batch(L, R, E, E->flux);

#ifdef _OPENMP
  /* the serial loop below split in two so that threads never add into the same row of f: (i) the Courant diagnostic,
   * one pass in edge order (first edge that reaches the maximum, as in the serial loop); (ii) per owned cell, its
   * edges' contributions in ascending edge order -- the order in which the serial loop adds them */
  for (int e = 0; e < m->num_internal_edges; ++e) {
    int edge = m->internal_edge_ids[e];
    int cl   = m->cell_ids[2 * edge];
    int cr   = m->cell_ids[2 * edge + 1];
    if (cr == -1 || (R->h[e] < tiny_h && L->h[e] < tiny_h)) continue;
    double areal = m->areas[cl], arear = m->areas[cr];
    double cnum  = E->amax[e] * m->lengths[edge] / fmin(areal, arear) * dt;
    if (cnum > op->courant.max_courant_num) {
      op->courant.max_courant_num = cnum;
      op->courant.global_edge_id  = m->edge_global_ids[edge];
      op->courant.global_cell_id  = (areal < arear) ? m->cell_global_ids[cl] : m->cell_global_ids[cr];
    }
  }
#pragma omp parallel for schedule(static)
  for (int c = 0; c < m->num_cells; ++c) {
    if (!m->is_owned[c]) continue;
    const int o = m->local_to_owned[c];
    for (int k = op->ce_off[o]; k < op->ce_off[o + 1]; ++k) {
      const int e = op->ce_idx[k], edge = m->internal_edge_ids[e];
      const int cl = m->cell_ids[2 * edge], cr = m->cell_ids[2 * edge + 1];
      if (R->h[e] < tiny_h && L->h[e] < tiny_h) continue;
      const double len = m->lengths[edge];
      const double w   = (c == cl) ? (-len / m->areas[cl]) : (len / m->areas[cr]);
      for (int q = 0; q < 3; ++q) f[3 * o + q] += E->flux[3 * e + q] * w;
    }
  }
  return;
#endif
  /* accumulate into owned cells + Courant diagnostic (275-310) */
  for (int e = 0; e < m->num_internal_edges; ++e) {
    int edge = m->internal_edge_ids[e];
    int cl   = m->cell_ids[2 * edge];
    int cr   = m->cell_ids[2 * edge + 1];
    if (cr == -1) continue;
    double len = m->lengths[edge];
    double hl  = L->h[e];
    double hr  = R->h[e];
    if (!(hr < tiny_h && hl < tiny_h)) {
      double areal = m->areas[cl];
      double arear = m->areas[cr];
      double cnum  = E->amax[e] * len / fmin(areal, arear) * dt;
      if (cnum > op->courant.max_courant_num) {
        op->courant.max_courant_num = cnum;
        op->courant.global_edge_id  = m->edge_global_ids[edge];
        op->courant.global_cell_id  = (areal < arear) ? m->cell_global_ids[cl] : m->cell_global_ids[cr];
      }
      for (int c = 0; c < 3; ++c) {
        if (m->is_owned[cl]) f[3 * m->local_to_owned[cl] + c] += E->flux[3 * e + c] * (-len / areal);
        if (m->is_owned[cr]) f[3 * m->local_to_owned[cr] + c] += E->flux[3 * e + c] * (len / arear);
      }
    }
  }
}


/* ApplyInteriorFluxHR, src/swe/swe_petsc.c:1000-1161: hydrostatic reconstruction
 * of the two depths of every interior edge, Roe flux on the reconstructed
 * states, and the hydrostatic pressure correction. */
static void apply_interior_flux_hr(OracleOperator *op, double dt, const double *u, double *f) {
  const OracleMesh *m      = &op->mesh;
  const double      tiny_h = op->config.tiny_h, h_anuga = op->config.h_anuga_regular;
  const double     *zc = m->zc;
  Side             *L = &op->left, *R = &op->right;
  Batch            *E = &op->edges;

  /* reconstruction (1031-1074) */
  for (int e = 0; e < m->num_internal_edges; ++e) {
    int edge = m->internal_edge_ids[e];
    int l    = m->cell_ids[2 * edge];
    int r    = m->cell_ids[2 * edge + 1];
    if (r == -1) continue;
    double h_L = u[3 * l + 0], hu_L = u[3 * l + 1], hv_L = u[3 * l + 2];
    double h_R = u[3 * r + 0], hu_R = u[3 * r + 1], hv_R = u[3 * r + 2];
    double zc_L = zc[l], zc_R = zc[r];
    double eta_L = h_L + zc_L, eta_R = h_R + zc_R;
    double z_max = fmax(zc_L, zc_R);
    double hL_rec = fmax(0.0, eta_L - z_max);
    double hR_rec = fmax(0.0, eta_R - z_max);
    double denom_L = sq(h_L) + sq(h_anuga);
    double denom_R = sq(h_R) + sq(h_anuga);
    L->h[e] = hL_rec;
    L->u[e] = (h_L > tiny_h) ? hu_L * h_L / denom_L : 0.0;
    L->v[e] = (h_L > tiny_h) ? hv_L * h_L / denom_L : 0.0;
    R->h[e] = hR_rec;
    R->u[e] = (h_R > tiny_h) ? hu_R * h_R / denom_R : 0.0;
    R->v[e] = (h_R > tiny_h) ? hv_R * h_R / denom_R : 0.0;
  }
  roe_batch(L, R, E, E->flux);

  /* accumulation + pressure correction (1085-1153) */
  for (int e = 0; e < m->num_internal_edges; ++e) {
    int edge = m->internal_edge_ids[e];
    int l    = m->cell_ids[2 * edge];
    int r    = m->cell_ids[2 * edge + 1];
    if (r == -1) continue;
    double h_L = u[3 * l + 0];
    double h_R = u[3 * r + 0];
    if (!(h_R < tiny_h && h_L < tiny_h)) {
      double len = m->lengths[edge];
      double areal = m->areas[l], arear = m->areas[r];
      double zc_L = zc[l], zc_R = zc[r];
      double eta_L = h_L + zc_L, eta_R = h_R + zc_R;
      double z_max = fmax(zc_L, zc_R);
      double hL_rec = fmax(0.0, eta_L - z_max);
      double hR_rec = fmax(0.0, eta_R - z_max);
      double scale_l = -len / areal;
      double scale_r = len / arear;
      if (hL_rec > tiny_h || hR_rec > tiny_h) {
        double cnum = E->amax[e] * len / fmin(areal, arear) * dt;
        if (cnum > op->courant.max_courant_num) {
          op->courant.max_courant_num = cnum;
          op->courant.global_edge_id  = m->edge_global_ids[edge];
          op->courant.global_cell_id  = (areal < arear) ? m->cell_global_ids[l] : m->cell_global_ids[r];
        }
        for (int c = 0; c < 3; ++c) {
          if (m->is_owned[l]) f[3 * m->local_to_owned[l] + c] += E->flux[3 * e + c] * scale_l;
          if (m->is_owned[r]) f[3 * m->local_to_owned[r] + c] += E->flux[3 * e + c] * scale_r;
        }
      }
      double corr_L = 0.5 * G * (sq(h_L) - sq(hL_rec));
      double corr_R = 0.5 * G * (sq(h_R) - sq(hR_rec));
      double cn = E->cn[e], sn = E->sn[e];
      if (m->is_owned[l]) {
        int lo = m->local_to_owned[l];
        f[3 * lo + 1] += corr_L * cn * scale_l;
        f[3 * lo + 2] += corr_L * sn * scale_l;
      }
      if (m->is_owned[r]) {
        int ro = m->local_to_owned[r];
        f[3 * ro + 1] += corr_R * cn * scale_r;
        f[3 * ro + 2] += corr_R * sn * scale_r;
      }
    }
  }
}

/* ApplyReflectingBC, src/swe/swe_petsc.c:434-461 */
static void reflecting_bc(const OracleMesh *m, BoundaryOp *b) {
  for (int e = 0; e < b->desc.num_edges; ++e) {
    int cl = m->cell_ids[2 * b->desc.edge_ids[e]];
    if (m->is_owned[cl]) {
      double sn = b->edges.sn[e], cn = b->edges.cn[e];
      b->right.h[e] = b->left.h[e];
      double dum1   = sq(sn) - sq(cn);
      double dum2   = 2.0 * sn * cn;
      b->right.u[e] = b->left.u[e] * dum1 - b->left.v[e] * dum2;
      b->right.v[e] = -b->left.u[e] * dum2 - b->left.v[e] * dum1;
    }
  }
}

/* ApplyCriticalOutflowBC, src/swe/swe_petsc.c:465-503 */
static void critical_outflow_bc(const OracleMesh *m, BoundaryOp *b) {
  for (int e = 0; e < b->desc.num_edges; ++e) {
    int cl = m->cell_ids[2 * b->desc.edge_ids[e]];
    if (m->is_owned[cl]) {
      double sn = b->edges.sn[e], cn = b->edges.cn[e];
      double uperp = b->left.u[e] * cn + b->left.v[e] * sn;
      if (uperp < 0.0) {
        /* inflow: both sides dry, so the edge is skipped by the both-dry guard */
        b->left.h[e] = b->left.u[e] = b->left.v[e] = 0.0;
        b->right.h[e] = b->right.u[e] = b->right.v[e] = 0.0;
      } else {
        double q      = b->left.h[e] * fabs(uperp);
        b->right.h[e] = pow(sq(q) / G, 1.0 / 3.0);
        double vel    = pow(G * b->right.h[e], 0.5);
        b->right.u[e] = vel * cn;
        b->right.v[e] = vel * sn;
      }
    }
  }
}

/* ApplyBoundaryFlux, src/swe/swe_petsc.c:506-630 */
static void apply_boundary_flux(OracleOperator *op, BoundaryOp *b, double dt, const double *u, double *f) {
  const OracleMesh *m      = &op->mesh;
  const double      tiny_h = op->config.tiny_h, h_anuga = op->config.h_anuga_regular;
  const int         n = b->desc.num_edges;

  /* left states (539-546) */
  for (int e = 0; e < n; ++e) {
    int cl        = m->cell_ids[2 * b->desc.edge_ids[e]];
    b->left.h[e]  = u[3 * cl + 0];
    b->left.hu[e] = u[3 * cl + 1];
    b->left.hv[e] = u[3 * cl + 2];
  }
  velocities(tiny_h, h_anuga, &b->left);

  /* right states from the boundary condition (549-569) */
  switch (b->desc.bc_type) {
    case ORACLE_BC_DIRICHLET:
      for (int e = 0; e < n; ++e) {
        b->right.h[e]  = b->values[3 * e + 0];
        b->right.hu[e] = b->values[3 * e + 1];
        b->right.hv[e] = b->values[3 * e + 2];
      }
      velocities(tiny_h, h_anuga, &b->right);
      break;
    case ORACLE_BC_REFLECTING: reflecting_bc(m, b); break;
    case ORACLE_BC_CRITICAL_OUTFLOW: critical_outflow_bc(m, b); break;
    default: break;
  }

  /* Riemann fluxes land directly in the boundary_fluxes vector (574) */
  roe_batch(&b->left, &b->right, &b->edges, b->fluxes);

  /* accumulate (583-607) */
  for (int e = 0; e < n; ++e) {
    int    edge = b->desc.edge_ids[e];
    double len  = m->lengths[edge];
    int    cl   = m->cell_ids[2 * edge];
    if (m->is_owned[cl]) {
      double area = m->areas[cl];
      double hl   = b->left.h[e];
      double hr   = b->right.h[e];
      if (!(hl < tiny_h && hr < tiny_h)) {
        double cnum = b->edges.amax[e] * len / area * dt;
        if (cnum > op->courant.max_courant_num) {
          op->courant.max_courant_num = cnum;
          op->courant.global_edge_id  = m->edge_global_ids[edge];
          op->courant.global_cell_id  = m->cell_global_ids[cl];
        }
        int o = m->local_to_owned[cl];
        for (int c = 0; c < 3; ++c) f[3 * o + c] += b->fluxes[3 * e + c] * (-len / area);
      }
    }
  }
  /* VecAXPY(boundary_fluxes_accum, dt, boundary_fluxes) (623) */
  for (int i = 0; i < 3 * n; ++i) b->fluxes_accum[i] += dt * b->fluxes[i];
}

/* ApplySourceSemiImplicit, src/swe/swe_petsc.c:704-804 */
static void apply_source_semi_implicit(OracleOperator *op, double dt, const double *u, double *f) {
  const OracleMesh *m      = &op->mesh;
  const double      tiny_h = op->config.tiny_h, h_anuga = op->config.h_anuga_regular;
  const double     *src = op->external_sources, *mat = op->material_properties, *fdiv = op->flux_divergence;
  double           *pv = op->primitive_variables;
  /* SourceOperator.include_bed_slope: false under HR, where the flux's pressure
   * correction carries the bed slope (src/swe/swe_petsc.c:700, 1243) */
  const int bed_slope = op->config.well_balancing != ORACLE_WB_HR;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int c = 0; c < m->num_cells; ++c) {
    if (!m->is_owned[c]) continue;
    int    o  = m->local_to_owned[c];
    double h  = u[3 * c + 0];
    double hu = u[3 * c + 1];
    double hv = u[3 * c + 2];

    double bedx = 0.0, bedy = 0.0;
    if (bed_slope) {
      bedx = m->dz_dx[c] * G * h;
      bedy = m->dz_dy[c] * G * h;
    }

    double Fsum_x = fdiv[3 * o + 1];
    double Fsum_y = fdiv[3 * o + 2];

    double tbx = 0.0, tby = 0.0;
    if (h >= tiny_h) {
      double uu = hu / h;
      double vv = hv / h;
      double n  = mat[o];
      /* Cd = g n^2 h^(-1/3) */
      double Cd     = G * sq(n) * pow(h, -1.0 / 3.0);
      double vel    = sqrt(sq(uu) + sq(vv));
      double tb     = Cd * vel / h;
      double factor = tb / (1.0 + dt * tb);
      tbx           = (hu + dt * Fsum_x - dt * bedx) * factor;
      tby           = (hv + dt * Fsum_y - dt * bedy) * factor;
    }
    f[3 * o + 0] += src[3 * o + 0];
    f[3 * o + 1] += -bedx - tbx + src[3 * o + 1];
    f[3 * o + 2] += -bedy - tby + src[3 * o + 2];

    double denom  = sq(h) + sq(h_anuga);
    pv[3 * o + 0] = h;
    pv[3 * o + 1] = (h >= tiny_h) ? (hu * h / denom) : 0.0;
    pv[3 * o + 2] = (h >= tiny_h) ? (hv * h / denom) : 0.0;
  }
}

/* ApplySourceImplicitXQ2018, src/swe/swe_petsc.c:816-932 */
static void apply_source_xq2018(OracleOperator *op, double dt, const double *u, double *f) {
  const OracleMesh *m      = &op->mesh;
  const double      tiny_h = op->config.tiny_h, h_anuga = op->config.h_anuga_regular;
  const double      thresh = op->config.xq2018_threshold;
  const double     *src = op->external_sources, *mat = op->material_properties, *fdiv = op->flux_divergence;
  double           *pv = op->primitive_variables;
  /* SourceOperator.include_bed_slope: false under HR, where the flux's pressure
   * correction carries the bed slope (src/swe/swe_petsc.c:700, 1243) */
  const int bed_slope = op->config.well_balancing != ORACLE_WB_HR;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int c = 0; c < m->num_cells; ++c) {
    if (!m->is_owned[c]) continue;
    int    o  = m->local_to_owned[c];
    double h  = u[3 * c + 0];
    double hu = u[3 * c + 1];
    double hv = u[3 * c + 2];

    double bedx = 0.0, bedy = 0.0;
    if (bed_slope) {
      bedx = m->dz_dx[c] * G * h;
      bedy = m->dz_dy[c] * G * h;
    }

    double tbx = 0.0, tby = 0.0;
    if (h >= tiny_h) {
      double n      = mat[o];
      double Fsum_x = fdiv[3 * o + 1];
      double Fsum_y = fdiv[3 * o + 2];
      double Ax     = Fsum_x - bedx;
      double Ay     = Fsum_y - bedy;
      double mx     = hu + Ax * dt;
      double my     = hv + Ay * dt;
      double lambda = G * sq(n) * pow(h, -4.0 / 3.0) * pow(sq(mx / h) + sq(my / h), 0.5);
      double qx, qy;
      if (dt * lambda < thresh) {
        qx = mx;
        qy = my;
      } else {
        qx = (mx - mx * pow(1.0 + 4.0 * dt * lambda, 0.5)) / (-2.0 * dt * lambda);
        qy = (my - my * pow(1.0 + 4.0 * dt * lambda, 0.5)) / (-2.0 * dt * lambda);
      }
      double qmag = pow(sq(qx) + sq(qy), 0.5);
      tbx         = G * sq(n) * pow(h, -7.0 / 3.0) * qx * qmag;
      tby         = G * sq(n) * pow(h, -7.0 / 3.0) * qy * qmag;
    }
    f[3 * o + 0] += src[3 * o + 0];
    f[3 * o + 1] += -bedx - tbx + src[3 * o + 1];
    f[3 * o + 2] += -bedy - tby + src[3 * o + 2];

    double denom  = sq(h) + sq(h_anuga);
    pv[3 * o + 0] = h;
    pv[3 * o + 1] = (h >= tiny_h) ? (hu * h / denom) : 0.0;
    pv[3 * o + 2] = (h >= tiny_h) ? (hv * h / denom) : 0.0;
  }
}


/* ------------------------------------------------------------------------ */
/* Second-order MUSCL path.                                                  */

static int edge_owned(const OracleMesh *m, int edge) { return m->edge_is_owned ? m->edge_is_owned[edge] : 1; }

/* PrecomputeLSGradCoeffs, src/operator_fluxes_ceed.c:884-980: inverse-distance weighted
 * least squares; per cell the 2x2 normal matrix M = sum w [dx dx, dx dy; dx dy, dy dy] over
 * its internal edges, per edge the two columns inv(M_cell) * w (dx, dy). */
static void precompute_ls_grad_coeffs(const OracleMesh *m, double *coeffs) {
  int     nc = m->num_cells;
  double *M  = calloc((size_t)(nc > 0 ? nc : 1) * 3, sizeof(double));
  for (int ie = 0; ie < m->num_internal_edges; ++ie) {
    int    e  = m->internal_edge_ids[ie];
    int    cl = m->cell_ids[2 * e], cr = m->cell_ids[2 * e + 1];
    double dx = m->centroids[3 * cr + 0] - m->centroids[3 * cl + 0];
    double dy = m->centroids[3 * cr + 1] - m->centroids[3 * cl + 1];
    double d  = sqrt(dx * dx + dy * dy);
    double w  = (d > 0.0) ? 1.0 / d : 0.0;
    M[cl * 3 + 0] += w * dx * dx;
    M[cl * 3 + 1] += w * dx * dy;
    M[cl * 3 + 2] += w * dy * dy;
    M[cr * 3 + 0] += w * dx * dx;
    M[cr * 3 + 1] += w * dx * dy;
    M[cr * 3 + 2] += w * dy * dy;
  }
  double *inv = calloc((size_t)(nc > 0 ? nc : 1) * 4, sizeof(double));
  for (int c = 0; c < nc; ++c) {
    double m00 = M[c * 3 + 0], m01 = M[c * 3 + 1], m11 = M[c * 3 + 2];
    double det = m00 * m11 - m01 * m01;
    if (fabs(det) < 1e-15) { /* degenerate stencil: zero gradient (926-933) */
      inv[c * 4 + 0] = inv[c * 4 + 1] = inv[c * 4 + 2] = inv[c * 4 + 3] = 0.0;
    } else {
      double inv_det = 1.0 / det;
      inv[c * 4 + 0] = m11 * inv_det;
      inv[c * 4 + 1] = -m01 * inv_det;
      inv[c * 4 + 2] = -m01 * inv_det;
      inv[c * 4 + 3] = m00 * inv_det;
    }
  }
  free(M);
  for (int ie = 0; ie < m->num_internal_edges; ++ie) {
    int    e  = m->internal_edge_ids[ie];
    int    cl = m->cell_ids[2 * e], cr = m->cell_ids[2 * e + 1];
    double dx = m->centroids[3 * cr + 0] - m->centroids[3 * cl + 0];
    double dy = m->centroids[3 * cr + 1] - m->centroids[3 * cl + 1];
    double d  = sqrt(dx * dx + dy * dy);
    double w  = (d > 0.0) ? 1.0 / d : 0.0;
    double wdx = w * dx, wdy = w * dy;
    coeffs[ie * 4 + 0] = inv[cl * 4 + 0] * wdx + inv[cl * 4 + 1] * wdy;
    coeffs[ie * 4 + 1] = inv[cl * 4 + 2] * wdx + inv[cl * 4 + 3] * wdy;
    coeffs[ie * 4 + 2] = inv[cr * 4 + 0] * wdx + inv[cr * 4 + 1] * wdy;
    coeffs[ie * 4 + 3] = inv[cr * 4 + 2] * wdx + inv[cr * 4 + 3] * wdy;
  }
  free(inv);
}

/* ComputeLeastSquaresGradients, src/operator_fluxes_ceed.c:998-1042 */
void oracle_compute_gradients(OracleOperator *op, const double *q) {
  const OracleMesh *m = &op->mesh;
  for (int k = 0; k < 3; ++k) memset(op->grad[k], 0, sizeof(double) * 2 * (size_t)m->num_cells);
  for (int ie = 0; ie < m->num_internal_edges; ++ie) {
    int           e  = m->internal_edge_ids[ie];
    int           cl = m->cell_ids[2 * e], cr = m->cell_ids[2 * e + 1];
    const double *c  = &op->ls_grad_coeffs[ie * 4];
    for (int k = 0; k < 3; ++k) {
      double dq = q[cr * 3 + k] - q[cl * 3 + k];
      op->grad[k][cl * 2 + 0] += c[0] * dq;
      op->grad[k][cl * 2 + 1] += c[1] * dq;
      op->grad[k][cr * 2 + 0] += c[2] * dq;
      op->grad[k][cr * 2 + 1] += c[3] * dq;
    }
  }
}

/* Minmod / VanLeer / LimitSlope, src/operator_fluxes_ceed.c:1110-1138 */
static double minmod(double a, double b) {
  if (a * b <= 0.0) return 0.0;
  return fabs(a) < fabs(b) ? a : b;
}
static double vanleer(double a, double b) {
  if (a * b <= 0.0) return 0.0;
  return 2.0 * a * b / (a + b);
}
static double limit_slope(int limiter, double extrap, double half_dq) {
  switch (limiter) {
    case ORACLE_LIMITER_NONE: return extrap;
    case ORACLE_LIMITER_VANLEER: return vanleer(extrap, half_dq);
    default: return minmod(extrap, half_dq);
  }
}

/* ReconstructFaceValues, src/operator_fluxes_ceed.c:1155-1206 */
static void reconstruct_face_values(OracleOperator *op, const double *q) {
  const OracleMesh *m       = &op->mesh;
  double           *q_face  = op->q_reconstructed;
  int               owned_e = 0;
  for (int ie = 0; ie < m->num_internal_edges; ++ie) {
    int e = m->internal_edge_ids[ie];
    if (!edge_owned(m, e)) continue;
    int    cl = m->cell_ids[2 * e], cr = m->cell_ids[2 * e + 1];
    int    v0 = m->vertex_ids[2 * e], v1 = m->vertex_ids[2 * e + 1];
    double x_mid = 0.5 * (m->points[3 * v0 + 0] + m->points[3 * v1 + 0]);
    double y_mid = 0.5 * (m->points[3 * v0 + 1] + m->points[3 * v1 + 1]);
    double dx_l = x_mid - m->centroids[3 * cl + 0], dy_l = y_mid - m->centroids[3 * cl + 1];
    double dx_r = x_mid - m->centroids[3 * cr + 0], dy_r = y_mid - m->centroids[3 * cr + 1];
    for (int k = 0; k < 3; ++k) {
      double extrap_l = op->grad[k][cl * 2 + 0] * dx_l + op->grad[k][cl * 2 + 1] * dy_l;
      double extrap_r = op->grad[k][cr * 2 + 0] * dx_r + op->grad[k][cr * 2 + 1] * dy_r;
      double dq       = q[cr * 3 + k] - q[cl * 3 + k];
      q_face[owned_e * 6 + k]     = q[cl * 3 + k] + limit_slope(op->config.limiter, extrap_l, 0.5 * dq);
      q_face[owned_e * 6 + 3 + k] = q[cr * 3 + k] + limit_slope(op->config.limiter, extrap_r, -0.5 * dq);
    }
    q_face[owned_e * 6 + 0] = fmax(0.0, q_face[owned_e * 6 + 0]);
    q_face[owned_e * 6 + 3] = fmax(0.0, q_face[owned_e * 6 + 3]);
    owned_e++;
  }
}

/* ApplyInteriorFlux2R, src/swe/swe_petsc.c:98-213 */
static void apply_interior_flux_2r(OracleOperator *op, double dt, const double *u, double *f) {
  const OracleMesh *m      = &op->mesh;
  const double      tiny_h = op->config.tiny_h;
  Side             *L = &op->left2, *R = &op->right2;
  Batch            *E = &op->edges2;

  if (!op->gradients_ready) oracle_compute_gradients(op, u); /* + CommunicateCellGradients: the identity on one rank */
  reconstruct_face_values(op, u);

  int owned_e = 0;
  for (int ie = 0; ie < m->num_internal_edges; ++ie) {
    if (!edge_owned(m, m->internal_edge_ids[ie])) continue;
    const double *qf = &op->q_reconstructed[owned_e * 6];
    L->h[owned_e]  = fmax(0.0, qf[0]);
    L->hu[owned_e] = qf[1];
    L->hv[owned_e] = qf[2];
    R->h[owned_e]  = fmax(0.0, qf[3]);
    R->hu[owned_e] = qf[4];
    R->hv[owned_e] = qf[5];
    owned_e++;
  }
  velocities(tiny_h, op->config.h_anuga_regular, L);
  velocities(tiny_h, op->config.h_anuga_regular, R);
  roe_batch(L, R, E, E->flux);

  double *rhs = op->rhs_local;
  memset(rhs, 0, sizeof(double) * 3 * (size_t)m->num_cells);
  owned_e = 0;
  for (int ie = 0; ie < m->num_internal_edges; ++ie) {
    int edge = m->internal_edge_ids[ie];
    if (!edge_owned(m, edge)) continue;
    int    cl = m->cell_ids[2 * edge], cr = m->cell_ids[2 * edge + 1];
    double len = m->lengths[edge];
    double hl = L->h[owned_e], hr = R->h[owned_e];
    if (!(hr < tiny_h && hl < tiny_h)) {
      double areal = m->areas[cl], arear = m->areas[cr];
      double cnum  = E->amax[owned_e] * len / fmin(areal, arear) * dt;
      if (cnum > op->courant.max_courant_num) {
        op->courant.max_courant_num = cnum;
        op->courant.global_edge_id  = m->edge_global_ids[edge];
        op->courant.global_cell_id  = (areal < arear) ? m->cell_global_ids[cl] : m->cell_global_ids[cr];
      }
      for (int c = 0; c < 3; ++c) {
        rhs[3 * cl + c] += E->flux[3 * owned_e + c] * (-len / areal);
        rhs[3 * cr + c] += E->flux[3 * owned_e + c] * (len / arear);
      }
    }
    owned_e++;
  }
  /* DMLocalToGlobal(ADD_VALUES): the owned rows here; the ghost rows stay in rhs_local for the harness */
  for (int c = 0; c < m->num_cells; ++c) {
    if (!m->is_owned[c]) continue;
    for (int k = 0; k < 3; ++k) f[3 * m->local_to_owned[c] + k] += rhs[3 * c + k];
  }
}

void    oracle_set_gradients_ready(OracleOperator *op, int ready) { op->gradients_ready = ready; }
double *oracle_gradients(OracleOperator *op, int k) { return op->grad[k]; }
double *oracle_rhs_local(OracleOperator *op) { return op->rhs_local; }
double *oracle_ls_grad_coeffs(OracleOperator *op) { return op->ls_grad_coeffs; }

/* ------------------------------------------------------------------------ */
OracleOperator *oracle_create(const OracleMesh *mesh, const OracleConfig *config, int num_boundaries, const OracleBoundary *boundaries) {
  OracleOperator *op = calloc(1, sizeof(*op));
  op->mesh           = *mesh;
  op->config         = *config;
  oracle_reset_diagnostics(op);

  /* CreatePetscSWEInteriorFluxOperator, src/swe/swe_petsc.c:341-408 */
  int ni = mesh->num_internal_edges;
  side_alloc(&op->left, ni);
  side_alloc(&op->right, ni);
  batch_alloc(&op->edges, ni);
  for (int e = 0; e < ni; ++e) {
    int edge = mesh->internal_edge_ids[e];
    if (mesh->cell_ids[2 * edge + 1] != -1) {
      op->edges.cn[e] = mesh->cn[edge];
      op->edges.sn[e] = mesh->sn[edge];
    }
  }

  /* the second_order branch of CreatePetscSWEInteriorFluxOperator (357-403): owned-edge batch, LS coefficients */
  if (config->second_order) {
    int nown = 0;
    for (int e = 0; e < ni; ++e) nown += edge_owned(mesh, mesh->internal_edge_ids[e]);
    op->num_owned_internal_edges = nown;
    side_alloc(&op->left2, nown);
    side_alloc(&op->right2, nown);
    batch_alloc(&op->edges2, nown);
    int owned_e = 0;
    for (int e = 0; e < ni; ++e) {
      int edge = mesh->internal_edge_ids[e];
      if (!edge_owned(mesh, edge)) continue;
      op->edges2.cn[owned_e] = mesh->cn[edge];
      op->edges2.sn[owned_e] = mesh->sn[edge];
      owned_e++;
    }
    size_t nc          = mesh->num_cells > 0 ? mesh->num_cells : 1;
    op->ls_grad_coeffs = calloc((size_t)(ni > 0 ? ni : 1) * 4, sizeof(double));
    for (int k = 0; k < 3; ++k) op->grad[k] = calloc(nc * 2, sizeof(double));
    op->q_reconstructed = calloc((size_t)(nown > 0 ? nown : 1) * 6, sizeof(double));
    op->rhs_local       = calloc(nc * 3, sizeof(double));
    precompute_ls_grad_coeffs(mesh, op->ls_grad_coeffs);
  }

  /* CreatePetscSWEBoundaryFluxOperator, src/swe/swe_petsc.c:653-687 */
  op->num_boundaries = num_boundaries;
  op->boundaries     = calloc(num_boundaries > 0 ? num_boundaries : 1, sizeof(BoundaryOp));
  for (int b = 0; b < num_boundaries; ++b) {
    BoundaryOp *bo = &op->boundaries[b];
    bo->desc       = boundaries[b];
    int n          = boundaries[b].num_edges;
    side_alloc(&bo->left, n);
    side_alloc(&bo->right, n);
    batch_alloc(&bo->edges, n);
    bo->values       = calloc(n > 0 ? 3 * n : 1, sizeof(double));
    bo->fluxes       = calloc(n > 0 ? 3 * n : 1, sizeof(double));
    bo->fluxes_accum = calloc(n > 0 ? 3 * n : 1, sizeof(double));
    for (int e = 0; e < n; ++e) {
      bo->edges.cn[e] = mesh->cn[boundaries[b].edge_ids[e]];
      bo->edges.sn[e] = mesh->sn[boundaries[b].edge_ids[e]];
    }
  }

#ifdef _OPENMP
  {
    const int nown = mesh->num_owned_cells;
    op->ce_off     = calloc((size_t)nown + 2, sizeof(int));
    for (int e = 0; e < ni; ++e) {
      int edge = mesh->internal_edge_ids[e], cl = mesh->cell_ids[2 * edge], cr = mesh->cell_ids[2 * edge + 1];
      if (cr == -1) continue;
      if (mesh->is_owned[cl]) op->ce_off[mesh->local_to_owned[cl] + 1]++;
      if (mesh->is_owned[cr]) op->ce_off[mesh->local_to_owned[cr] + 1]++;
    }
    for (int o = 0; o < nown; ++o) op->ce_off[o + 1] += op->ce_off[o];
    op->ce_idx = calloc((size_t)(op->ce_off[nown] > 0 ? op->ce_off[nown] : 1), sizeof(int));
    int *fill  = calloc((size_t)nown + 1, sizeof(int));
    for (int e = 0; e < ni; ++e) {  /* ascending e: each cell's list comes out in the serial loop's order */
      int edge = mesh->internal_edge_ids[e], cl = mesh->cell_ids[2 * edge], cr = mesh->cell_ids[2 * edge + 1];
      if (cr == -1) continue;
      if (mesh->is_owned[cl]) { int o = mesh->local_to_owned[cl]; op->ce_idx[op->ce_off[o] + fill[o]++] = e; }
      if (mesh->is_owned[cr]) { int o = mesh->local_to_owned[cr]; op->ce_idx[op->ce_off[o] + fill[o]++] = e; }
    }
    free(fill);
  }
#endif
  int no                  = mesh->num_owned_cells > 0 ? mesh->num_owned_cells : 1;
  op->external_sources    = calloc(3 * no, sizeof(double));
  op->material_properties = calloc(no, sizeof(double));
  op->flux_divergence     = calloc(3 * no, sizeof(double));
  op->primitive_variables = calloc(3 * no, sizeof(double));
  return op;
}

void oracle_destroy(OracleOperator *op) {
  if (!op) return;
  free(op->ce_off);
  free(op->ce_idx);
  side_free(&op->left);
  side_free(&op->right);
  batch_free(&op->edges);
  for (int b = 0; b < op->num_boundaries; ++b) {
    BoundaryOp *bo = &op->boundaries[b];
    side_free(&bo->left);
    side_free(&bo->right);
    batch_free(&bo->edges);
    free(bo->values);
    free(bo->fluxes);
    free(bo->fluxes_accum);
  }
  if (op->config.second_order) {
    side_free(&op->left2);
    side_free(&op->right2);
    batch_free(&op->edges2);
    free(op->ls_grad_coeffs);
    for (int k = 0; k < 3; ++k) free(op->grad[k]);
    free(op->q_reconstructed);
    free(op->rhs_local);
  }
  free(op->boundaries);
  free(op->external_sources);
  free(op->material_properties);
  free(op->flux_divergence);
  free(op->primitive_variables);
  free(op);
}

/* ApplyPetscOperator, src/operator.c:656-672: flux composite (interior, then
 * one sub-operator per boundary, src/operator_fluxes_petsc.c:17-53), copy of
 * f into flux_divergence, source composite. */
int oracle_apply_interior(OracleOperator *op, double dt, const double *u_local, double *f_global) {
  /* CreatePetscFluxOperator vs CreatePetscFluxHROperator (src/operator.c:186-200, src/operator_fluxes_petsc.c:17-95) */
  if (op->config.second_order && op->config.well_balancing == ORACLE_WB_HR) return 1; /* rejected, src/operator.c:388-389 */
  if (op->config.well_balancing == ORACLE_WB_HR) apply_interior_flux_hr(op, dt, u_local, f_global);
  else if (op->config.second_order) apply_interior_flux_2r(op, dt, u_local, f_global); /* src/swe/swe_petsc.c:403-404 */
  else apply_interior_flux(op, dt, u_local, f_global);
  return 0;
}

int oracle_apply_rest(OracleOperator *op, double dt, const double *u_local, double *f_global) {
  for (int b = 0; b < op->num_boundaries; ++b) apply_boundary_flux(op, &op->boundaries[b], dt, u_local, f_global);
  memcpy(op->flux_divergence, f_global, sizeof(double) * 3 * (size_t)op->mesh.num_owned_cells);
  switch (op->config.source_method) {
    case ORACLE_SOURCE_SEMI_IMPLICIT: apply_source_semi_implicit(op, dt, u_local, f_global); break;
    case ORACLE_SOURCE_IMPLICIT_XQ2018: apply_source_xq2018(op, dt, u_local, f_global); break;
    default: return 1;
  }
  return 0;
}

int oracle_apply(OracleOperator *op, double dt, const double *u_local, double *f_global) {
  int rc = oracle_apply_interior(op, dt, u_local, f_global);
  if (rc) return rc;
  return oracle_apply_rest(op, dt, u_local, f_global);
}

double *oracle_boundary_values(OracleOperator *op, int b) { return op->boundaries[b].values; }
double *oracle_boundary_fluxes(OracleOperator *op, int b) { return op->boundaries[b].fluxes; }
double *oracle_boundary_fluxes_accum(OracleOperator *op, int b) { return op->boundaries[b].fluxes_accum; }
double *oracle_external_sources(OracleOperator *op) { return op->external_sources; }
double *oracle_material_properties(OracleOperator *op) { return op->material_properties; }
double *oracle_flux_divergence(OracleOperator *op) { return op->flux_divergence; }
double *oracle_primitive_variables(OracleOperator *op) { return op->primitive_variables; }

/* OpenMP build: the number of threads (the environment variable is read once per process by libgomp, which a host
 * such as torch has usually initialised already); a no-op in the serial build */
#ifdef _OPENMP
#include <omp.h>
int oracle_set_num_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}
#else
int oracle_set_num_threads(int n) {
  (void)n;
  return 1;
}
#endif

void oracle_reset_diagnostics(OracleOperator *op) {
  op->courant.max_courant_num = 0.0;
  op->courant.global_edge_id  = -1;
  op->courant.global_cell_id  = -1;
}
void oracle_get_diagnostics(OracleOperator *op, OracleCourant *out) { *out = op->courant; }
