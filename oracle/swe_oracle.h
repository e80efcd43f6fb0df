/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement, in plain C, of the arithmetic RDycore's PETSc backend runs
 * for one shallow-water right-hand-side evaluation.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Pinning: the reference's own build (PETSc + libCEED) is absent from this
 * image, so the reference sources cannot be compiled here; this restatement is
 * pinned by (1) the Roe-flux known-answer vectors recorded from the reference
 * arithmetic in SURVEY.md section 8.a and (2) the reference's only accuracy
 * gate for this path, the MMS convergence-rate thresholds of
 * driver/tests/swe_roe/mms_conv_study.yaml:48-64 (tests/test_oracle_pins.py).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the RDycore source tree).
 */
#ifndef SWE_ORACLE_H
#define SWE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* RDyConditionType, include/rdycore.h:133-139 */
enum { ORACLE_BC_DIRICHLET = 0, ORACLE_BC_REFLECTING = 2, ORACLE_BC_CRITICAL_OUTFLOW = 3 };
/* RDyFlowSourceMethod, include/private/rdyconfigimpl.h:52-56 */
enum { ORACLE_SOURCE_SEMI_IMPLICIT = 0, ORACLE_SOURCE_IMPLICIT_XQ2018 = 1 };
enum { ORACLE_WB_NONE = 0, ORACLE_WB_HR = 2 };
enum { ORACLE_LIMITER_MINMOD = 0, ORACLE_LIMITER_NONE = 1, ORACLE_LIMITER_VANLEER = 2 };

/* the RDyMesh fields the operators read (include/private/rdymeshimpl.h) */
typedef struct {
  int num_cells, num_owned_cells, num_edges, num_internal_edges;
  const int       *is_owned;          /* cells.is_owned        [num_cells] */
  const int       *local_to_owned;    /* cells.local_to_owned  [num_cells] */
  const long long *cell_global_ids;   /* cells.global_ids      [num_cells] */
  const double    *areas;             /* cells.areas           [num_cells] */
  const double    *dz_dx, *dz_dy;     /* cells.dz_dx/dz_dy     [num_cells] */
  const double    *zc;                /* vertex-averaged bed elevation per cell (HR only; src/swe/swe_petsc.c:1209-1224) */
  const int       *cell_ids;          /* edges.cell_ids        [2*num_edges] */
  const int       *internal_edge_ids; /* edges.internal_edge_ids [num_internal_edges] */
  const long long *edge_global_ids;   /* edges.global_ids      [num_edges] */
  const double    *lengths, *cn, *sn; /* edges.lengths/cn/sn   [num_edges] */
  /* second-order (MUSCL) reconstruction only; may be NULL otherwise */
  int              num_vertices;
  const double    *centroids;         /* cells.centroids[c].X  [num_cells][3] */
  const int       *vertex_ids;        /* edges.vertex_ids      [2*num_edges] */
  const double    *points;            /* vertices.points[v].X  [num_vertices][3] */
  const int       *edge_is_owned;     /* edges.is_owned        [num_edges] (src/rdymesh.c:599) */
} OracleMesh;

typedef struct {
  int        num_edges;
  const int *edge_ids; /* RDyBoundary.edge_ids */
  int        bc_type;  /* RDyCondition.flow->type */
} OracleBoundary;

typedef struct {
  double tiny_h, h_anuga_regular, xq2018_threshold; /* RDyPhysicsFlow, rdyconfigimpl.h:74-85 */
  int    source_method;
  int    well_balancing; /* RDyWellBalanceMethod: 0 none, 2 hydrostatic reconstruction (rdyconfigimpl.h:58-62) */
  int    second_order;   /* RDyNumericsSection.second_order (rdyconfigimpl.h:129): MUSCL reconstruction, ApplyInteriorFlux2R */
  int    limiter;        /* RDyLimiterType (rdyconfigimpl.h:67-71): 0 minmod, 1 none, 2 van Leer */
} OracleConfig;

/* CourantNumberDiagnostics, include/private/rdyoperatorimpl.h:21-25 */
typedef struct {
  double    max_courant_num;
  long long global_edge_id, global_cell_id;
} OracleCourant;

typedef struct OracleOperator OracleOperator;

/* CreateOperator (src/operator.c:348-417): allocates the per-sub-operator
 * scratch and the operator-owned vectors.  The mesh arrays are borrowed. */
OracleOperator *oracle_create(const OracleMesh *mesh, const OracleConfig *config, int num_boundaries, const OracleBoundary *boundaries);
void            oracle_destroy(OracleOperator *op);

/* ApplyPetscOperator (src/operator.c:656-672): f_global += flux divergence + sources. */
int oracle_apply(OracleOperator *op, double dt, const double *u_local, double *f_global);
/* the same in two steps: the interior-flux sub-operator, then the boundary-flux sub-operators and
 * the source operator -- so that a multi-rank harness can add the second-order path's ghost rows
 * (DMLocalToGlobal ADD_VALUES inside ApplyInteriorFlux2R, src/swe/swe_petsc.c:207) in between */
int oracle_apply_interior(OracleOperator *op, double dt, const double *u_local, double *f_global);
int oracle_apply_rest(OracleOperator *op, double dt, const double *u_local, double *f_global);

/* operator-owned vectors (src/operator.c:91-129, 224-335) */
double *oracle_boundary_values(OracleOperator *op, int b);       /* [num_edges][3] */
double *oracle_boundary_fluxes(OracleOperator *op, int b);       /* [num_edges][3] */
double *oracle_boundary_fluxes_accum(OracleOperator *op, int b); /* [num_edges][3] */
double *oracle_external_sources(OracleOperator *op);             /* [owned][3] */
double *oracle_material_properties(OracleOperator *op);          /* [owned][1] Manning n */
double *oracle_flux_divergence(OracleOperator *op);              /* [owned][3] */
double *oracle_primitive_variables(OracleOperator *op);          /* [owned][3] */

/* ---- second order (config.second_order): ApplyInteriorFlux2R, src/swe/swe_petsc.c:98-213 ----
 * oracle_apply then runs ComputeLeastSquaresGradients -> ReconstructFaceValues -> Roe on the
 * OWNED internal edges, accumulates into a local vector (ghost rows included) and adds its
 * owned rows to f_global.  Across ranks the reference exchanges the gradients
 * (CommunicateCellGradients) and adds the ghost rows onto their owners (DMLocalToGlobal
 * ADD_VALUES); a multi-rank test harness does both by hand between these calls:
 *   oracle_compute_gradients(op, u)   ComputeLeastSquaresGradients into oracle_gradients()
 *   [overwrite ghost rows of oracle_gradients(op,k) with the owners' values]
 *   oracle_set_gradients_ready(op, 1) the next oracle_apply uses the gradients as they are
 *   oracle_apply(...)                 then add the ghost rows of oracle_rhs_local() to their owners */
void    oracle_compute_gradients(OracleOperator *op, const double *u_local);
void    oracle_set_gradients_ready(OracleOperator *op, int ready);
double *oracle_gradients(OracleOperator *op, int k); /* k = 0,1,2: grad_h, grad_hu, grad_hv, each [num_cells][2] */
double *oracle_rhs_local(OracleOperator *op);        /* [num_cells][3] interior-flux contributions of the last apply */
double *oracle_ls_grad_coeffs(OracleOperator *op);   /* [num_internal_edges][4] (PrecomputeLSGradCoeffs) */

void oracle_reset_diagnostics(OracleOperator *op); /* ResetOperatorDiagnostics, src/operator.c:772-784 */
int  oracle_set_num_threads(int n); /* OpenMP build only: threads of the parallel loops; returns the count in effect */
void oracle_get_diagnostics(OracleOperator *op, OracleCourant *out);

/* ComputeSWERoeFlux for one edge (src/swe/swe_roe_flux_petsc.h:91-132) */
void oracle_roe_flux(double hl, double ul, double vl, double hr, double ur, double vr, double sn, double cn, double fij[3], double *amax);

#ifdef __cplusplus
}
#endif
#endif
