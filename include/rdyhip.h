/*
 * rdyhip.h -- C ABI of the MI355X-native shallow-water RHS operator.
 *
 * This library replaces ONE path of RDycore: the operator that PETSc's
 * explicit TS calls for every right-hand-side evaluation (first-order Roe
 * fluxes over edges + bed-slope / Manning friction / external sources over
 * cells).  Every entry point below names the RDycore interface it stands in
 * for (file:line relative to the RDycore source tree); INTEGRATION.md shows
 * the adapter a maintainer would add on the RDycore side.
 *
 * Conventions
 *  - plain C, no PETSc / torch types: pointers, sizes, an opaque handle;
 *  - every function returns 0 on success (PETSC_SUCCESS) or one of the
 *    RDYHIP_ERR_* codes, whose values are PETSc's (petscerror.h) so an adapter
 *    can pass them straight to PetscCall(); rdyhip_last_error() gives the text;
 *  - "device" pointers are HIP device memory on the device that was current at
 *    rdyhip_create(); "host" pointers are ordinary memory;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *    the apply calls enqueue work and return without synchronising;
 *  - indices are 32-bit (a PetscInt=int32 build; a 64-bit-PetscInt adapter
 *    narrows local indices, which always fit), global ids are 64-bit;
 *  - one operator per rank, one rank per GPU, no threads (as in the reference).
 */
#ifndef RDYHIP_H
#define RDYHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDYHIP_VERSION 110

/* error codes = PETSc's values */
#define RDYHIP_SUCCESS 0
#define RDYHIP_ERR_MEM 55      /* PETSC_ERR_MEM */
#define RDYHIP_ERR_ARG_SIZ 60  /* PETSC_ERR_ARG_SIZ */
#define RDYHIP_ERR_ARG_OUTOFRANGE 63
#define RDYHIP_ERR_LIB 76      /* PETSC_ERR_LIB: a HIP runtime call failed */
#define RDYHIP_ERR_USER 83     /* PETSC_ERR_USER */

/* RDyConditionType (include/rdycore.h:133-139) */
#define RDYHIP_CONDITION_DIRICHLET 0
#define RDYHIP_CONDITION_REFLECTING 2
#define RDYHIP_CONDITION_CRITICAL_OUTFLOW 3

/* RDyFlowSourceMethod (include/private/rdyconfigimpl.h:52-56) */
#define RDYHIP_SOURCE_SEMI_IMPLICIT 0
#define RDYHIP_SOURCE_IMPLICIT_XQ2018 1

/* RDyWellBalanceMethod (include/private/rdyconfigimpl.h:58-62).  BS2002 exists only in
 * the reference's CEED backend (src/operator.c:388) and is rejected here too. */
#define RDYHIP_WELL_BALANCING_NONE 0
#define RDYHIP_WELL_BALANCING_HR 2

/* RDyLimiterType (include/private/rdyconfigimpl.h:64-71): slope limiter of the second-order reconstruction */
#define RDYHIP_LIMITER_MINMOD 0
#define RDYHIP_LIMITER_NONE 1
#define RDYHIP_LIMITER_VANLEER 2

/* RDyNumericsRiemann (include/private/rdyconfigimpl.h:118-122); only Roe exists
 * in the reference (src/swe/swe_petsc.c:264-270) */
#define RDYHIP_RIEMANN_ROE 0

/* The scalars of RDyConfig that reach the kernels
 * (config.physics.flow.{tiny_h,h_anuga_regular,source.method,source.xq2018_threshold,well_balancing},
 *  config.numerics.riemann; include/private/rdyconfigimpl.h:73-85,125-132). */
typedef struct {
  double  tiny_h;
  double  h_anuga_regular;
  double  xq2018_threshold;
  int32_t source_method; /* RDYHIP_SOURCE_* */
  int32_t riemann;       /* RDYHIP_RIEMANN_ROE */
  int32_t well_balancing; /* RDYHIP_WELL_BALANCING_*: HR = ApplyInteriorFluxHR + bed-slope-free source
                             (src/swe/swe_petsc.c:1000-1263); the mesh must then carry cell_zc and, as the
                             reference does (RDyMeshOverride2DProjection, src/rdymesh.c:1478-1509), x-y
                             projected edge lengths and cell areas */
  int32_t second_order;  /* config.numerics.second_order (rdyconfigimpl.h:129): MUSCL reconstruction of the interior
                            face states, ApplyInteriorFlux2R (src/swe/swe_petsc.c:98-213); the mesh must then carry
                            cell_centroids, edge_vertex_ids and vertex_points; not combinable with HR (src/operator.c:388-389) */
  int32_t limiter;       /* RDYHIP_LIMITER_* (config.numerics.limiter after the -no_limiter / -van_leer overrides,
                            src/swe/swe_petsc.c:357-367); read only if second_order */
  int32_t flags;         /* RDYHIP_CONFIG_* bits; 0 = defaults */
} RDyHipConfig;

/* RDyHipConfig.flags.
 * CACHED_F_STORES: the tiled kernels store F with the default cache policy instead of the non-temporal hint.  For a host that
 *   reads F straight back in a separate kernel -- PETSc's TSEULER: VecAXPY(U, dt, F) after every RHS (TSStep_Euler) -- the
 *   hint costs ~6 % of the RHS + axpy pair, because F has then left the Infinity Cache (DESIGN.md section 7 note 6); a host
 *   that takes the step through rdyhip_euler_step (F never stored) or lets F sit (RK stages summed later) leaves it clear.
 *   Honoured by the first-order and HR tiled kernels; the second-order kernels keep the hint. */
#define RDYHIP_CONFIG_CACHED_F_STORES 1

/* The RDyMesh arrays the SWE operators read (include/private/rdymeshimpl.h:26-202).
 * All host pointers, borrowed for the duration of rdyhip_create() only. */
typedef struct {
  int32_t num_cells;          /* mesh->num_cells (owned + ghost) */
  int32_t num_owned_cells;    /* mesh->num_owned_cells */
  int32_t num_edges;          /* mesh->num_edges */
  int32_t num_internal_edges; /* mesh->num_internal_edges */
  const int32_t *cell_is_owned;       /* cells.is_owned       [num_cells] */
  const int32_t *cell_local_to_owned; /* cells.local_to_owned [num_cells] */
  const int64_t *cell_global_ids;     /* cells.global_ids     [num_cells] */
  const double  *cell_areas;          /* cells.areas          [num_cells] */
  const double  *cell_dz_dx;          /* cells.dz_dx          [num_cells] */
  const double  *cell_dz_dy;          /* cells.dz_dy          [num_cells] */
  const int32_t *edge_cell_ids;       /* edges.cell_ids       [2*num_edges], right = -1 on the boundary */
  const int32_t *edge_internal_ids;   /* edges.internal_edge_ids [num_internal_edges] */
  const int64_t *edge_global_ids;     /* edges.global_ids     [num_edges] */
  const double  *edge_lengths;        /* edges.lengths        [num_edges] */
  const double  *edge_cn;             /* edges.cn             [num_edges] */
  const double  *edge_sn;             /* edges.sn             [num_edges] */
  const double  *cell_zc;             /* vertex-averaged bed elevation per cell [num_cells] (InteriorFluxHROperator.zc,
                                         src/swe/swe_petsc.c:1209-1224); may be NULL unless well_balancing == HR */
  /* second_order only (may be NULL / 0 otherwise): what PrecomputeLSGradCoeffs and ReconstructFaceValues read
   * (src/operator_fluxes_ceed.c:884-980, 1155-1206) */
  int32_t        num_vertices;        /* mesh->num_vertices */
  const double  *cell_centroids;      /* cells.centroids      [num_cells][3]    (RDyPoint.X) */
  const int32_t *edge_vertex_ids;     /* edges.vertex_ids     [2*num_edges] */
  const double  *vertex_points;       /* vertices.points      [num_vertices][3] (RDyPoint.X) */
  const int32_t *edge_is_owned;       /* edges.is_owned       [num_edges] (PetscBool; src/rdymesh.c:571-599).  Read by second_order
                                         only: ApplyInteriorFlux2R evaluates the Courant number of an edge on the rank that owns
                                         it (src/swe/swe_petsc.c:172-190); NULL: every edge of an owned cell counts (one rank) */
} RDyHipMesh;

/* RDyBoundary (include/private/rdyboundaryimpl.h:7-14) + the flow condition
 * type of the RDyCondition attached to it (include/private/rdyconditionimpl.h:12-24). */
typedef struct {
  int32_t        num_edges;
  const int32_t *edge_ids; /* local edge ids, host, borrowed during create */
  int32_t        condition_type; /* RDYHIP_CONDITION_* */
} RDyHipBoundary;

/* CourantNumberDiagnostics (include/private/rdyoperatorimpl.h:21-25) */
typedef struct {
  double  max_courant_num;
  int64_t global_edge_id;
  int64_t global_cell_id;
} RDyHipCourant;

/* opaque Operator (include/private/rdyoperatorimpl.h:103-201) */
typedef struct RDyHipOperator_s *RDyHipOperator;

/* device-resident operator fields that may be read (or, for inputs, written)
 * in place; see rdyhip_field_ptr() */
typedef enum {
  RDYHIP_FIELD_PRIMITIVE_VARIABLES = 0, /* Operator.primitive_variables [owned][3] (h,u,v); out */
  RDYHIP_FIELD_EXTERNAL_SOURCES    = 1, /* Operator.petsc.external_sources [owned][3]; in; also src_inst */
  RDYHIP_FIELD_MANNINGS            = 2, /* Operator.petsc.material_properties [owned][1]; in */
  RDYHIP_FIELD_FLUX_DIVERGENCE     = 3, /* Operator.flux_divergence [owned][3]; out, only if enabled */
  RDYHIP_FIELD_GRADIENTS           = 4, /* second order: InteriorFluxOperator.grad_h/grad_hu/grad_hv interleaved,
                                           [num_cells][6] = (dh/dx, dh/dy, dhu/dx, dhu/dy, dhv/dx, dhv/dy), LOCAL cell index:
                                           owned rows are written by rdyhip_compute_gradients, ghost rows by the caller's
                                           halo exchange (CommunicateCellGradients, src/operator_fluxes_ceed.c:1058-1107) */
} RDyHipField;

/* which cells a partial apply covers (multi-GPU overlap, see rdyhip_apply_phase) */
#define RDYHIP_PHASE_ALL 0
#define RDYHIP_PHASE_INTERIOR 1 /* owned cells with no ghost neighbour */
#define RDYHIP_PHASE_HALO 2     /* owned cells with at least one ghost neighbour */

const char *rdyhip_last_error(void);
int32_t     rdyhip_version(void);

/* ---- lifecycle -------------------------------------------------------------
 * CreateOperator(RDyConfig*, DM, RDyMesh*, num_comp, num_regions, RDyRegion*,
 *                num_boundaries, RDyBoundary*, RDyCondition*, Operator**)
 *   include/private/rdyoperatorimpl.h:208, src/operator.c:348-417;
 * together with CreatePetscSWEInteriorFluxOperator / ...BoundaryFluxOperator /
 * ...SourceOperator (include/private/rdysweimpl.h:37-42, src/swe/swe_petsc.c:341,653,948).
 * num_comp is fixed at 3 (SWE); regions only matter to the setters below.
 * Repacks the mesh into the device layout and allocates the operator-owned
 * vectors (boundary values/fluxes/accum, external sources, Manning n,
 * primitive variables), all zero-initialised as in src/operator.c:91-129. */
int rdyhip_create(const RDyHipConfig *config, const RDyHipMesh *mesh, int32_t num_boundaries, const RDyHipBoundary *boundaries,
                  RDyHipOperator *op);

/* DestroyOperator(Operator**)  include/private/rdyoperatorimpl.h:210, src/operator.c:421-493 */
int rdyhip_destroy(RDyHipOperator *op);

/* ---- the hot path -----------------------------------------------------------
 * ApplyOperator(Operator*, PetscReal dt, Vec u_local, Vec f_global)
 *   include/private/rdyoperatorimpl.h:213, src/operator.c:680-690 (-> ApplyPetscOperator 656-672).
 * u_local: device, [num_cells][3] (h,hu,hv), ghosts filled by the caller.
 * f_global: device, [num_owned_cells][3]; the operator ADDS into it exactly as
 * the reference does, and the friction term sees the incoming content through
 * the flux-divergence sum (src/operator.c:663). */
int rdyhip_apply(RDyHipOperator op, double dt, const double *u_local, double *f_global, void *stream);

/* OperatorRHSFunction's "VecZeroEntries(F); ResetOperatorDiagnostics; ApplyOperator"
 *   src/rdysetup.c:1130,1136,1139 -- fused: f_global is overwritten (never read),
 * diagnostics are reset on the stream first.  The halo update of u_local
 * (rdysetup.c:1133-1134) stays with the caller. */
int rdyhip_rhs_function(RDyHipOperator op, double dt, const double *u_local, double *f_global, void *stream);

/* The same as rdyhip_rhs_function restricted to a subset of the owned cells, so
 * that a caller can overlap the halo exchange with the interior cells:
 *   apply_phase(INTERIOR, RDYHIP_PHASE_OVERWRITE | RDYHIP_PHASE_RESET_DIAGNOSTICS) || exchange;
 *   apply_phase(HALO, RDYHIP_PHASE_OVERWRITE).
 * `flags`: RDYHIP_PHASE_OVERWRITE gives rdyhip_rhs_function semantics (f = F(u)),
 * without it rdyhip_apply's (f += F(u)); RDYHIP_PHASE_RESET_DIAGNOSTICS resets the
 * Courant diagnostic first (no extra launch). */
#define RDYHIP_PHASE_OVERWRITE 1
#define RDYHIP_PHASE_RESET_DIAGNOSTICS 2
#define RDYHIP_PHASE_GRADIENTS_READY 4 /* second order: use the gradient field as it is (the caller has run
                                          rdyhip_compute_gradients and filled the ghost rows) */
int rdyhip_apply_phase(RDyHipOperator op, int32_t phase, int32_t flags, double dt, const double *u_local, double *f_global,
                       void *stream);

/* ---- second order (config.second_order) -------------------------------------
 * rdyhip_apply / rdyhip_rhs_function then run ApplyInteriorFlux2R (src/swe/swe_petsc.c:98-213):
 * least-squares gradients of the owned cells, limited reconstruction of both face states of every
 * interior edge, Roe flux on the reconstructed states; boundary edges and sources are unchanged.
 * On one rank nothing else is needed.  With ghost cells the gradients of the ghosts must come from
 * their owners (CommunicateCellGradients) before the fluxes are taken:
 *   [halo update of u_local]
 *   rdyhip_compute_gradients(op, RDYHIP_PHASE_ALL, u_local, stream)     ComputeLeastSquaresGradients, owned cells
 *   [halo update of the RDYHIP_FIELD_GRADIENTS rows, 6 values per cell: rdyhip_pack_rows / rdyhip_unpack_rows]
 *   rdyhip_apply_phase(op, RDYHIP_PHASE_ALL, flags | RDYHIP_PHASE_GRADIENTS_READY, ...)
 * Every rank then evaluates all edges of its owned cells itself (a cut edge on both ranks, from
 * identical operands), so the reference's reverse DMLocalToGlobal(ADD_VALUES) has no counterpart.
 * PHASE_INTERIOR / PHASE_HALO select owned cells without / with a ghost neighbour, for overlap.  With the default
 * fused kernel (RDyHipLayoutInfo.second_order_fused) the gradients of interior cells never leave the chip: only
 * rdyhip_compute_gradients(RDYHIP_PHASE_HALO) is needed to feed the exchange, and the INTERIOR tiles of
 * rdyhip_apply_phase need no data from other ranks at all. */
int rdyhip_compute_gradients(RDyHipOperator op, int32_t phase, const double *u_local, void *stream);

/* ---- operator data (host-side setters, as in the reference) -----------------
 * SetOperatorBoundaryValues(Operator*, RDyBoundary, comp_offset, num_comp, num_edges, values[num_comp*e+c])
 *   include/private/rdyoperatorimpl.h:254, src/operator.c:1045-1061.  `values` is a host pointer. */
int rdyhip_set_boundary_values(RDyHipOperator op, int32_t boundary, int32_t comp_offset, int32_t num_comp, int32_t num_edges,
                               const double *values);

/* ExtractOperatorBoundaryFluxes (include/private/rdyoperatorimpl.h:256): copies
 * boundary_fluxes[b] or boundary_fluxes_accum[b] ([num_edges][3], src/operator.c:124-129) to the host. */
int rdyhip_get_boundary_fluxes(RDyHipOperator op, int32_t boundary, int32_t accumulated, int32_t num_edges, double *fluxes);
int rdyhip_reset_boundary_fluxes_accum(RDyHipOperator op);

/* Get/RestoreOperator{Regional,Domain}ExternalSource (include/private/rdyoperatorimpl.h:258-261,
 *   src/operator.c:1203-1241,1394-1429) as used by RDySet{Regional,Domain}{Water,XMomentum,YMomentum}Source
 *   (src/rdydata.c:225-366): sets component `comp` of the external source of
 *   `n` owned cells; owned_cell_ids == NULL means owned cells 0..n-1 (domain). Host pointers. */
int rdyhip_set_external_source(RDyHipOperator op, int32_t comp, int32_t n, const int32_t *owned_cell_ids, const double *values);

/* Get/RestoreOperator{Regional,Domain}MaterialProperties (rdyoperatorimpl.h:263-266)
 *   as used by RDySet{Regional,Domain}ManningsN (src/rdydata.c:506-539). */
int rdyhip_set_mannings(RDyHipOperator op, int32_t n, const int32_t *owned_cell_ids, const double *values);

/* The same three setters ordered on a stream instead of synchronising the device: launches enqueued on `stream` before the
 * call see the old values, launches enqueued on it afterwards the new ones (applies running on OTHER streams are the
 * caller's to order).  `values` / `owned_cell_ids` are host arrays and may be reused as soon as the call returns: they are
 * copied into pinned staging memory of the operator, travel to the device on the operator's own copy stream beside whatever
 * `stream` is executing, and only the last step -- a scatter launch or a device-to-device copy -- is ordered on `stream`.
 * Arrays above 8 MB go in 8-MB chunks, the upload of a chunk beside the host copies of the chunks behind it: with an adaptive
 * time step the caller has just read the Courant struct back, the device is idle and waits for exactly this.
 * Nothing blocks and nothing drains the device: with a fixed time step RDyAdvance needs no synchronisation at all
 * (src/rdyadvance.c:303-305), and a drained device runs its next ~40 launches 20-30 % slow (profiles/r02_launch_series.json).
 * rdyhip_refresh_field replaces a whole input field (RDYHIP_FIELD_EXTERNAL_SOURCES [owned][3], RDYHIP_FIELD_MANNINGS [owned])
 * from a host or a device array -- what an adapter does when a PETSc Vec of the Operator changed (src/operator.c:91-96). */
int rdyhip_set_boundary_values_on(RDyHipOperator op, int32_t boundary, int32_t comp_offset, int32_t num_comp, int32_t num_edges,
                                  const double *values, void *stream);
int rdyhip_set_external_source_on(RDyHipOperator op, int32_t comp, int32_t n, const int32_t *owned_cell_ids, const double *values, void *stream);
int rdyhip_set_mannings_on(RDyHipOperator op, int32_t n, const int32_t *owned_cell_ids, const double *values, void *stream);

/* ---- forcing ingestion on the device -----------------------------------------
 * The per-step fill loops of RDyApplyForcing (src/forcing/rdyforcing.c:688-770)
 * with the dataset, the data->mesh map and the region's cell list resident in
 * HBM: nothing crosses PCIe between RHS evaluations.  All pointers are DEVICE
 * pointers; the calls are enqueued on `stream` and do not synchronise.
 * `d_owned_cell_ids` (the region's owned-cell indices, RDyRegion.owned_cell_ids;
 * NULL = owned cells 0..n-1) must hold valid indices: they cannot be range-checked
 * on the host.  The scalar time lookup (RDyForcingGetCurrentData,
 * src/forcing/rdyforcing_dataset.c:32-67) stays on the host (rdycore_amd/forcing.py).
 *
 * fill_source:     RDyForcingSetConstantRainfall / RDyForcingSetHomogeneousData (rdyforcing_dataset.c:282-288,
 *                  320-344) + RDySetRegionalWaterSource / RDySetHomogeneousRegionalWaterSource (src/rdydata.c:253-309):
 *                  ext[id[i]][comp] = value
 * gather_source:   RDyForcingSetRasterData (rdyforcing_dataset.c:295-314: stride 1, offset = header_offset,
 *                  scale = 1/(1000*3600)) and RDyForcingSetUnstructuredData (350-373: offset 2, scale 1):
 *                  ext[id[i]][comp] = data[data2mesh_idx[i]*stride + offset] * scale
 * fill_boundary:   RDyForcingSetHomogeneousBoundary (380-406) + RDySetFlowDirichletBoundaryValues: bvalues[e] = [h,0,0]
 * gather_boundary: RDyForcingSetUnstructuredData on a boundary dataset (stride 3):
 *                  bvalues[e][c] = data[data2mesh_idx[e]*stride + c + offset]
 * nearest_map:     RDyForcingCreateRasterDatasetMapping (src/forcing/rdyforcing_map.c:111-141; min_dist0 =
 *                  (max(ncols,nrows)+1)*cellsize, entries with no point nearer than that are left untouched) and
 *                  RDyForcingCreateUnstructuredDatasetMap (77-104; pass min_dist0 < 0): brute-force nearest
 *                  data point, first index wins ties. */
int rdyhip_forcing_fill_source(RDyHipOperator op, int32_t comp, int32_t n, const int32_t *d_owned_cell_ids, double value, void *stream);
int rdyhip_forcing_gather_source(RDyHipOperator op, int32_t comp, int32_t n, const int32_t *d_owned_cell_ids, const double *d_data,
                                 const int32_t *d_data2mesh_idx, int64_t stride, int64_t offset, double scale, void *stream);
int rdyhip_forcing_fill_boundary(RDyHipOperator op, int32_t boundary, int32_t num_edges, double h, void *stream);
int rdyhip_forcing_gather_boundary(RDyHipOperator op, int32_t boundary, int32_t num_edges, const double *d_data,
                                   const int32_t *d_data2mesh_idx, int64_t stride, int64_t offset, void *stream);
int rdyhip_forcing_nearest_map(int32_t n, const double *d_xc, const double *d_yc, int32_t ndata, const double *d_data_xc,
                               const double *d_data_yc, double min_dist0, int32_t *d_map, void *stream);

/* In-place access to the device-resident fields (no copy): e.g. a forcing
 * kernel can write the external source on the GPU, an output routine can read
 * primitive_variables (read by src/rdyadvance.c's averaging monitors). */
int rdyhip_field_ptr(RDyHipOperator op, RDyHipField field, double **device_ptr, int64_t *num_values);
int rdyhip_refresh_field(RDyHipOperator op, RDyHipField field, const double *values, int64_t num_values, int32_t values_on_device, void *stream);
/* keep Operator.flux_divergence (one extra [owned][3] store per apply); off by default */
int rdyhip_enable_flux_divergence(RDyHipOperator op, int32_t enable);

/* ---- diagnostics ------------------------------------------------------------
 * ResetOperatorDiagnostics / UpdateOperatorDiagnostics / GetOperatorDiagnostics
 *   include/private/rdyoperatorimpl.h:269-271, src/operator.c:772-784,867-893.
 * reset is enqueued on `stream`; update synchronises `stream`, copies the
 * 16-byte result to the host and resolves the edge/cell ids.  The cross-rank
 * MPI_Allreduce of src/operator.c:879 stays with the caller (one struct). */
int rdyhip_reset_diagnostics(RDyHipOperator op, void *stream);
int rdyhip_update_diagnostics(RDyHipOperator op, void *stream);
int rdyhip_get_diagnostics(RDyHipOperator op, RDyHipCourant *courant);

/* ---- halo helpers (DMGlobalToLocalBegin/End, src/rdysetup.c:1133-1134) -------
 * pack:   buf[i][0..2] = u_local[cell_ids[i]][0..2]   (cells a neighbour rank needs)
 * unpack: u_local[cell_ids[i]][0..2] = buf[i][0..2]   (this rank's ghost cells)
 * all device pointers; the transport between ranks (RCCL) is the caller's. */
int rdyhip_pack_cells(const double *u_local, const int32_t *cell_ids, int32_t n, double *buf, void *stream);
int rdyhip_unpack_cells(double *u_local, const int32_t *cell_ids, int32_t n, const double *buf, void *stream);
/* the same for rows of `ncomp` values (the [num_cells][6] gradient field): buf[i][:] = src[ids[i]][:] and back */
int rdyhip_pack_rows(const double *src, int32_t ncomp, const int32_t *row_ids, int32_t n, double *buf, void *stream);
int rdyhip_unpack_rows(double *dst, int32_t ncomp, const int32_t *row_ids, int32_t n, const double *buf, void *stream);

/* ---- the ghost update and its overlap with the interior cells, behind the ABI ----
 * OperatorRHSFunction's DMGlobalToLocalBegin/End (src/rdysetup.c:1133-1134) for a C host: the operator packs the owned
 * cells its neighbours need, exchanges them and unpacks into its ghost cells on an internal stream, while
 * the cells without ghost neighbours are evaluated on the caller's stream; the ghost-adjacent cells follow.
 *
 *   rdyhip_halo_create   the exchange pattern of this rank: for each of `npeers` neighbour ranks, the LOCAL ids of the
 *                        owned cells it needs from this rank (send lists) and of this rank's ghost cells it owns (receive
 *                        lists), in an order both sides agree on (e.g. ascending global id), concatenated in peer order.
 *                        `nccl_comm`: an RCCL communicator (ncclComm_t passed as void*) spanning the operator's ranks, or
 *                        NULL if the bytes travel through rdyhip_halo_set_transport.  With RCCL one exchange is
 *                        ncclGroupStart(); ncclSend / ncclRecv per peer; ncclGroupEnd() over xGMI -- point-to-point, no collective.
 *   rdyhip_halo_set_transport  replaces RCCL by the caller's transport (GPU-aware MPI_Isend/Irecv in an MPI code, a
 *                        host-staged exchange in the single-GPU tests): called once per exchange, after the send buffer has
 *                        been packed on `stream`, it must deliver d_send's per-peer slices into the peers' d_recv slices
 *                        ([cells][ncomp] doubles, peer order as at create) so that work enqueued on `stream` afterwards sees
 *                        d_recv filled; it may block the host (the interior launch is already enqueued by then).
 *   rdyhip_halo_exchange the plain ghost update of a [num_cells][ncomp] array (ncomp = 3: the solution; 6: the
 *                        second-order gradients, CommunicateCellGradients), all on `stream`
 *   rdyhip_rhs_overlapped        = halo update of u_local + rdyhip_rhs_function, overlapped (the whole OperatorRHSFunction)
 *   rdyhip_euler_step_overlapped = halo update of u_local + rdyhip_euler_step(PHASE_ALL), overlapped
 * The tiles that need ghost data follow the unpack on the library's stream, beside the tail of the other tiles.
 * Second order: the state exchange hides behind the tiles that need no ghost data, then the ghost-adjacent gradients are
 * computed, exchanged (6 values per cell) and the remaining tiles follow.  The exchanges are ordered after everything
 * already enqueued on `stream`; when the call returns all work is enqueued and later work on `stream` is ordered after it.
 * The form of a step -- everything in order on `stream`: exchange, (gradients, their exchange,) ONE launch over all tiles; or
 * on two streams as described above -- is chosen per halo and per kind of step by a TRIAL on the communicator the halo really
 * has: the first 16 calls alternate between the forms, each timed on the device, and the faster one stays (both forms post
 * the same send / receive group, so every rank chooses for itself).  A small part wants the first form (its interior tiles
 * run for less time than the exchange chain, and the two cross-stream dependencies cost more than they hide), a large one
 * behind a slow link the second.  rdyhip_halo_form_info() reports the choice and the two timings, rdyhip_halo_set_form()
 * forces a form or restarts the trial; RDYHIP_OVERLAP=0 / 1 and RDYHIP_OVERLAP_MIN_ROUNDS=n (the size rule of earlier
 * versions), read once at rdyhip_halo_create, force as well.  rdyhip_halo_overlaps() = the RHS step currently runs on two streams.
 *
 * Two ways to shorten the exchange chain of a small part (a 0.36 M-cell rank's kernel runs 17 us; a pack and an unpack launch
 * cost 3.5 us each):
 *   direct receive   when the ghost cells this rank receives are one run of consecutive local rows in arrival order -- peer
 *                    by peer, a peer's cells in the agreed order, which is how rdyhip_local_cell_order numbers them -- the
 *                    transfer lands in the caller's array itself (ncclRecv into u_local's ghost rows): no receive buffer, no
 *                    unpack launch.  Detected at rdyhip_halo_create; rdyhip_halo_direct_receive() says whether it applies.
 *   fused pack       rdyhip_halo_fuse_pack(halo, 1): the Euler-step kernels (first order, HR and fused second order) also store the new state of
 *                    the cells other ranks need into the send buffer as they store u_local_out, so that the NEXT
 *                    rdyhip_euler_step_overlapped, called with that array as its u_local, starts with the transfer: no pack
 *                    launch.  The promise the caller makes: between two such steps it does not write the owned rows of that
 *                    array itself -- or calls rdyhip_halo_invalidate() if it did (a host that sets the state, a restart).  A
 *                    step whose u_local is any other array packs as before.  One halo per operator can hold the fused pack.
 *                    Second order (fused form): the gradient launch over the ghost-adjacent cells then also stores each gradient
 *                    into the send rows it travels in, so the gradient exchange of rdyhip_rhs_overlapped /
 *                    rdyhip_euler_step_overlapped needs no pack launch either (RDYHIP_GRAD_PACK_FUSED=0: measurement knob).
 * With both, a step of rdyhip_euler_step_overlapped is the transfer and ONE kernel launch.
 * (A fused pack whose send lists hold a cell in a tile no ghost touches -- possible when the DM's overlap is vertex-adjacent,
 * src/rdydm.c:150 -- is only safe when nothing runs beside the transfer: such a halo keeps its Euler steps in order,
 * rdyhip_halo_form_info says RDYHIP_HALO_FORM_LOCKED_IN_ORDER.) */
typedef struct RDyHipHalo_s *RDyHipHalo;
typedef int (*RDyHipTransportFn)(void *ctx, const double *d_send, double *d_recv, int32_t ncomp, void *stream);
int rdyhip_halo_create(RDyHipOperator op, void *nccl_comm, int32_t npeers, const int32_t *peers, const int32_t *send_counts,
                       const int32_t *send_cell_ids, const int32_t *recv_counts, const int32_t *recv_cell_ids, RDyHipHalo *halo);
int rdyhip_halo_destroy(RDyHipHalo *halo);
int32_t rdyhip_halo_overlaps(RDyHipHalo halo);
int32_t rdyhip_halo_direct_receive(RDyHipHalo halo);
int rdyhip_halo_fuse_pack(RDyHipHalo halo, int32_t enable);
int32_t rdyhip_halo_pack_fused(RDyHipHalo halo);
/* which form the steps of a halo take and why (see above); kind: RDYHIP_HALO_STEP_RHS = rdyhip_rhs_overlapped,
 * RDYHIP_HALO_STEP_EULER = rdyhip_euler_step_overlapped */
#define RDYHIP_HALO_STEP_RHS 0
#define RDYHIP_HALO_STEP_EULER 1
#define RDYHIP_HALO_FORM_TRIAL_RUNNING 0   /* the first calls still alternate */
#define RDYHIP_HALO_FORM_MEASURED 1        /* the faster form of the trial */
#define RDYHIP_HALO_FORM_FORCED 2          /* environment or rdyhip_halo_set_form */
#define RDYHIP_HALO_FORM_DEFAULT 3         /* nothing to choose (no peers) or no timing available: in order */
#define RDYHIP_HALO_FORM_LOCKED_IN_ORDER 4 /* fused pack with a send cell outside the ghost-adjacent tiles */
typedef struct {
  int32_t form;           /* 0: in order on the caller's stream, 1: two streams */
  int32_t source;         /* RDYHIP_HALO_FORM_* */
  int32_t trial_steps;    /* calls the trial has used so far (16 when done) */
  double  in_order_ms;    /* mean device time per step of each form in the trial (0 until it is over) */
  double  two_stream_ms;
} RDyHipHaloFormInfo;
int rdyhip_halo_form_info(RDyHipHalo halo, int32_t kind, RDyHipHaloFormInfo *info);
int rdyhip_halo_set_form(RDyHipHalo halo, int32_t kind, int32_t form /* 0, 1, or < 0: run the trial again */);
int rdyhip_halo_invalidate(RDyHipHalo halo);
int rdyhip_halo_set_transport(RDyHipHalo halo, RDyHipTransportFn fn, void *ctx);
int rdyhip_halo_exchange(RDyHipHalo halo, double *rows, int32_t ncomp, void *stream);
int rdyhip_rhs_overlapped(RDyHipOperator op, RDyHipHalo halo, double dt, double *u_local, double *f_global, void *stream);
int rdyhip_euler_step_overlapped(RDyHipOperator op, RDyHipHalo halo, double dt, double *u_local, double *u_local_out, double *f_global,
                                 void *stream);
/* RCCL communicator helpers, so that a host needs no RCCL binding of its own: rank 0 calls rdyhip_comm_unique_id and
 * broadcasts the 128 bytes (MPI_Bcast in RDycore), then every rank calls rdyhip_comm_init_rank (ncclCommInitRank on the
 * current device; collective over the ranks). */
#define RDYHIP_COMM_ID_BYTES 128
int rdyhip_comm_unique_id(char id[RDYHIP_COMM_ID_BYTES]);
int rdyhip_comm_init_rank(int32_t nranks, int32_t rank, const char id[RDYHIP_COMM_ID_BYTES], void **nccl_comm);
int rdyhip_comm_destroy(void *nccl_comm);
int rdyhip_comm_count(void *nccl_comm, int32_t *nranks);   /* ncclCommCount: how many ranks the communicator really spans */
int32_t rdyhip_rccl_version(void);

/* ---- planning the exchange and the local numbering (host only: no device is touched, testable without a GPU) ----
 * What the DM's point SF knows (PetscSFGetGraph on DMGetPointSF(dm): for every ghost cell its owner rank and the owner's
 * index for it; the 1-cell overlap of src/rdydm.c:145-157) turned into rdyhip_halo_create's arguments.  The send side is the
 * transpose of the receive side, which needs ONE all-to-all; the library leaves that to the caller's own communicator:
 *
 *   rdyhip_halo_plan_create(world, rank, num_ghosts, ghost_cell_ids, ghost_owner_ranks, ghost_keys, &plan)
 *        ghost_cell_ids[i]    local id of my i-th ghost cell
 *        ghost_owner_ranks[i] the rank that owns it (iremote[i].rank)
 *        ghost_keys[i]        the OWNER's name for it: its local cell id there (iremote[i].index - cStart), or a global
 *                             cell id -- whichever rdyhip_halo_plan_finish is told the owner's cells are keyed by
 *   rdyhip_halo_plan_requests(plan, &counts, &keys)
 *        counts[r] = number of cells I need from rank r; keys = their keys grouped by r, ascending inside a group (the
 *        order both sides use).  Caller: MPI_Alltoall(counts -> incoming_counts), MPI_Alltoallv(keys -> incoming_keys).
 *   rdyhip_halo_plan_finish(plan, incoming_counts, incoming_keys, num_cells, cell_is_owned, cell_keys)
 *        resolves what the others ask of me to my local cells: cell_keys == NULL: a key IS my local cell id; else
 *        cell_keys[c] is the key of local cell c (e.g. cells.global_ids).  Every request must name a cell I own.
 *   rdyhip_halo_plan_get(plan, &npeers, &peers, &send_counts, &send_cell_ids, &recv_counts, &recv_cell_ids)
 *        exactly the arguments of rdyhip_halo_create (pointers valid until rdyhip_halo_plan_destroy).
 *
 *   rdyhip_hilbert_cell_order(num_cells, xy, stride, cell_is_owned, perm): perm[new] = old local cell id, owned cells first
 *        (so that they are a prefix: the owned rows of a local Vec are then one block), each group along a Hilbert curve
 *        through the centroids xy[c*stride + 0..1].  The tiles of the operator are runs of 256 consecutive owned cells, so
 *        this is the numbering a host should give its local cells before it builds RDyMesh (DMPlexPermute; DESIGN.md
 *        section 7 note 7 has the measurements: row-major -18 %, random order 3x slower than a curve order).
 *   rdyhip_local_cell_order(..., cell_owner_rank, cell_keys, perm): the same for the owned cells; the GHOST cells are grouped by
 *        owner rank (cell_owner_rank[c], read for ghosts only) and sorted by cell_keys[c] inside a group -- the key the plan is
 *        given for them (a global cell id; or the owner's local id, which is known once the owners have renumbered: a second
 *        pass over the ghosts only).  rdyhip_halo_plan_* then lists every peer's ghosts as consecutive rows in arrival order,
 *        and rdyhip_halo_create receives in place (direct receive, above).
 *   rdyhip_copy_owned_rows(op, u_global, u_local, stream): u_local[owned cell o] = u_global[o] -- the local half of
 *        DMGlobalToLocal (device pointers; one contiguous copy when the owned cells are numbered first) */
typedef struct RDyHipHaloPlan_s *RDyHipHaloPlan;
int rdyhip_halo_plan_create(int32_t world, int32_t rank, int32_t num_ghosts, const int32_t *ghost_cell_ids, const int32_t *ghost_owner_ranks,
                            const int64_t *ghost_keys, RDyHipHaloPlan *plan);
int rdyhip_halo_plan_requests(RDyHipHaloPlan plan, const int32_t **request_counts, const int64_t **request_keys);
int rdyhip_halo_plan_finish(RDyHipHaloPlan plan, const int32_t *incoming_counts, const int64_t *incoming_keys, int32_t num_cells,
                            const int32_t *cell_is_owned, const int64_t *cell_keys);
int rdyhip_halo_plan_get(RDyHipHaloPlan plan, int32_t *npeers, const int32_t **peers, const int32_t **send_counts, const int32_t **send_cell_ids,
                         const int32_t **recv_counts, const int32_t **recv_cell_ids);
int rdyhip_halo_plan_destroy(RDyHipHaloPlan *plan);
int rdyhip_hilbert_cell_order(int32_t num_cells, const double *xy, int32_t stride, const int32_t *cell_is_owned, int32_t *perm);
int rdyhip_local_cell_order(int32_t num_cells, const double *xy, int32_t stride, const int32_t *cell_is_owned, const int32_t *cell_owner_rank,
                            const int64_t *cell_keys, int32_t *perm);
int rdyhip_copy_owned_rows(RDyHipOperator op, const double *u_global, double *u_local, void *stream);

/* ---- explicit update kept on the device (what TSEULER does between RHS calls)
 * u_local[owned cell o] += dt * f_global[o]   (PETSc TSStep_Euler VecAXPY; the
 * scatter from the global to the local vector is folded in). */
int rdyhip_axpy_owned(RDyHipOperator op, double dt, const double *f_global, double *u_local, void *stream);

/* The whole forward-Euler step in one pass (TSStep_Euler: F = RHS(U); U += dt F, with OperatorRHSFunction's zeroing
 * of F, src/rdysetup.c:1120-1172): u_local_out[owned cell] = u_local[owned cell] + dt * F, where F is what
 * rdyhip_rhs_function would produce.  Not in place: the caller ping-pongs two local state arrays (the ghost rows of
 * u_local_out are left for the next halo update).  f_global may be NULL -- the first-order and HR tiled kernels then
 * never write F (one 24 B/cell stream less and no separate axpy pass); primitive_variables, boundary fluxes and the
 * Courant diagnostic are produced as by any RHS evaluation.  `phase` / `flags` as in rdyhip_apply_phase
 * (RDYHIP_PHASE_OVERWRITE is implied). */
int rdyhip_euler_step(RDyHipOperator op, int32_t phase, int32_t flags, double dt, const double *u_local, double *u_local_out, double *f_global,
                      void *stream);

/* ---- introspection ----------------------------------------------------------
 * numbers describing the device layout, for DESIGN.md / bench.py */
typedef struct {
  int32_t num_owned_cells, num_cells, slots_per_cell, num_boundary_edges;
  int32_t num_halo_cells;     /* owned cells with a ghost neighbour */
  int32_t tiled_kernel;       /* 1: tiled LDS kernel (default), 0: cell-centric kernel (RDYHIP_KERNEL=cell) */
  int32_t num_tiles;          /* tiles of 256 owned cells */
  int32_t num_halo_tiles;     /* tiles with a ghost-adjacent cell (second order: or a ghost-adjacent first-ring cell) */
  int32_t max_tile_edges;     /* largest edge list of a tile */
  int32_t max_tile_halo_cells; /* largest number of out-of-tile neighbour cells of a tile */
  int64_t num_halo_entries;   /* sum of the tiles' halo-cell lists */
  int64_t num_edge_records;   /* sum of the tiles' edge lists (cut edges appear in two tiles) */
  int32_t owned_is_prefix;    /* 1 if owned cell o is local cell o */
  int64_t device_bytes;       /* bytes of device memory held by the operator */
  int64_t bytes_per_apply;    /* bytes one full apply must move (layout-exact, not the 176 B/cell model) */
  int32_t second_order_fused; /* 1 for a second-order operator: the gradients are formed in LDS by the flux kernel, only
                                 rdyhip_compute_gradients(RDYHIP_PHASE_HALO) is needed before the gradient exchange (the
                                 two-launch form of rounds 1-4 is gone); 0: first order */
  int32_t max_tile_ring2_cells; /* second order: largest first + second ring of a tile */
  int32_t persistent_grid;    /* workgroups of a full apply of the tiled kernel (resident workgroups per CU x CUs) */
  int32_t lds_bytes;          /* dynamic LDS per workgroup of that kernel */
  int32_t lds_fixed_layout;   /* 1 for the tiled kernels: every tile is cut to their fixed capacities at create, the LDS planes have
                                 compile-time lengths (plane offsets are instruction immediates); 0: the cell-centric kernel */
} RDyHipLayoutInfo;
int rdyhip_layout_info(RDyHipOperator op, RDyHipLayoutInfo *info);
/* the same numbers from the host-side layout pass alone (validation of the mesh, slot tables, tiles): no device is
 * touched, so a mesh and its numbering can be checked -- and every rdyhip_create argument error reproduced -- on a
 * machine without a GPU; device_bytes and persistent_grid are 0 */
int rdyhip_probe_layout(const RDyHipConfig *config, const RDyHipMesh *mesh, int32_t num_boundaries, const RDyHipBoundary *boundaries,
                        RDyHipLayoutInfo *info);

#ifdef __cplusplus
}
#endif
#endif /* RDYHIP_H */
