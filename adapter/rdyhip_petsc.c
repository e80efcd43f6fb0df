/*
 * rdyhip_petsc.c -- the RDycore-side adapter of librdyhip.so: ONE translation unit a maintainer adds to
 * src/ (and to src/CMakeLists.txt) to put the MI355X-native SWE operator behind RDycore's own operator
 * seam.  It is compiled only where PETSc and RDycore's private headers exist; in this repository's image
 * neither does, so the TU is empty here (tests/test_adapter_cpu.py checks exactly that it still compiles)
 * and nothing below could be run by this repository's tests.  No stand-in headers are used anywhere.
 *
 * What it implements (file:line relative to the RDycore tree):
 *
 *   CreateHipSWEFluxOperator    same signature as CreatePetscFluxOperator   (include/private/rdyoperatorimpl.h:234,
 *                               src/operator_fluxes_petsc.c:17); it also covers well_balancing = HR, for which the reference has the
 *                               separate factory CreatePetscFluxHROperator (236: the same arguments WITHOUT the MPI_Comm) -- the
 *                               dispatch in CreateOperatorSubOperators calls this one factory for both
 *   CreateHipSWESourceOperator  same signature as CreatePetscSourceOperator (rdyoperatorimpl.h:238, src/operator_sources_petsc.c:13)
 *
 * i.e. the two PetscOperators that ApplyPetscOperator (src/operator.c:656-672) applies, built with
 * PetscOperatorCreate(context, apply, destroy, &op) (rdyoperatorimpl.h:75-90, src/petsc_operator.c:11-21) exactly as
 * CreatePetscSWEInteriorFluxOperator / ...BoundaryFluxOperator / ...SourceOperator are
 * (include/private/rdysweimpl.h:10-15, src/swe/swe_petsc.c:341, 653, 948).  The native kernel fuses interior flux,
 * boundary flux and source into one launch, so the two PetscOperators share one context (the RDyHipOperator handle):
 *
 *   flux.apply    (first in ApplyPetscOperator)  refreshes the device copies of the Dirichlet values the host changed
 *                                                since the last apply (PetscObjectState of each boundary_values Vec)
 *   VecCopy(F, flux_divergence)                  copies the still untouched F (harmless: the native source term takes
 *                                                the flux sum from registers, not from that Vec)
 *   source.apply  (last)                         refreshes external sources / Manning n the same way, then ONE
 *                                                rdyhip_apply(h, dt, u_local, f_global): F += flux divergence + sources
 *
 * and the multi-rank binding (row (h) of the scope table: the DM's 1-cell ghost halo on RCCL, overlapped with the interior):
 *
 *   RDyHipPermuteLocalCells   called in CreateDM right after DMPlexDistributeOverlap (src/rdydm.c:145-157), before the labels and
 *                             RDyMeshCreateFromDM (src/rdymesh.c:95): DMPlexPermute of the local cells -- owned cells first, each
 *                             group along a Hilbert curve through the centroids (rdyhip_hilbert_cell_order) -- so that the
 *                             operator's tiles (runs of 256 consecutive owned cells) are compact patches
 *   RDyHipCreateHaloFromDM    after CreateOperator (src/rdysetup.c:1108): DMGetPointSF -> PetscSFGetGraph (for every ghost cell
 *                             its owner rank and the owner's point number) -> rdyhip_halo_plan_* with ONE MPI_Alltoall(v) for the
 *                             transpose -> an RCCL communicator over the DM's ranks (id from rank 0 by MPI_Bcast) -> rdyhip_halo_create
 *   OperatorRHSFunctionHip    replaces OperatorRHSFunction (src/rdysetup.c:1120-1172) in TSSetRHSFunction: the local half of
 *                             DMGlobalToLocal (rdyhip_copy_owned_rows) and then rdyhip_rhs_overlapped = VecZeroEntries(F) + ghost
 *                             update over RCCL on the library's stream + ResetOperatorDiagnostics + ApplyOperator, interior tiles
 *                             meanwhile (1130-1139 in one call); second order included (state and gradient exchanges inside)
 *
 *   TSRDyHipEuler ("rdyhip_euler")   a TS type that takes the WHOLE forward-Euler step in the native kernel: TSStep = ghost update over
 *                             RCCL + rdyhip_euler_step_overlapped on two ping-pong local arrays (F is never stored, no separate
 *                             VecAXPY: 0.32 instead of 0.44 ms per 10 M-cell step); the solution Vec is placed on the owned rows of the
 *                             array that holds the current state (VecHIPPlaceArray), so monitors and output read it as ever.
 *                             Selected in InitSolver when -rdy_hip_native and temporal: euler (src/rdysetup.c:1183-1185).
 *
 * plus four small hooks the patch in INTEGRATION.md section 2 wires in: RDyHipResetDiagnostics, RDyHipUpdateDiagnostics
 * (called by UpdateOperatorDiagnostics before its MPI_Allreduce, src/operator.c:867-883), RDyHipSyncBoundaryFluxes (called by
 * ExtractOperatorBoundaryFluxes before it reads boundary_fluxes_accum, src/operator.c:1069-1086) and
 * RDyHipResetBoundaryFluxesAccum (ResetAccumulatedBoundaryFluxes, src/time_series.c:505-527).
 *
 * Vec memory: with -dm_vec_type hip u_local and f_global are device Vecs and their arrays are handed to the kernel
 * as they are (VecGetArrayReadAndMemType, the pattern of src/operator.c:563-573); host Vecs are staged through device
 * scratch (correct, slow -- meant for checking the backend against the PETSc one on a workstation).
 *
 * Streams: every launch AND every copy goes on the stream of PETSc's current device context (PetscDeviceContextGetStreamHandle),
 * i.e. it is ordered with PETSc's own Vec kernels (VecAXPY of TSEULER, VecZeroEntries) without a device synchronisation.  The
 * per-advance refresh of rain, Dirichlet values and Manning n is stream-ordered too (rdyhip_refresh_field,
 * rdyhip_set_boundary_values_on: pinned staging, the library's copy stream): with a fixed time step nothing in RDyAdvance
 * drains the device (a drained MI355X runs its next ~40 launches 20-30 % slow).
 *
 * Parallel runs: through the plain PetscOperator seam (ApplyHipSource after PETSc's DMGlobalToLocal) only FIRST order
 * is possible -- a second-order apply on a mesh with ghost cells needs the ghost gradients exchanged between its two
 * phases (RDYHIP_PHASE_GRADIENTS_READY) and is refused by the library otherwise; ApplyHipSource says so itself.  The
 * second-order parallel path is OperatorRHSFunctionHip.
 */
#if defined(__has_include)
#if __has_include(<petsc.h>) && __has_include(<private/rdyoperatorimpl.h>) && __has_include(<hip/hip_runtime_api.h>)
#define RDYHIP_PETSC_ADAPTER 1
#endif
#endif

#ifdef RDYHIP_PETSC_ADAPTER

#include <hip/hip_runtime_api.h>
#include <petsc.h>
#include <petsc/private/tsimpl.h>
#include <private/rdycoreimpl.h>
#include <private/rdyoperatorimpl.h>
#include <private/rdysweimpl.h>
#include <rdyhip.h>

#define HipCall(expr)                                                                                                            \
  do {                                                                                                                           \
    hipError_t e_ = (expr);                                                                                                      \
    PetscCheck(e_ == hipSuccess, PETSC_COMM_SELF, PETSC_ERR_LIB, "%s failed: %s", #expr, hipGetErrorString(e_));                 \
  } while (0)
#define RDyHipCall(expr)                                                                                 \
  do {                                                                                                   \
    int rc_ = (expr);                                                                                    \
    PetscCheck(rc_ == 0, PETSC_COMM_SELF, (PetscErrorCode)rc_, "%s: %s", #expr, rdyhip_last_error());    \
  } while (0)

// one native operator per RDyMesh, shared by the flux and the source PetscOperator (and found again by the hooks)
typedef struct RDyHipShared {
  RDyMesh            *mesh;
  RDyHipOperator      handle;
  PetscInt            refs;
  PetscInt            num_boundaries;
  PetscInt           *boundary_num_edges;
  Vec                *boundary_values, *boundary_fluxes, *boundary_fluxes_accum;  // borrowed (owned by Operator, src/operator.c:117-129)
  PetscObjectState   *boundary_values_state;
  Vec                 external_sources, material_properties;  // borrowed (operator.c:91-96)
  PetscObjectState    external_sources_state, material_properties_state;
  OperatorDiagnostics *diagnostics;  // borrowed
  PetscReal           *d_u, *d_f;    // device staging for host Vecs
  RDyHipHalo           halo;         // multi-rank: the ghost exchange (RDyHipCreateHaloFromDM), NULL on one rank
  void                *nccl_comm;    // the library-side RCCL communicator over the DM's ranks
  PetscBool            second_order;
  struct RDyHipShared *next;
} RDyHipShared;

static RDyHipShared *shared_list = NULL;

static RDyHipShared *FindShared(RDyMesh *mesh) {
  for (RDyHipShared *s = shared_list; s; s = s->next)
    if (s->mesh == mesh) return s;
  return NULL;
}

static PetscErrorCode ReleaseShared(RDyHipShared *s) {
  PetscFunctionBegin;
  if (--s->refs > 0) PetscFunctionReturn(PETSC_SUCCESS);
  if (s->halo) RDyHipCall(rdyhip_halo_destroy(&s->halo));
  if (s->nccl_comm) RDyHipCall(rdyhip_comm_destroy(s->nccl_comm));
  RDyHipCall(rdyhip_destroy(&s->handle));
  if (s->d_u) HipCall(hipFree(s->d_u));
  if (s->d_f) HipCall(hipFree(s->d_f));
  PetscCall(PetscFree(s->boundary_num_edges));
  PetscCall(PetscFree(s->boundary_values_state));
  for (RDyHipShared **p = &shared_list; *p; p = &(*p)->next) {
    if (*p == s) {
      *p = s->next;
      break;
    }
  }
  PetscCall(PetscFree(s));
  PetscFunctionReturn(PETSC_SUCCESS);
}

// PetscInt arrays as the ABI's int32 arrays: a view in a 32-bit-index build, a narrowed copy otherwise (local indices fit)
static PetscErrorCode AsInt32(PetscInt n, const PetscInt *src, int32_t **owned_copy, const int32_t **out) {
  PetscFunctionBegin;
  *owned_copy = NULL;
  if (sizeof(PetscInt) == sizeof(int32_t)) {
    *out = (const int32_t *)src;
  } else {
    PetscCall(PetscMalloc1(n > 0 ? n : 1, owned_copy));
    for (PetscInt i = 0; i < n; ++i) (*owned_copy)[i] = (int32_t)src[i];
    *out = *owned_copy;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode AsInt64(PetscInt n, const PetscInt *src, int64_t **owned_copy, const int64_t **out) {
  PetscFunctionBegin;
  *owned_copy = NULL;
  if (sizeof(PetscInt) == sizeof(int64_t)) {
    *out = (const int64_t *)src;
  } else {
    PetscCall(PetscMalloc1(n > 0 ? n : 1, owned_copy));
    for (PetscInt i = 0; i < n; ++i) (*owned_copy)[i] = (int64_t)src[i];
    *out = *owned_copy;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

// rdyhip_create from the RDyMesh / RDyConfig / boundaries the reference's factories receive
static PetscErrorCode CreateShared(RDyConfig *config, RDyMesh *mesh, PetscInt num_boundaries, RDyBoundary *boundaries, RDyCondition *conditions,
                                   Vec *boundary_values, Vec *boundary_fluxes, Vec *boundary_fluxes_accum, OperatorDiagnostics *diagnostics,
                                   RDyHipShared **shared) {
  PetscFunctionBegin;
  PetscCheck(config->physics.flow.mode == FLOW_SWE, PETSC_COMM_WORLD, PETSC_ERR_USER, "SWE is the only supported flow model!");  // operator_fluxes_petsc.c:24
  PetscCheck(config->physics.sediment.num_classes == 0, PETSC_COMM_WORLD, PETSC_ERR_USER, "the native HIP operator has no tracers");
  RDyHipShared *s;
  PetscCall(PetscCalloc1(1, &s));
  s->mesh         = mesh;
  s->second_order = config->numerics.second_order ? PETSC_TRUE : PETSC_FALSE;

  const PetscInt nc = mesh->num_cells, ne = mesh->num_edges;
  int32_t       *c_is_owned, *c_l2o, *c_cells, *c_internal, *c_vertex = NULL;
  int64_t       *c_cgid, *c_egid;
  const int32_t *l2o, *ecells, *einternal, *evertex = NULL;
  const int64_t *cgid, *egid;
  PetscCall(PetscMalloc1(nc > 0 ? nc : 1, &c_is_owned));  // PetscBool is an enum: copy, do not cast
  for (PetscInt c = 0; c < nc; ++c) c_is_owned[c] = mesh->cells.is_owned[c] ? 1 : 0;
  PetscCall(AsInt32(nc, mesh->cells.local_to_owned, &c_l2o, &l2o));
  PetscCall(AsInt64(nc, mesh->cells.global_ids, &c_cgid, &cgid));
  PetscCall(AsInt32(2 * ne, mesh->edges.cell_ids, &c_cells, &ecells));
  PetscCall(AsInt32(mesh->num_internal_edges, mesh->edges.internal_edge_ids, &c_internal, &einternal));
  PetscCall(AsInt64(ne, mesh->edges.global_ids, &c_egid, &egid));

  RDyHipMesh hm = {0};
  hm.num_cells           = (int32_t)nc;
  hm.num_owned_cells     = (int32_t)mesh->num_owned_cells;
  hm.num_edges           = (int32_t)ne;
  hm.num_internal_edges  = (int32_t)mesh->num_internal_edges;
  hm.cell_is_owned       = c_is_owned;
  hm.cell_local_to_owned = l2o;
  hm.cell_global_ids     = cgid;
  hm.cell_areas          = mesh->cells.areas;
  hm.cell_dz_dx          = mesh->cells.dz_dx;
  hm.cell_dz_dy          = mesh->cells.dz_dy;
  hm.edge_cell_ids       = ecells;
  hm.edge_internal_ids   = einternal;
  hm.edge_global_ids     = egid;
  hm.edge_lengths        = mesh->edges.lengths;
  hm.edge_cn             = mesh->edges.cn;
  hm.edge_sn             = mesh->edges.sn;

  // hydrostatic reconstruction: the per-cell bed elevation of CreatePetscSWEInteriorFluxHROperator (src/swe/swe_petsc.c:1209-1224)
  PetscReal *zc = NULL;
  if (config->physics.flow.well_balancing == WELL_BALANCING_HR) {
    PetscCall(PetscMalloc1(nc > 0 ? nc : 1, &zc));
    for (PetscInt c = 0; c < nc; ++c) {
      if (config->grid.cell_elevation.file[0]) {
        zc[c] = mesh->cells.centroids[c].X[2];
      } else {
        PetscReal z_sum = 0.0;
        for (PetscInt v = mesh->cells.vertex_offsets[c]; v < mesh->cells.vertex_offsets[c + 1]; v++) z_sum += mesh->vertices.points[mesh->cells.vertex_ids[v]].X[2];
        zc[c] = z_sum / (PetscReal)mesh->cells.num_vertices[c];
      }
    }
    hm.cell_zc = zc;
  }
  // second order: what PrecomputeLSGradCoeffs / ReconstructFaceValues read (src/operator_fluxes_ceed.c:884-980, 1155-1206)
  if (config->numerics.second_order) {
    PetscCall(AsInt32(2 * ne, mesh->edges.vertex_ids, &c_vertex, &evertex));
    hm.num_vertices    = (int32_t)mesh->num_vertices;
    hm.cell_centroids  = (const double *)mesh->cells.centroids;  // RDyPoint is PetscReal X[3]
    hm.edge_vertex_ids = evertex;
    hm.vertex_points   = (const double *)mesh->vertices.points;
    hm.edge_is_owned   = (const int32_t *)mesh->edges.is_owned;  // PetscBool is an enum (int): who reports a cut edge's Courant number
  }

  RDyHipBoundary *hb;
  int32_t       **c_bedges;
  PetscCall(PetscCalloc1(num_boundaries > 0 ? num_boundaries : 1, &hb));
  PetscCall(PetscCalloc1(num_boundaries > 0 ? num_boundaries : 1, &c_bedges));
  PetscCall(PetscCalloc1(num_boundaries > 0 ? num_boundaries : 1, &s->boundary_num_edges));
  PetscCall(PetscCalloc1(num_boundaries > 0 ? num_boundaries : 1, &s->boundary_values_state));
  for (PetscInt b = 0; b < num_boundaries; ++b) {
    const int32_t *ids;
    PetscCall(AsInt32(boundaries[b].num_edges, boundaries[b].edge_ids, &c_bedges[b], &ids));
    hb[b].num_edges          = (int32_t)boundaries[b].num_edges;
    hb[b].edge_ids           = ids;
    hb[b].condition_type     = (int32_t)conditions[b].flow->type;  // RDyConditionType: the ABI shares its values (include/rdycore.h:133-139)
    s->boundary_num_edges[b] = boundaries[b].num_edges;
    s->boundary_values_state[b] = (PetscObjectState)-1;
  }

  // the limiter after the -no_limiter / -van_leer overrides of src/swe/swe_petsc.c:357-367
  int32_t   limiter    = RDYHIP_LIMITER_MINMOD;
  PetscBool no_limiter = PETSC_FALSE, van_leer = PETSC_FALSE;
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-no_limiter", &no_limiter, NULL));
  PetscCall(PetscOptionsGetBool(NULL, NULL, "-van_leer", &van_leer, NULL));
  if (no_limiter) limiter = RDYHIP_LIMITER_NONE;
  else if (van_leer) limiter = RDYHIP_LIMITER_VANLEER;

  RDyHipConfig hc = {0};
  hc.tiny_h           = config->physics.flow.tiny_h;
  hc.h_anuga_regular  = config->physics.flow.h_anuga_regular;
  hc.xq2018_threshold = config->physics.flow.source.xq2018_threshold;
  hc.source_method    = (int32_t)config->physics.flow.source.method;   // RDyFlowSourceMethod: shared values (rdyconfigimpl.h:52-56)
  hc.riemann          = (int32_t)config->numerics.riemann;
  hc.well_balancing   = (int32_t)config->physics.flow.well_balancing;  // none | HR (BS2002 is CEED-only, src/operator.c:388)
  hc.second_order     = config->numerics.second_order ? 1 : 0;
  hc.limiter          = limiter;
  RDyHipCall(rdyhip_create(&hc, &hm, (int32_t)num_boundaries, hb, &s->handle));  // the arrays above are borrowed during the call only

  for (PetscInt b = 0; b < num_boundaries; ++b) PetscCall(PetscFree(c_bedges[b]));
  PetscCall(PetscFree(c_bedges));
  PetscCall(PetscFree(hb));
  PetscCall(PetscFree(zc));
  PetscCall(PetscFree(c_is_owned));
  PetscCall(PetscFree(c_l2o));
  PetscCall(PetscFree(c_cgid));
  PetscCall(PetscFree(c_cells));
  PetscCall(PetscFree(c_internal));
  PetscCall(PetscFree(c_egid));
  PetscCall(PetscFree(c_vertex));

  s->num_boundaries             = num_boundaries;
  s->boundary_values            = boundary_values;
  s->boundary_fluxes            = boundary_fluxes;
  s->boundary_fluxes_accum      = boundary_fluxes_accum;
  s->diagnostics                = diagnostics;
  s->external_sources_state     = (PetscObjectState)-1;
  s->material_properties_state  = (PetscObjectState)-1;
  s->next                       = shared_list;
  shared_list                   = s;
  *shared                       = s;
  PetscFunctionReturn(PETSC_SUCCESS);
}

// the stream PETSc's own device kernels run on (the current device context's), so that our launches are ordered with them
static PetscErrorCode PetscHipStream(hipStream_t *stream) {
  PetscFunctionBegin;
  PetscDeviceContext dctx;
  void              *handle = NULL;
  PetscCall(PetscDeviceContextGetCurrentContext(&dctx));
  PetscCall(PetscDeviceContextGetStreamHandle(dctx, &handle));  // for a HIP context: a pointer to its hipStream_t
  PetscCheck(handle, PETSC_COMM_SELF, PETSC_ERR_SUP, "PETSc's current device context has no HIP stream (a host context?): run with -dm_vec_type hip");
  *stream = *(hipStream_t *)handle;
  PetscFunctionReturn(PETSC_SUCCESS);
}

// replaces an input field of the operator by a (host or device) Vec's array if the Vec changed since the last refresh: ordered
// on PETSc's stream (launches already enqueued there still see the old values), nothing blocks, nothing drains the device
static PetscErrorCode RefreshField(RDyHipShared *s, Vec v, PetscObjectState *seen, RDyHipField field, hipStream_t stream) {
  PetscFunctionBegin;
  PetscObjectState st;
  PetscCall(PetscObjectStateGet((PetscObject)v, &st));
  if (st == *seen) PetscFunctionReturn(PETSC_SUCCESS);
  PetscInt n;
  PetscCall(VecGetLocalSize(v, &n));
  const PetscScalar *a;
  PetscMemType       mt;
  PetscCall(VecGetArrayReadAndMemType(v, &a, &mt));
  RDyHipCall(rdyhip_refresh_field(s->handle, field, a, (int64_t)n, PetscMemTypeDevice(mt) ? 1 : 0, (void *)stream));  // checks n against the field
  PetscCall(VecRestoreArrayReadAndMemType(v, &a));
  *seen = st;
  PetscFunctionReturn(PETSC_SUCCESS);
}

// the Dirichlet values the host changed since the last apply (SetOperatorBoundaryValues writes the Vecs, src/operator.c:1045-1061)
static PetscErrorCode RefreshBoundaryValues(RDyHipShared *s, hipStream_t stream) {
  PetscFunctionBegin;
  for (PetscInt b = 0; b < s->num_boundaries; ++b) {
    PetscObjectState st;
    PetscCall(PetscObjectStateGet((PetscObject)s->boundary_values[b], &st));
    if (st == s->boundary_values_state[b] || s->boundary_num_edges[b] == 0) continue;
    const PetscScalar *a;
    PetscCall(VecGetArrayRead(s->boundary_values[b], &a));  // VECSEQ in the PETSc backend (operator.c:47-75): a host array [edge][3]
    RDyHipCall(rdyhip_set_boundary_values_on(s->handle, (int32_t)b, 0, 3, (int32_t)s->boundary_num_edges[b], a, (void *)stream));  // staged: `a` is free again on return
    PetscCall(VecRestoreArrayRead(s->boundary_values[b], &a));
    s->boundary_values_state[b] = st;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

// external sources / Manning n: whole-array copies when the Vecs changed (their layouts are the device fields', operator.c:91-96)
static PetscErrorCode RefreshCellFields(RDyHipShared *s, hipStream_t stream) {
  PetscFunctionBegin;
  PetscCall(RefreshField(s, s->external_sources, &s->external_sources_state, RDYHIP_FIELD_EXTERNAL_SOURCES, stream));
  PetscCall(RefreshField(s, s->material_properties, &s->material_properties_state, RDYHIP_FIELD_MANNINGS, stream));
  PetscFunctionReturn(PETSC_SUCCESS);
}

//-------------------------------------------------------------------------------------------------
// flux operator: the Dirichlet values (SetOperatorBoundaryValues writes the boundary_values Vecs, src/operator.c:1045-1061)
//-------------------------------------------------------------------------------------------------
static PetscErrorCode ApplyHipFlux(void *context, PetscOperatorFields fields, PetscReal dt, Vec u_local, Vec f_global) {
  PetscFunctionBegin;
  RDyHipShared *s = context;
  (void)fields; (void)dt; (void)u_local; (void)f_global;  // the launch itself is ApplyHipSource's
  hipStream_t stream;
  PetscCall(PetscHipStream(&stream));
  PetscCall(RefreshBoundaryValues(s, stream));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode DestroyHipFlux(void *context) {
  PetscFunctionBegin;
  PetscCall(ReleaseShared(context));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode CreateHipSWEFluxOperator(RDyConfig *config, RDyMesh *mesh, MPI_Comm comm, PetscInt num_boundaries, RDyBoundary *boundaries,
                                        RDyCondition *boundary_conditions, Vec *boundary_values, Vec *boundary_fluxes, Vec *boundary_fluxes_accum,
                                        OperatorDiagnostics *diagnostics, PetscOperator *flux_op) {
  PetscFunctionBegin;
  (void)comm;
  RDyHipShared *s;
  PetscCall(CreateShared(config, mesh, num_boundaries, boundaries, boundary_conditions, boundary_values, boundary_fluxes, boundary_fluxes_accum,
                         diagnostics, &s));
  s->refs = 1;
  PetscCall(PetscOperatorCreate(s, ApplyHipFlux, DestroyHipFlux, flux_op));
  PetscFunctionReturn(PETSC_SUCCESS);
}

//-------------------------------------------------------------------------------------------------
// source operator: sources, Manning n, and the ONE launch that evaluates flux divergence + sources
//-------------------------------------------------------------------------------------------------
static PetscErrorCode ApplyHipSource(void *context, PetscOperatorFields fields, PetscReal dt, Vec u_local, Vec f_global) {
  PetscFunctionBegin;
  RDyHipShared *s = context;
  (void)fields;
  hipStream_t stream;
  PetscCall(PetscHipStream(&stream));
  PetscCall(RefreshCellFields(s, stream));
  // second order on a partitioned mesh needs the ghost gradients exchanged between the two phases of the apply: that is
  // OperatorRHSFunctionHip's job (the library would refuse this call with "needs RDYHIP_PHASE_GRADIENTS_READY")
  PetscCheck(!(s->second_order && s->mesh->num_cells > s->mesh->num_owned_cells), PETSC_COMM_WORLD, PETSC_ERR_SUP,
             "second_order on several ranks: install OperatorRHSFunctionHip (RDyHipCreateHaloFromDM) instead of the plain PetscOperator path");

  const PetscScalar *u;
  PetscScalar       *f;
  PetscMemType       mu, mf;
  PetscInt           nu, nf;
  PetscCall(VecGetLocalSize(u_local, &nu));
  PetscCall(VecGetLocalSize(f_global, &nf));
  PetscCall(VecGetArrayReadAndMemType(u_local, &u, &mu));
  PetscCall(VecGetArrayAndMemType(f_global, &f, &mf));
  if (PetscMemTypeDevice(mu) && PetscMemTypeDevice(mf)) {
    // -dm_vec_type hip: the Vecs' device arrays as they are, on PETSc's stream; F += flux divergence + sources as the
    // reference's sub-operators do (src/swe/swe_petsc.c:301-305, 783-785)
    RDyHipCall(rdyhip_apply(s->handle, dt, u, f, (void *)stream));
  } else {
    if (!s->d_u) HipCall(hipMalloc((void **)&s->d_u, sizeof(double) * (size_t)(nu > 0 ? nu : 1)));
    if (!s->d_f) HipCall(hipMalloc((void **)&s->d_f, sizeof(double) * (size_t)(nf > 0 ? nf : 1)));
    // every copy on the same stream as the launch: ordered after PETSc's producers of the arrays and before the kernel
    HipCall(hipMemcpyAsync(s->d_u, u, sizeof(double) * (size_t)nu, PetscMemTypeDevice(mu) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    HipCall(hipMemcpyAsync(s->d_f, f, sizeof(double) * (size_t)nf, PetscMemTypeDevice(mf) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
    RDyHipCall(rdyhip_apply(s->handle, dt, s->d_u, s->d_f, (void *)stream));
    HipCall(hipMemcpyAsync(f, s->d_f, sizeof(double) * (size_t)nf, PetscMemTypeDevice(mf) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
    HipCall(hipStreamSynchronize(stream));  // the host arrays are PETSc's again on return
  }
  PetscCall(VecRestoreArrayAndMemType(f_global, &f));
  PetscCall(VecRestoreArrayReadAndMemType(u_local, &u));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode DestroyHipSource(void *context) {
  PetscFunctionBegin;
  PetscCall(ReleaseShared(context));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PetscErrorCode CreateHipSWESourceOperator(RDyConfig *config, RDyMesh *mesh, Vec external_sources, Vec material_properties, PetscOperator *source_op) {
  PetscFunctionBegin;
  (void)config;
  RDyHipShared *s = FindShared(mesh);
  PetscCheck(s, PETSC_COMM_WORLD, PETSC_ERR_ORDER, "CreateHipSWEFluxOperator must be called before CreateHipSWESourceOperator (as in CreateOperatorSubOperators, src/operator.c:195-214)");
  s->external_sources    = external_sources;
  s->material_properties = material_properties;
  s->refs++;
  PetscCall(PetscOperatorCreate(s, ApplyHipSource, DestroyHipSource, source_op));
  PetscFunctionReturn(PETSC_SUCCESS);
}

//-------------------------------------------------------------------------------------------------
// the multi-rank binding: local cell numbering, the ghost exchange from the DM's point SF, the overlapped RHS function
//-------------------------------------------------------------------------------------------------

// DMPlexPermute of the cells [c_start, c_end) by perm[new cell] = old cell (the other points keep their numbers).  Collective:
// DMPlexPermute remaps the point SF's remote indices with a PetscSFBcast, so EVERY rank must call it -- a rank without cells
// with the identity -- or the others' remote indices go stale (or the broadcast hangs).
static PetscErrorCode PermuteCells(DM *dm, PetscInt c_start, PetscInt nc, const int32_t *perm) {
  PetscFunctionBegin;
  PetscInt p_start, p_end;
  PetscCall(DMPlexGetChart(*dm, &p_start, &p_end));
  PetscInt *new_of_old;  // DMPlexPermute wants perm[old point] = new point, over the whole chart
  PetscCall(PetscMalloc1(p_end - p_start > 0 ? p_end - p_start : 1, &new_of_old));
  for (PetscInt p = p_start; p < p_end; ++p) new_of_old[p - p_start] = p;
  for (PetscInt i = 0; i < nc; ++i) new_of_old[c_start + perm[i] - p_start] = c_start + i;
  IS is;
  DM pdm;
  PetscCall(ISCreateGeneral(PETSC_COMM_SELF, p_end - p_start, new_of_old, PETSC_OWN_POINTER, &is));
  PetscCall(DMPlexPermute(*dm, is, &pdm));  // carries coordinates, labels, the local section and the point SF along
  PetscCall(ISDestroy(&is));
  PetscCall(DMDestroy(dm));
  *dm = pdm;
  PetscFunctionReturn(PETSC_SUCCESS);
}

// which local cells are ghosts (leaves of the point SF), their owner ranks and the owners' point numbers for them
static PetscErrorCode GhostCells(DM dm, PetscInt c_start, PetscInt c_end, int32_t *is_owned, int32_t *owner, int64_t *key) {
  PetscFunctionBegin;
  PetscSF            sf;
  PetscInt           nroots, nleaves;
  const PetscInt    *ilocal;
  const PetscSFNode *iremote;
  for (PetscInt c = 0; c < c_end - c_start; ++c) {
    is_owned[c] = 1;
    owner[c]    = -1;
    key[c]      = -1;
  }
  PetscCall(DMGetPointSF(dm, &sf));
  PetscCall(PetscSFGetGraph(sf, &nroots, &nleaves, &ilocal, &iremote));
  for (PetscInt i = 0; i < (nroots >= 0 ? nleaves : 0); ++i) {
    const PetscInt p = ilocal ? ilocal[i] : i;
    if (p < c_start || p >= c_end) continue;
    is_owned[p - c_start] = 0;
    owner[p - c_start]    = (int32_t)iremote[i].rank;
    key[p - c_start]      = (int64_t)iremote[i].index;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

// Renumbers the LOCAL cells of a distributed (and overlapped) DMPlex in two collective passes.  Call between
// DMPlexDistributeOverlap and everything that reads point numbers (labels made afterwards, RDyMeshCreateFromDM, the sections).
//   pass 1  owned cells first, along a Hilbert curve through the centroids (the operator's tiles are runs of 256 consecutive
//           owned cells); the ghosts after them in any order.  DMPlexPermute remaps the SF: every ghost's remote index is now
//           its owner's NEW number for it.
//   pass 2  the ghosts alone, grouped by owner rank and ascending remote index inside a group -- the order in which
//           rdyhip_halo_plan_* lists them on both sides, so that every peer's ghosts are consecutive rows in arrival order and
//           the exchange receives straight into u_local (rdyhip_halo_direct_receive; no unpack launch).  Owned cells keep the
//           numbers of pass 1, so nobody's remote indices move.
PetscErrorCode RDyHipPermuteLocalCells(DM *dm) {
  PetscFunctionBegin;
  for (int pass = 1; pass <= 2; ++pass) {
    PetscInt c_start, c_end;
    PetscCall(DMPlexGetHeightStratum(*dm, 0, &c_start, &c_end));
    const PetscInt nc = c_end - c_start;
    int32_t       *is_owned, *owner, *perm;
    int64_t       *key;
    double        *xy;
    PetscCall(PetscMalloc3(nc > 0 ? nc : 1, &is_owned, nc > 0 ? nc : 1, &owner, nc > 0 ? nc : 1, &perm));
    PetscCall(PetscMalloc1(nc > 0 ? nc : 1, &key));
    PetscCall(PetscMalloc1(nc > 0 ? 2 * nc : 1, &xy));
    PetscCall(GhostCells(*dm, c_start, c_end, is_owned, owner, key));
    for (PetscInt c = c_start; c < c_end; ++c) {
      PetscReal area, centroid[3], normal[3];
      PetscCall(DMPlexComputeCellGeometryFVM(*dm, c, &area, centroid, normal));
      xy[2 * (c - c_start)]     = centroid[0];
      xy[2 * (c - c_start) + 1] = centroid[1];
    }
    // perm[new cell] = old cell.  Pass 2 runs on the numbering of pass 1: its owned prefix is already in curve order (the sort
    // is stable in the old cell id for equal keys and the Hilbert keys are unchanged), only the ghosts move.
    if (pass == 1) RDyHipCall(rdyhip_hilbert_cell_order((int32_t)nc, xy, 2, is_owned, perm));
    else RDyHipCall(rdyhip_local_cell_order((int32_t)nc, xy, 2, is_owned, owner, key, perm));
    PetscCall(PermuteCells(dm, c_start, nc, perm));  // every rank, also with nc == 0
    PetscCall(PetscFree3(is_owned, owner, perm));
    PetscCall(PetscFree(key));
    PetscCall(PetscFree(xy));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

// The exchange pattern of this rank from the DM's point SF, and an RCCL communicator of the library's own over the DM's
// ranks.  After CreateOperator (the native operator of `mesh` must exist).  RDyMesh numbers its cells c - cStart
// (src/rdymesh.c:113-118), and so does every other rank: an SF leaf (p, {rank, index}) that is a cell says "my local cell
// p - cStart is rank's local cell index - cStart(rank)"; the height-0 stratum of a DMPlex starts at point 0 on every rank.
PetscErrorCode RDyHipCreateHaloFromDM(DM dm, RDyMesh *mesh) {
  PetscFunctionBegin;
  RDyHipShared *s = FindShared(mesh);
  PetscCheck(s, PETSC_COMM_WORLD, PETSC_ERR_ORDER, "RDyHipCreateHaloFromDM must follow CreateOperator (no native operator for this mesh yet)");
  MPI_Comm    comm;
  PetscMPIInt size, rank;
  PetscCall(PetscObjectGetComm((PetscObject)dm, &comm));
  PetscCallMPI(MPI_Comm_size(comm, &size));
  PetscCallMPI(MPI_Comm_rank(comm, &rank));
  if (size == 1 || s->halo) PetscFunctionReturn(PETSC_SUCCESS);

  PetscInt c_start, c_end;
  PetscCall(DMPlexGetHeightStratum(dm, 0, &c_start, &c_end));
  PetscCheck(c_start == 0, comm, PETSC_ERR_SUP, "the cells of this DMPlex do not start at point 0 (cStart = %" PetscInt_FMT ")", c_start);
  PetscSF            sf;
  PetscInt           nroots, nleaves;
  const PetscInt    *ilocal;
  const PetscSFNode *iremote;
  PetscCall(DMGetPointSF(dm, &sf));
  PetscCall(PetscSFGetGraph(sf, &nroots, &nleaves, &ilocal, &iremote));
  if (nroots < 0) nleaves = 0;  // graph not set: a rank without shared points

  // the ghost CELLS among the leaves: local id, owner, the owner's local id
  int32_t *g_cell, *g_owner;
  int64_t *g_key;
  PetscCall(PetscMalloc3(nleaves > 0 ? nleaves : 1, &g_cell, nleaves > 0 ? nleaves : 1, &g_owner, nleaves > 0 ? nleaves : 1, &g_key));
  int32_t ng = 0;
  for (PetscInt i = 0; i < nleaves; ++i) {
    const PetscInt p = ilocal ? ilocal[i] : i;
    if (p < c_start || p >= c_end) continue;
    PetscCheck(!mesh->cells.is_owned[p - c_start], comm, PETSC_ERR_PLIB, "SF leaf %" PetscInt_FMT " is a cell RDyMesh counts as owned", p);
    g_cell[ng]  = (int32_t)(p - c_start);
    g_owner[ng] = (int32_t)iremote[i].rank;
    g_key[ng]   = (int64_t)iremote[i].index;  // = the owner's local cell id (its cStart is 0 too)
    ++ng;
  }
  PetscCheck(ng == mesh->num_cells - mesh->num_owned_cells, comm, PETSC_ERR_PLIB, "%d ghost cells in the point SF, %" PetscInt_FMT " in RDyMesh", ng,
             mesh->num_cells - mesh->num_owned_cells);

  RDyHipHaloPlan plan;
  RDyHipCall(rdyhip_halo_plan_create((int32_t)size, (int32_t)rank, ng, g_cell, g_owner, g_key, &plan));
  const int32_t *req_counts;
  const int64_t *req_keys;
  RDyHipCall(rdyhip_halo_plan_requests(plan, &req_counts, &req_keys));
  // the transpose of the leaf -> root relation: who asks ME for what (PetscSFGetRootRanks holds the same lists; one
  // all-to-all keeps this independent of the SF's internal rank ordering)
  PetscMPIInt *in_counts, *sdispl, *rdispl;
  PetscCall(PetscMalloc3(size, &in_counts, size + 1, &sdispl, size + 1, &rdispl));
  PetscCallMPI(MPI_Alltoall((void *)req_counts, 1, MPI_INT, in_counts, 1, MPI_INT, comm));
  sdispl[0] = rdispl[0] = 0;
  for (PetscMPIInt r = 0; r < size; ++r) {
    sdispl[r + 1] = sdispl[r] + req_counts[r];
    rdispl[r + 1] = rdispl[r] + in_counts[r];
  }
  int64_t *in_keys;
  PetscCall(PetscMalloc1(rdispl[size] > 0 ? rdispl[size] : 1, &in_keys));
  PetscCallMPI(MPI_Alltoallv((void *)req_keys, (int *)req_counts, sdispl, MPI_INT64_T, in_keys, in_counts, rdispl, MPI_INT64_T, comm));

  int32_t *owned;
  PetscCall(PetscMalloc1(mesh->num_cells > 0 ? mesh->num_cells : 1, &owned));
  for (PetscInt c = 0; c < mesh->num_cells; ++c) owned[c] = mesh->cells.is_owned[c] ? 1 : 0;
  RDyHipCall(rdyhip_halo_plan_finish(plan, in_counts, in_keys, (int32_t)mesh->num_cells, owned, NULL));  // keys are local cell ids
  int32_t        npeers;
  const int32_t *peers, *send_counts, *send_cells, *recv_counts, *recv_cells;
  RDyHipCall(rdyhip_halo_plan_get(plan, &npeers, &peers, &send_counts, &send_cells, &recv_counts, &recv_cells));

  // RCCL communicator over the DM's ranks: rank 0 draws the id, MPI carries it (PETSc has selected this rank's device)
  char id[RDYHIP_COMM_ID_BYTES];
  int  ok = 1;
  if (rank == 0) ok = rdyhip_comm_unique_id(id) == 0;
  PetscCallMPI(MPI_Bcast(&ok, 1, MPI_INT, 0, comm));  // a failure on rank 0 is everybody's, before anybody blocks in RCCL
  PetscCheck(ok, comm, PETSC_ERR_LIB, "ncclGetUniqueId failed on rank 0");
  PetscCallMPI(MPI_Bcast(id, RDYHIP_COMM_ID_BYTES, MPI_BYTE, 0, comm));
  RDyHipCall(rdyhip_comm_init_rank((int32_t)size, (int32_t)rank, id, &s->nccl_comm));
  RDyHipCall(rdyhip_halo_create(s->handle, s->nccl_comm, npeers, peers, send_counts, send_cells, recv_counts, recv_cells, &s->halo));

  // Cross-check of the whole chain (SF remap under DMPlexPermute, plan, communicator, exchange) with one exchange at setup:
  // every owned cell carries its global id, every ghost row must come back holding the id RDyMesh has for that ghost.  The
  // plan's own checks only catch requests for cells that are not owned; a request for the WRONG owned cell would be silent.
  {
    const PetscInt nloc = mesh->num_cells;
    double        *h_ids, *d_ids;
    PetscCall(PetscMalloc1(nloc > 0 ? 3 * nloc : 1, &h_ids));
    for (PetscInt c = 0; c < nloc; ++c)
      for (int k = 0; k < 3; ++k) h_ids[3 * c + k] = mesh->cells.is_owned[c] ? (double)mesh->cells.global_ids[c] : -1.0;
    HipCall(hipMalloc((void **)&d_ids, sizeof(double) * (size_t)(nloc > 0 ? 3 * nloc : 1)));
    HipCall(hipMemcpy(d_ids, h_ids, sizeof(double) * 3 * (size_t)nloc, hipMemcpyHostToDevice));
    RDyHipCall(rdyhip_halo_exchange(s->halo, d_ids, 3, NULL));
    HipCall(hipStreamSynchronize(NULL));
    HipCall(hipMemcpy(h_ids, d_ids, sizeof(double) * 3 * (size_t)nloc, hipMemcpyDeviceToHost));
    HipCall(hipFree(d_ids));
    PetscInt bad = 0;
    for (PetscInt c = 0; c < nloc; ++c)
      if (!mesh->cells.is_owned[c] && h_ids[3 * c] != (double)mesh->cells.global_ids[c]) ++bad;
    PetscCall(PetscFree(h_ids));
    // every rank fails together: a rank that stopped alone would leave the others waiting in their next RCCL call
    {
      MPI_Comm comm;
      PetscCall(PetscObjectGetComm((PetscObject)dm, &comm));
      PetscCallMPI(MPI_Allreduce(MPI_IN_PLACE, &bad, 1, MPIU_INT, MPI_SUM, comm));
      PetscCheck(bad == 0, comm, PETSC_ERR_PLIB, "%" PetscInt_FMT " ghost cells (over all ranks) received another cell's data in the setup exchange: the point SF and RDyMesh disagree", bad);
    }
  }

  RDyHipCall(rdyhip_halo_plan_destroy(&plan));
  PetscCall(PetscFree(owned));
  PetscCall(PetscFree(in_keys));
  PetscCall(PetscFree3(in_counts, sdispl, rdispl));
  PetscCall(PetscFree3(g_cell, g_owner, g_key));
  PetscFunctionReturn(PETSC_SUCCESS);
}

// OperatorRHSFunction (src/rdysetup.c:1120-1172) on the native operator: TSSetRHSFunction(ts, NULL, OperatorRHSFunctionHip, rdy)
// when -rdy_hip_native is on and the Vecs are device Vecs (-dm_vec_type hip).
PetscErrorCode OperatorRHSFunctionHip(TS ts, PetscReal t, Vec U, Vec F, void *ctx) {
  PetscFunctionBegin;
  RDy           rdy = ctx;
  RDyHipShared *s   = FindShared(&rdy->mesh);
  PetscCheck(s, rdy->comm, PETSC_ERR_ORDER, "no native operator for this RDy (CreateOperator with -rdy_hip_native first)");
  PetscCheck(s->halo || rdy->mesh.num_cells == rdy->mesh.num_owned_cells, rdy->comm, PETSC_ERR_ORDER,
             "the mesh has ghost cells: call RDyHipCreateHaloFromDM after CreateOperator");
  (void)t;
  PetscScalar dt;
  PetscCall(TSGetTimeStep(ts, &dt));

  // what the two PetscOperators' apply functions do before the launch: inputs the host changed since the last RHS
  hipStream_t stream;
  PetscCall(PetscHipStream(&stream));
  PetscCall(RefreshBoundaryValues(s, stream));
  PetscCall(RefreshCellFields(s, stream));

  const PetscScalar *u;
  PetscScalar       *ul, *f;
  PetscMemType       mu, ml, mf;
  PetscCall(VecGetArrayReadAndMemType(U, &u, &mu));
  PetscCall(VecGetArrayAndMemType(rdy->u_local, &ul, &ml));
  PetscCall(VecGetArrayWriteAndMemType(F, &f, &mf));
  PetscCheck(PetscMemTypeDevice(mu) && PetscMemTypeDevice(ml) && PetscMemTypeDevice(mf), rdy->comm, PETSC_ERR_SUP,
             "OperatorRHSFunctionHip needs device Vecs (-dm_vec_type hip); host Vecs go through the PetscOperator path");
  // DMGlobalToLocal, local half: the owned rows of u_local (one contiguous copy after RDyHipPermuteLocalCells) ...
  RDyHipCall(rdyhip_copy_owned_rows(s->handle, u, ul, (void *)stream));
  if (s->halo) {
    // ... and the ghost rows over RCCL, hidden behind the interior tiles; F is overwritten (VecZeroEntries + accumulate),
    // the Courant diagnostic starts over (ResetOperatorDiagnostics)
    RDyHipCall(rdyhip_rhs_overlapped(s->handle, s->halo, dt, ul, f, (void *)stream));
  } else {
    RDyHipCall(rdyhip_rhs_function(s->handle, dt, ul, f, (void *)stream));
  }
  PetscCall(VecRestoreArrayWriteAndMemType(F, &f));
  PetscCall(VecRestoreArrayAndMemType(rdy->u_local, &ul));
  PetscCall(VecRestoreArrayReadAndMemType(U, &u));

  // debug-level logging as in the reference (rdysetup.c:1155-1169)
  if (rdy->config.logging.level >= LOG_DEBUG) {
    PetscCall(UpdateOperatorDiagnostics(rdy->operator));
    OperatorDiagnostics diagnostics;
    PetscCall(GetOperatorDiagnostics(rdy->operator, &diagnostics));
    PetscReal time;
    PetscInt  stepnum;
    PetscCall(TSGetTime(ts, &time));
    PetscCall(TSGetStepNumber(ts, &stepnum));
    RDyLogDebug(rdy, "[%" PetscInt_FMT "] Time = %f [%s] Max courant number %g", stepnum, ConvertTimeFromSeconds(time, rdy->config.time.unit),
                TimeUnitAsString(rdy->config.time.unit), diagnostics.courant_number.max_courant_num);
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

//-------------------------------------------------------------------------------------------------
// TSRDyHipEuler: forward Euler with the update fused into the RHS kernel (rdyhip_euler_step_overlapped)
//-------------------------------------------------------------------------------------------------
// TSEULER evaluates F = RHS(U) and then U += dt F with a separate VecAXPY (TSStep_Euler): 0.44 ms per 10 M-cell step, of which
// the axpy pass and the F store are 0.12.  This TS type hands the whole step to the native kernel: two local arrays ([num_cells][3],
// owned rows first) ping-pong, the kernel reads one and writes the owned rows of the other, the ghost rows of the input are
// refreshed over RCCL at the start of each step, F never exists.  The TS's solution Vec (rdy->u_global, TSSetSolution) does not get
// copied either: it is PLACED on the owned rows of whichever array holds the current state (VecHIPPlaceArray), so monitors,
// output and RDyGet* read the current solution as ever, and whatever the host writes into it (initial conditions, a restart,
// RDySet*) lands in that array -- noticed through the Vec's PetscObjectState, which makes the next step pack its send cells
// again (rdyhip_halo_invalidate).  Needs device Vecs (-dm_vec_type hip) and owned cells numbered first
// (RDyHipPermuteLocalCells).  Registered as "rdyhip_euler"; InitSolver selects it instead of TSEULER.
typedef struct {
  RDy              rdy;
  double          *d_state[2];  // the two local arrays
  int              cur;         // which one holds the current state
  PetscBool        placed;      // vec_sol's array is d_state[cur]
  PetscObjectState seen;        // vec_sol's state after our last step
} TS_RDyHipEuler;

static PetscErrorCode TSSetUp_RDyHipEuler(TS ts) {
  PetscFunctionBegin;
  TS_RDyHipEuler *e = (TS_RDyHipEuler *)ts->data;
  PetscCall(TSGetApplicationContext(ts, &e->rdy));
  PetscCheck(e->rdy, PETSC_COMM_WORLD, PETSC_ERR_ORDER, "TSRDyHipEuler needs the RDy as application context (TSSetApplicationContext, src/rdysetup.c:1198)");
  RDyHipShared *s = FindShared(&e->rdy->mesh);
  PetscCheck(s, e->rdy->comm, PETSC_ERR_ORDER, "no native operator for this RDy (CreateOperator with -rdy_hip_native first)");
  PetscCheck(s->halo || e->rdy->mesh.num_cells == e->rdy->mesh.num_owned_cells, e->rdy->comm, PETSC_ERR_ORDER,
             "the mesh has ghost cells: call RDyHipCreateHaloFromDM after CreateOperator");
  // the solution Vec is placed on the first 3 * num_owned_cells values of a local array: the owned cells must be numbered first
  RDyHipLayoutInfo info;
  RDyHipCall(rdyhip_layout_info(s->handle, &info));
  PetscCheck(info.owned_is_prefix, e->rdy->comm, PETSC_ERR_ORDER, "TSRDyHipEuler needs the owned cells numbered before the ghosts: call RDyHipPermuteLocalCells in CreateDM");
  const size_t bytes = sizeof(double) * 3 * (size_t)(e->rdy->mesh.num_cells > 0 ? e->rdy->mesh.num_cells : 1);
  for (int k = 0; k < 2; ++k)
    if (!e->d_state[k]) {
      HipCall(hipMalloc((void **)&e->d_state[k], bytes));
      HipCall(hipMemset(e->d_state[k], 0, bytes));
    }
  // the state pack rides on the step kernel (first order, HR, second order).  Not available with RDYHIP_KERNEL=cell (an A/B
  // configuration): the call then says so and the pack launch simply stays -- not an error of the run
  if (s->halo) (void)rdyhip_halo_fuse_pack(s->halo, 1);
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode TSStep_RDyHipEuler(TS ts) {
  PetscFunctionBegin;
  TS_RDyHipEuler *e   = (TS_RDyHipEuler *)ts->data;
  RDy             rdy = e->rdy;
  RDyHipShared   *s   = FindShared(&rdy->mesh);
  Vec             U   = ts->vec_sol;
  hipStream_t     stream;
  PetscCall(PetscHipStream(&stream));
  PetscCall(TSPreStage(ts, ts->ptime));
  PetscCall(RefreshBoundaryValues(s, stream));
  PetscCall(RefreshCellFields(s, stream));

  PetscObjectState st;
  PetscCall(PetscObjectStateGet((PetscObject)U, &st));
  if (!e->placed) {
    // first step (or after a reset): the state is in U's own array -> owned rows of d_state[cur]
    const PetscScalar *u;
    PetscMemType       mu;
    PetscCall(VecGetArrayReadAndMemType(U, &u, &mu));
    PetscCheck(PetscMemTypeDevice(mu), rdy->comm, PETSC_ERR_SUP, "TSRDyHipEuler needs device Vecs (-dm_vec_type hip)");
    RDyHipCall(rdyhip_copy_owned_rows(s->handle, u, e->d_state[e->cur], (void *)stream));
    PetscCall(VecRestoreArrayReadAndMemType(U, &u));
    if (s->halo) RDyHipCall(rdyhip_halo_invalidate(s->halo));
  } else if (st != e->seen) {
    // somebody wrote the solution since our last step: it went straight into d_state[cur] (U is placed there), but the send
    // rows the last kernel packed are stale
    if (s->halo) RDyHipCall(rdyhip_halo_invalidate(s->halo));
  }
  double *in = e->d_state[e->cur], *out = e->d_state[1 - e->cur];
  if (s->halo) RDyHipCall(rdyhip_euler_step_overlapped(s->handle, s->halo, ts->time_step, in, out, NULL, (void *)stream));
  else RDyHipCall(rdyhip_euler_step(s->handle, RDYHIP_PHASE_ALL, RDYHIP_PHASE_RESET_DIAGNOSTICS, ts->time_step, in, out, NULL, (void *)stream));
  // the solution Vec now IS the owned rows of `out` (owned cells are numbered first: one contiguous block of 3 * no values)
  if (e->placed) PetscCall(VecHIPResetArray(U));
  PetscCall(VecHIPPlaceArray(U, out));
  e->placed = PETSC_TRUE;
  e->cur    = 1 - e->cur;
  PetscCall(PetscObjectStateIncrease((PetscObject)U));
  PetscCall(PetscObjectStateGet((PetscObject)U, &e->seen));
  ts->ptime += ts->time_step;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode TSReset_RDyHipEuler(TS ts) {
  PetscFunctionBegin;
  TS_RDyHipEuler *e = (TS_RDyHipEuler *)ts->data;
  if (e->placed && ts->vec_sol) {
    // give U its own array back, holding the current state
    hipStream_t stream;
    PetscCall(PetscHipStream(&stream));
    const double *cur = e->d_state[e->cur];
    PetscCall(VecHIPResetArray(ts->vec_sol));
    PetscScalar *u;
    PetscMemType mu;
    PetscInt     n;
    PetscCall(VecGetLocalSize(ts->vec_sol, &n));
    PetscCall(VecGetArrayWriteAndMemType(ts->vec_sol, &u, &mu));
    HipCall(hipMemcpyAsync(u, cur, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
    HipCall(hipStreamSynchronize(stream));
    PetscCall(VecRestoreArrayWriteAndMemType(ts->vec_sol, &u));
    e->placed = PETSC_FALSE;
  }
  for (int k = 0; k < 2; ++k) {
    if (e->d_state[k]) HipCall(hipFree(e->d_state[k]));
    e->d_state[k] = NULL;
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode TSDestroy_RDyHipEuler(TS ts) {
  PetscFunctionBegin;
  PetscCall(TSReset_RDyHipEuler(ts));
  PetscCall(PetscFree(ts->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode TSCreate_RDyHipEuler(TS ts) {
  PetscFunctionBegin;
  TS_RDyHipEuler *e;
  PetscCall(PetscNew(&e));
  ts->data         = (void *)e;
  ts->ops->setup   = TSSetUp_RDyHipEuler;
  ts->ops->step    = TSStep_RDyHipEuler;
  ts->ops->reset   = TSReset_RDyHipEuler;
  ts->ops->destroy = TSDestroy_RDyHipEuler;
  PetscFunctionReturn(PETSC_SUCCESS);
}

// once, before InitSolver: TSRegister makes "rdyhip_euler" a TSType; InitSolver then does TSSetType(rdy->ts, "rdyhip_euler")
// instead of TSEULER when the native backend is on, the rest of InitSolver unchanged
PetscErrorCode RDyHipRegisterTS(void) {
  PetscFunctionBegin;
  PetscCall(TSRegister("rdyhip_euler", TSCreate_RDyHipEuler));
  PetscFunctionReturn(PETSC_SUCCESS);
}

//-------------------------------------------------------------------------------------------------
// hooks (INTEGRATION.md section 2 shows where the three calls go)
//-------------------------------------------------------------------------------------------------

// ResetOperatorDiagnostics (src/operator.c:772-784): the device-side running maximum as well
PetscErrorCode RDyHipResetDiagnostics(RDyMesh *mesh) {
  PetscFunctionBegin;
  RDyHipShared *s = FindShared(mesh);
  if (s) RDyHipCall(rdyhip_reset_diagnostics(s->handle, NULL));
  PetscFunctionReturn(PETSC_SUCCESS);
}

// UpdateOperatorDiagnostics (src/operator.c:867-883), before its MPI_Allreduce with MPI_MAX_COURANT_NUMBER: the local
// {max_courant_num, global_edge_id, global_cell_id} comes from the device (16 bytes), where the PETSc operators update it in place
PetscErrorCode RDyHipUpdateDiagnostics(RDyMesh *mesh) {
  PetscFunctionBegin;
  RDyHipShared *s = FindShared(mesh);
  if (!s) PetscFunctionReturn(PETSC_SUCCESS);
  RDyHipCourant c;
  RDyHipCall(rdyhip_update_diagnostics(s->handle, NULL));
  RDyHipCall(rdyhip_get_diagnostics(s->handle, &c));
  s->diagnostics->courant_number.max_courant_num = c.max_courant_num;
  s->diagnostics->courant_number.global_edge_id  = (PetscInt)c.global_edge_id;
  s->diagnostics->courant_number.global_cell_id  = (PetscInt)c.global_cell_id;
  PetscFunctionReturn(PETSC_SUCCESS);
}

// ExtractOperatorBoundaryFluxes (rdyoperatorimpl.h:256) reads boundary_fluxes[b] / boundary_fluxes_accum[b]: bring them
// over from the device first (O(boundary edges), only when somebody asks)
PetscErrorCode RDyHipSyncBoundaryFluxes(RDyMesh *mesh) {
  PetscFunctionBegin;
  RDyHipShared *s = FindShared(mesh);
  if (!s) PetscFunctionReturn(PETSC_SUCCESS);
  for (PetscInt b = 0; b < s->num_boundaries; ++b) {
    if (s->boundary_num_edges[b] == 0) continue;
    PetscScalar *a;
    PetscCall(VecGetArray(s->boundary_fluxes[b], &a));
    RDyHipCall(rdyhip_get_boundary_fluxes(s->handle, (int32_t)b, 0, (int32_t)s->boundary_num_edges[b], a));
    PetscCall(VecRestoreArray(s->boundary_fluxes[b], &a));
    PetscCall(VecGetArray(s->boundary_fluxes_accum[b], &a));
    RDyHipCall(rdyhip_get_boundary_fluxes(s->handle, (int32_t)b, 1, (int32_t)s->boundary_num_edges[b], a));
    PetscCall(VecRestoreArray(s->boundary_fluxes_accum[b], &a));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

// ResetAccumulatedBoundaryFluxes (src/time_series.c:505-527) zeroes the boundary_fluxes_accum Vecs after each time-series
// record: the device-side accumulation starts over with them
PetscErrorCode RDyHipResetBoundaryFluxesAccum(RDyMesh *mesh) {
  PetscFunctionBegin;
  RDyHipShared *s = FindShared(mesh);
  if (s) RDyHipCall(rdyhip_reset_boundary_fluxes_accum(s->handle));
  PetscFunctionReturn(PETSC_SUCCESS);
}

#else  /* no PETSc / RDycore headers in this build environment: the adapter is not part of the build */

typedef int rdyhip_petsc_adapter_not_built;

#endif
