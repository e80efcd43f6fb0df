"""adapter/rdyhip_petsc.c, the RDycore-side translation unit: PETSc and RDycore's private headers are
absent from this image, so the TU is never built here: it compiles to nothing without them, every ABI function it calls is
declared by include/rdyhip.h, and -- LINT ONLY, no parity or boundary claim attaches to it -- it is type-checked with
`gcc -fsyntax-only` against the reference's own private headers (read in place from /root/reference, CPU container only)
plus declaration-only PETSc / libCEED prototypes kept under tests/adapter_lint/ (never compiled into an object, never linked)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "adapter", "rdyhip_petsc.c")


def test_adapter_is_an_empty_translation_unit_without_petsc(tmp_path):
    obj = str(tmp_path / "adapter.o")
    subprocess.check_call(["gcc", "-std=c11", "-pedantic", "-Wall", "-Werror", f"-I{os.path.join(ROOT, 'include')}", "-c", SRC, "-o", obj])
    syms = subprocess.run(["nm", obj], capture_output=True, text=True).stdout
    assert "CreateHipSWE" not in syms          # guarded out: nothing pretends to be PETSc here


def test_adapter_uses_only_declared_abi_functions():
    src = open(SRC).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rdyhip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(rdyhip_[a-z_0-9]+)\s*\(", hdr))
    used = set(re.findall(r"\b(rdyhip_[a-z_0-9]+)\s*\(", src))
    assert used and used <= declared, used - declared
    # the factories carry the reference's signatures (include/private/rdyoperatorimpl.h:234, 238)
    assert re.search(r"CreateHipSWEFluxOperator\(RDyConfig \*config, RDyMesh \*mesh, MPI_Comm comm, PetscInt num_boundaries, RDyBoundary \*boundaries,", src)
    assert re.search(r"CreateHipSWESourceOperator\(RDyConfig \*config, RDyMesh \*mesh, Vec external_sources, Vec material_properties, PetscOperator \*source_op\)", src)


def test_adapter_holds_the_multi_rank_binding_in_code():
    """row (h): the DM-side binding of the overlapped multi-rank RHS is code in the adapter, not prose -- the point SF is read,
    the plan goes through its one all-to-all, the halo is created on an RCCL communicator bootstrapped over MPI, and the
    RHS function replaces src/rdysetup.c:1130-1139 by the local copy + rdyhip_rhs_overlapped (in this order)"""
    src = open(SRC).read()
    assert re.search(r"PetscErrorCode RDyHipPermuteLocalCells\(DM \*dm\)", src)
    assert re.search(r"PetscErrorCode RDyHipCreateHaloFromDM\(DM dm, RDyMesh \*mesh\)", src)
    assert re.search(r"PetscErrorCode OperatorRHSFunctionHip\(TS ts, PetscReal t, Vec U, Vec F, void \*ctx\)", src)   # TSRHSFunction
    body = src[src.index("PetscErrorCode RDyHipCreateHaloFromDM"):src.index("PetscErrorCode OperatorRHSFunctionHip")]
    order = ["DMGetPointSF", "PetscSFGetGraph", "rdyhip_halo_plan_create", "rdyhip_halo_plan_requests", "MPI_Alltoall(", "MPI_Alltoallv(",
             "rdyhip_halo_plan_finish", "rdyhip_halo_plan_get", "rdyhip_comm_unique_id", "MPI_Bcast(id", "rdyhip_comm_init_rank", "rdyhip_halo_create"]
    pos = [body.index(k) for k in order]
    assert pos == sorted(pos)
    rhs = src[src.index("PetscErrorCode OperatorRHSFunctionHip"):]
    pos = [rhs.index(k) for k in ["TSGetTimeStep", "RefreshBoundaryValues", "RefreshCellFields", "rdyhip_copy_owned_rows", "rdyhip_rhs_overlapped"]]
    assert pos == sorted(pos)
    perm = src[src.index("static PetscErrorCode PermuteCells"):src.index("PetscErrorCode RDyHipCreateHaloFromDM")]
    assert "rdyhip_hilbert_cell_order" in perm and "DMPlexPermute" in perm and "DMPlexComputeCellGeometryFVM" in perm
    # second pass: the ghosts by (owner rank, owner's index), so that the exchange receives in place; every rank permutes
    assert "rdyhip_local_cell_order" in perm and "if (nc == 0) PetscFunctionReturn" not in perm
    # the setup exchange that cross-checks the plan against RDyMesh's global ids
    halo = src[src.index("PetscErrorCode RDyHipCreateHaloFromDM"):src.index("PetscErrorCode OperatorRHSFunctionHip")]
    assert halo.index("rdyhip_halo_create") < halo.index("rdyhip_halo_exchange") < halo.index("received another cell's data")
    # no device-wide synchronisation and no blocking copy on the per-advance path (the refresh is stream-ordered)
    assert "hipDeviceSynchronize" not in src and not re.search(r"\bhipMemcpy\(", src[:src.index("PetscErrorCode RDyHipCreateHaloFromDM")])
    # the fused Euler step as a TS type
    ts = src[src.index("static PetscErrorCode TSStep_RDyHipEuler"):src.index("static PetscErrorCode TSReset_RDyHipEuler")]
    pos = [ts.index(k) for k in ["RefreshBoundaryValues", "RefreshCellFields", "rdyhip_euler_step_overlapped", "VecHIPPlaceArray", "ts->ptime += ts->time_step"]]
    assert pos == sorted(pos) and 'TSRegister("rdyhip_euler"' in src
    # every launch of the adapter goes on PETSc's stream, none on the NULL stream
    assert not re.search(r"rdyhip_(apply|rhs_function|rhs_overlapped)\([^;]*NULL\)", src)


REF_INCLUDE = "/root/reference/include"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF_INCLUDE, "private")), reason="the reference's headers exist in the build container only")
def test_adapter_type_checks_against_rdycores_private_headers():
    """LINT ONLY.  `gcc -fsyntax-only -Wall -Wextra -Werror` of the adapter with the reference's include/private/*.h on the include
    path (so RDy, RDyMesh, RDyConfig, Operator, OperatorDiagnostics, PetscOperator ... are the reference's own definitions) and
    tests/adapter_lint/ standing in for <petsc.h>, <petsc/private/*.h>, <ceed/*.h> and the cmake-generated private/config.h with
    prototypes only.  Nothing is compiled to an object or linked."""
    lint = os.path.join(ROOT, "tests", "adapter_lint")
    cmd = ["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", f"-I{lint}", f"-I{REF_INCLUDE}",
           f"-I{os.path.join(ROOT, 'include')}", "-I/opt/rocm/include", SRC]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-4000:]
    # the stand-ins declare, they never define: no function body anywhere under tests/adapter_lint/
    for dirpath, _, files in os.walk(lint):
        for f in files:
            text = re.sub(r"/\*.*?\*/", "", open(os.path.join(dirpath, f)).read(), flags=re.S)
            text = re.sub(r"#define[^\n]*(\\\n[^\n]*)*", "", text)                     # macros
            assert not re.search(r"\)\s*\{", text), f"{f}: a function body in a lint header"
