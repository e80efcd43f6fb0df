"""adapter/rdyhip_petsc.c, the RDycore-side translation unit: PETSc and RDycore's private headers are
absent from this image, so all that can be checked here is that the TU is valid C11 and compiles to nothing without
them (no stand-in headers anywhere), and that every ABI function it calls is declared by include/rdyhip.h."""
import os
import re
import subprocess

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
    perm = src[src.index("PetscErrorCode RDyHipPermuteLocalCells"):src.index("PetscErrorCode RDyHipCreateHaloFromDM")]
    assert "rdyhip_hilbert_cell_order" in perm and "DMPlexPermute" in perm and "DMPlexComputeCellGeometryFVM" in perm
    # every launch of the adapter goes on PETSc's stream, none on the NULL stream
    assert not re.search(r"rdyhip_(apply|rhs_function|rhs_overlapped)\([^;]*NULL\)", src)
