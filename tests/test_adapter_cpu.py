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
