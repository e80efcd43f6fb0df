/* LINT ONLY.  Declaration-only stand-ins for the PETSc / MPI names that adapter/rdyhip_petsc.c and RDycore's private headers
 * mention, so that `gcc -fsyntax-only` can type-check the adapter in an image that has no PETSc (tests/test_adapter_cpu.py,
 * CPU container only).  Nothing here is ever compiled into an object, linked, shipped, or used as evidence of parity or of a
 * working boundary: prototypes follow PETSc's documented signatures as far as the adapter uses them, bodies do not exist. */
#ifndef RDYHIP_LINT_PETSC_DECLS_H
#define RDYHIP_LINT_PETSC_DECLS_H
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

typedef double  PetscReal;
typedef double  PetscScalar;
typedef int     PetscInt;
typedef int     PetscMPIInt;
typedef int64_t PetscInt64;
typedef int64_t PetscObjectState;
typedef int     PetscClassId;
typedef int     PetscLogEvent;
typedef int     PetscLogStage;
typedef double  PetscLogDouble;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef enum { PETSC_SUCCESS = 0, PETSC_ERR_MEM = 55, PETSC_ERR_SUP = 56, PETSC_ERR_ARG_SIZ = 60, PETSC_ERR_ARG_OUTOFRANGE = 63, PETSC_ERR_ORDER = 58,
               PETSC_ERR_LIB = 76, PETSC_ERR_PLIB = 77, PETSC_ERR_USER = 83 } PetscErrorCode;
typedef enum { PETSC_MEMTYPE_HOST = 0, PETSC_MEMTYPE_DEVICE = 1, PETSC_MEMTYPE_HIP = 5 } PetscMemType;
typedef enum { PETSC_COPY_VALUES, PETSC_OWN_POINTER, PETSC_USE_POINTER } PetscCopyMode;
typedef enum { INSERT_VALUES = 1, ADD_VALUES = 2 } InsertMode;
#define PetscMemTypeDevice(m) (((m) & 0x1) == 0x1)
#define PETSC_MAX_PATH_LEN 4096
#define PetscInt_FMT "d"
#define PETSC_INTERN extern
#define PETSC_EXTERN extern
#define PETSC_UNUSED __attribute__((unused))

typedef int MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
#define MPI_INT 1
#define MPI_INT64_T 2
#define MPI_BYTE 3
#define MPI_DOUBLE 4
#define MPIU_INT MPI_INT
#define MPI_SUM 1
#define MPI_IN_PLACE ((void *)1)
extern MPI_Comm PETSC_COMM_WORLD, PETSC_COMM_SELF;
int MPI_Comm_size(MPI_Comm, int *);
int MPI_Comm_rank(MPI_Comm, int *);
int MPI_Alltoall(const void *, int, MPI_Datatype, void *, int, MPI_Datatype, MPI_Comm);
int MPI_Alltoallv(const void *, const int *, const int *, MPI_Datatype, void *, const int *, const int *, MPI_Datatype, MPI_Comm);
int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm);
int MPI_Allreduce(const void *, void *, int, MPI_Datatype, MPI_Op, MPI_Comm);

struct _p_PetscObject {
  MPI_Comm comm;
};
typedef struct _p_PetscObject *PetscObject;
#define PETSCHEADER(ObjectOps) \
  struct _p_PetscObject hdr;   \
  ObjectOps             ops[1]
typedef struct _p_Vec         *Vec;
typedef struct _p_Mat         *Mat;
typedef struct _p_DM          *DM;
typedef struct _p_TS          *TS;
typedef struct _p_IS          *IS;
typedef struct _p_PetscSF     *PetscSF;
typedef struct _p_PetscSection *PetscSection;
typedef struct _p_DMLabel     *DMLabel;
typedef struct _p_PetscViewer *PetscViewer;
typedef struct _p_PetscBag    *PetscBag;
typedef struct _p_VecScatter  *VecScatter;
typedef struct _p_PetscDeviceContext *PetscDeviceContext;
typedef struct _p_PetscViewerAndFormat PetscViewerAndFormat;
typedef int         PetscViewerFormat;
typedef const char *VecType;
typedef const char *MatType;
typedef const char *TSType;
typedef const char *DMType;
typedef struct {
  PetscInt rank, index;
} PetscSFNode;
typedef enum { TS_CONVERGED_ITERATING = 0, TS_CONVERGED_TIME = 1 } TSConvergedReason;

/* error handling: early return up the stack */
#define PetscFunctionBegin
#define PetscFunctionBeginUser
#define PetscFunctionReturn(x) return (x)
#define PetscCall(...)                                  \
  do {                                                  \
    PetscErrorCode ierr_ = (PetscErrorCode)(__VA_ARGS__); \
    if (ierr_ != PETSC_SUCCESS) return ierr_;           \
  } while (0)
#define PetscCallMPI(...) PetscCall(__VA_ARGS__)
PetscErrorCode PetscErrorPrintfLint(MPI_Comm, PetscErrorCode, const char *, ...) __attribute__((format(printf, 3, 4)));
#define PetscCheck(cond, comm, code, ...)                                        \
  do {                                                                           \
    if (!(cond)) return PetscErrorPrintfLint(comm, (PetscErrorCode)(code), __VA_ARGS__); \
  } while (0)
#define PetscAssert PetscCheck

PetscErrorCode PetscMallocLint(size_t, void *);
#define PetscMalloc1(n, p) PetscMallocLint((size_t)(n) * sizeof(**(p)), (p))
#define PetscCalloc1(n, p) PetscMallocLint((size_t)(n) * sizeof(**(p)), (p))
#define PetscMalloc3(n1, p1, n2, p2, n3, p3) (PetscMalloc1(n1, p1) || PetscMalloc1(n2, p2) || PetscMalloc1(n3, p3))
PetscErrorCode PetscFreeLint(void *);
#define PetscFree(p) PetscFreeLint((void *)(p))
#define PetscFree3(a, b, c) (PetscFree(a) || PetscFree(b) || PetscFree(c))
#define PetscRealloc(n, p) PetscMallocLint((size_t)(n), (p))

PetscErrorCode PetscFPrintf(MPI_Comm, FILE *, const char[], ...);
PetscErrorCode PetscObjectGetComm(PetscObject, MPI_Comm *);
PetscErrorCode PetscObjectStateGet(PetscObject, PetscObjectState *);
PetscErrorCode PetscObjectStateIncrease(PetscObject);
PetscErrorCode PetscOptionsGetBool(void *, const char *, const char *, PetscBool *, PetscBool *);
PetscErrorCode PetscOptionsHasName(void *, const char *, const char *, PetscBool *);

PetscErrorCode VecGetLocalSize(Vec, PetscInt *);
PetscErrorCode VecGetSize(Vec, PetscInt *);
PetscErrorCode VecGetArray(Vec, PetscScalar **);
PetscErrorCode VecRestoreArray(Vec, PetscScalar **);
PetscErrorCode VecGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecGetArrayAndMemType(Vec, PetscScalar **, PetscMemType *);
PetscErrorCode VecRestoreArrayAndMemType(Vec, PetscScalar **);
PetscErrorCode VecGetArrayReadAndMemType(Vec, const PetscScalar **, PetscMemType *);
PetscErrorCode VecRestoreArrayReadAndMemType(Vec, const PetscScalar **);
PetscErrorCode VecGetArrayWriteAndMemType(Vec, PetscScalar **, PetscMemType *);
PetscErrorCode VecRestoreArrayWriteAndMemType(Vec, PetscScalar **);
PetscErrorCode VecHIPPlaceArray(Vec, const PetscScalar *);
PetscErrorCode VecHIPResetArray(Vec);
PetscErrorCode VecDuplicate(Vec, Vec *);
PetscErrorCode VecDestroy(Vec *);
PetscErrorCode VecCopy(Vec, Vec);

PetscErrorCode ISCreateGeneral(MPI_Comm, PetscInt, const PetscInt[], PetscCopyMode, IS *);
PetscErrorCode ISDestroy(IS *);

PetscErrorCode DMDestroy(DM *);
PetscErrorCode DMGetPointSF(DM, PetscSF *);
PetscErrorCode DMPlexGetHeightStratum(DM, PetscInt, PetscInt *, PetscInt *);
PetscErrorCode DMPlexGetChart(DM, PetscInt *, PetscInt *);
PetscErrorCode DMPlexComputeCellGeometryFVM(DM, PetscInt, PetscReal *, PetscReal[], PetscReal[]);
PetscErrorCode DMPlexPermute(DM, IS, DM *);
PetscErrorCode PetscSFGetGraph(PetscSF, PetscInt *, PetscInt *, const PetscInt **, const PetscSFNode **);

PetscErrorCode PetscDeviceContextGetCurrentContext(PetscDeviceContext *);
PetscErrorCode PetscDeviceContextGetStreamHandle(PetscDeviceContext, void **);

PetscErrorCode TSGetTimeStep(TS, PetscReal *);
PetscErrorCode TSGetTime(TS, PetscReal *);
PetscErrorCode TSGetStepNumber(TS, PetscInt *);
PetscErrorCode TSGetSolution(TS, Vec *);
PetscErrorCode TSGetDM(TS, DM *);
PetscErrorCode TSGetApplicationContext(TS, void *);
PetscErrorCode TSPreStage(TS, PetscReal);
PetscErrorCode TSPostStage(TS, PetscReal, PetscInt, Vec *);
PetscErrorCode TSRegister(const char[], PetscErrorCode (*)(TS));
PetscErrorCode TSSetType(TS, TSType);
PetscErrorCode PetscNew_Lint(size_t, void *);
#define PetscNew(p) PetscNew_Lint(sizeof(**(p)), (p))
#endif
