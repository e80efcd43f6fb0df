/* LINT ONLY (see ../../petsc_decls.h): the members of PETSc's private TS struct that a TS implementation touches
 * (src/ts/impls/explicit/euler/euler.c is the model) */
#include "../../petsc_decls.h"
struct _TSOps {
  PetscErrorCode (*setup)(TS);
  PetscErrorCode (*step)(TS);
  PetscErrorCode (*reset)(TS);
  PetscErrorCode (*destroy)(TS);
  PetscErrorCode (*interpolate)(TS, PetscReal, Vec);
};
struct _p_TS {
  struct _TSOps     ops[1];
  void             *data;
  Vec               vec_sol;
  PetscReal         time_step, ptime;
  PetscInt          steps;
  TSConvergedReason reason;
};
