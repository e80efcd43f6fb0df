#include "petsc_decls.h"
