/* LINT ONLY (see ../petsc_decls.h): the libCEED handle types RDycore's private headers mention */
#ifndef RDYHIP_LINT_CEED_H
#define RDYHIP_LINT_CEED_H
typedef struct Ceed_private                 *Ceed;
typedef struct CeedVector_private           *CeedVector;
typedef struct CeedOperator_private         *CeedOperator;
typedef struct CeedQFunction_private        *CeedQFunction;
typedef struct CeedQFunctionContext_private *CeedQFunctionContext;
typedef struct CeedElemRestriction_private  *CeedElemRestriction;
typedef struct CeedBasis_private            *CeedBasis;
typedef int    CeedInt;
typedef double CeedScalar;
typedef int    CeedMemType;
typedef long   CeedSize;
#define CEED_QFUNCTION(name) static int name
#define CEED_QFUNCTION_HELPER static inline
#endif
