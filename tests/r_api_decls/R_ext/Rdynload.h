/* TEST INFRASTRUCTURE: see ../Rinternals.h */
#ifndef COCONS_TEST_RDYNLOAD_H
#define COCONS_TEST_RDYNLOAD_H
#include "../Rinternals.h"
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
typedef struct { const char *name; DL_FUNC fun; int numArgs; void *types; } R_CMethodDef;
int R_registerRoutines(DllInfo *, const R_CMethodDef *, const R_CallMethodDef *, const void *, const void *);
Rboolean R_useDynamicSymbols(DllInfo *, Rboolean);
#endif
