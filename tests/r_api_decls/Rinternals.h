/* TEST INFRASTRUCTURE, not R: prototypes of the part of R's public C API (R >= 3.4, Rinternals.h / R_ext/Rdynload.h as
 * documented in "Writing R Extensions", section 5) that glue/cocons_hip_glue.c uses -- declarations only, no definitions.
 * R is not installed in the build image, so the glue can never be LINKED here; with these declarations
 * tests/test_abi.py::test_glue_compiles_against_declared_apis runs the compiler's front end over it (-fsyntax-only,
 * -Werror), which checks every call of a cocons_* function against include/cocons_hip.h (argument count and types) and
 * every use of the R API against the documented signatures. */
#ifndef COCONS_TEST_RINTERNALS_H
#define COCONS_TEST_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef enum { FALSE = 0, TRUE } Rboolean;
typedef unsigned int SEXPTYPE;
#define INTSXP 13
#define REALSXP 14
#define VECSXP 19
extern SEXP R_NilValue, R_NamesSymbol;
extern double R_NaReal;            /* R_ext/Arith.h */
#define NA_REAL R_NaReal
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
double *REAL(SEXP);
int *INTEGER(SEXP);
R_xlen_t XLENGTH(SEXP);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
const char *CHAR(SEXP);
SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_ScalarReal(double);
SEXP Rf_ScalarInteger(int);
SEXP Rf_getAttrib(SEXP, SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
double Rf_asReal(SEXP);
Rboolean Rf_isMatrix(SEXP);
Rboolean Rf_isInteger(SEXP);
Rboolean Rf_isReal(SEXP);
Rboolean Rf_isNull(SEXP);
void Rf_error(const char *, ...) __attribute__((noreturn));
char *R_alloc(size_t, int);
void *R_ExternalPtrAddr(SEXP);
SEXP R_ExternalPtrTag(SEXP);
void R_ClearExternalPtr(SEXP);
SEXP R_MakeExternalPtr(void *, SEXP, SEXP);
typedef void (*R_CFinalizer_t)(SEXP);
void R_RegisterCFinalizerEx(SEXP, R_CFinalizer_t, Rboolean);
void R_PreserveObject(SEXP);
void R_ReleaseObject(SEXP);
void (MARK_NOT_MUTABLE)(SEXP);     /* R >= 3.5: NAMED / reference count to its maximum -- R code must duplicate before modifying */
#endif
