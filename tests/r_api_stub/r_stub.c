/* tests/r_api_stub/r_stub.c -- TEST INFRASTRUCTURE, not R and not product code.
 *
 * A small stand-in for the part of R's C API that glue/cocons_hip_glue.c uses (the functions declared in
 * tests/r_api_decls/), with enough of R's semantics to EXECUTE the glue outside R: typed vectors with dim / names
 * attributes, external pointers with finalizers, the preserve list, the "not mutable" mark, R_alloc, and Rf_error as a
 * long jump back to the test driver (R's own Rf_error unwinds to the top level in the same way).  No garbage collector:
 * objects live until stub_reset() -- PROTECT / UNPROTECT only keep a balance that the driver checks after every call.
 * tests/test_glue_exec.py builds this file together with the glue into one shared object and drives it through ctypes.
 */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <unistd.h>
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#define NILSXP 0
#define CHARSXP 9
#define LGLSXP 10
#define STRSXP 16
#define EXTPTRSXP 22

struct SEXPREC {
    unsigned type;
    R_xlen_t len;
    int nrow, ncol, is_matrix;
    void *data;                 /* double[] / int[] / SEXP[] / char[] */
    SEXP names;                 /* STRSXP or NULL */
    void *ext_addr; SEXP ext_tag; R_CFinalizer_t fin;
    int preserved, not_mutable;
    struct SEXPREC *next;       /* allocation list */
};

static struct SEXPREC nil_rec = {NILSXP, 0, 0, 0, 0, NULL, NULL, NULL, NULL, NULL, 0, 0, NULL};
static struct SEXPREC names_sym = {NILSXP, 0, 0, 0, 0, NULL, NULL, NULL, NULL, NULL, 0, 0, NULL};
SEXP R_NilValue = &nil_rec, R_NamesSymbol = &names_sym;
double R_NaReal;

static SEXP all_objects = NULL;
static int protect_depth = 0;
static jmp_buf *err_jmp = NULL;
static char err_msg[1024];
static char **ralloc_list = NULL;
static size_t ralloc_n = 0, ralloc_cap = 0;

static SEXP new_obj(unsigned type, R_xlen_t len, size_t elt)
{
    SEXP s = (SEXP)calloc(1, sizeof *s);
    s->type = type; s->len = len;
    s->data = len > 0 && elt ? calloc((size_t)len, elt) : NULL;
    s->next = all_objects; all_objects = s;
    return s;
}

SEXP Rf_protect(SEXP s) { ++protect_depth; return s; }
void Rf_unprotect(int n) { protect_depth -= n; }
double *REAL(SEXP s) { if (s->type != REALSXP) Rf_error("REAL() of a non-double object"); return (double *)s->data; }
int *INTEGER(SEXP s) { if (s->type != INTSXP && s->type != LGLSXP) Rf_error("INTEGER() of a non-integer object"); return (int *)s->data; }
R_xlen_t XLENGTH(SEXP s) { return s->len; }
SEXP VECTOR_ELT(SEXP s, R_xlen_t i) { if (s->type != VECSXP || i < 0 || i >= s->len) Rf_error("VECTOR_ELT out of range"); return ((SEXP *)s->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP s, R_xlen_t i, SEXP v) { if (s->type != VECSXP || i < 0 || i >= s->len) Rf_error("SET_VECTOR_ELT out of range"); ((SEXP *)s->data)[i] = v; return v; }
SEXP STRING_ELT(SEXP s, R_xlen_t i) { if (s->type != STRSXP || i < 0 || i >= s->len) Rf_error("STRING_ELT out of range"); return ((SEXP *)s->data)[i]; }
const char *CHAR(SEXP s) { return (const char *)s->data; }

SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t len)
{
    SEXP s;
    if (type == REALSXP) s = new_obj(type, len, sizeof(double));
    else if (type == INTSXP || type == LGLSXP) s = new_obj(type, len, sizeof(int));
    else if (type == VECSXP || type == STRSXP) {
        s = new_obj(type, len, sizeof(SEXP));
        for (R_xlen_t i = 0; i < len; ++i) ((SEXP *)s->data)[i] = R_NilValue;
    } else { Rf_error("stub: Rf_allocVector of type %u", type); return R_NilValue; }
    return s;
}
SEXP Rf_allocMatrix(SEXPTYPE type, int nr, int nc)
{
    SEXP s = Rf_allocVector(type, (R_xlen_t)nr * nc);
    s->is_matrix = 1; s->nrow = nr; s->ncol = nc;
    return s;
}
SEXP Rf_ScalarReal(double v) { SEXP s = Rf_allocVector(REALSXP, 1); REAL(s)[0] = v; return s; }
SEXP Rf_ScalarInteger(int v) { SEXP s = Rf_allocVector(INTSXP, 1); INTEGER(s)[0] = v; return s; }
SEXP Rf_getAttrib(SEXP s, SEXP what) { return (what == R_NamesSymbol && s->names) ? s->names : R_NilValue; }
int Rf_nrows(SEXP s) { return s->is_matrix ? s->nrow : (int)s->len; }
int Rf_ncols(SEXP s) { return s->is_matrix ? s->ncol : 1; }
int Rf_asInteger(SEXP s) { return s->len < 1 ? 0 : (s->type == REALSXP ? (int)REAL(s)[0] : INTEGER(s)[0]); }
int Rf_asLogical(SEXP s) { return Rf_asInteger(s) != 0; }
double Rf_asReal(SEXP s) { return s->len < 1 ? R_NaReal : (s->type == REALSXP ? REAL(s)[0] : (double)INTEGER(s)[0]); }
Rboolean Rf_isMatrix(SEXP s) { return s->is_matrix ? TRUE : FALSE; }
Rboolean Rf_isInteger(SEXP s) { return s->type == INTSXP ? TRUE : FALSE; }
Rboolean Rf_isReal(SEXP s) { return s->type == REALSXP ? TRUE : FALSE; }
Rboolean Rf_isNull(SEXP s) { return s == R_NilValue ? TRUE : FALSE; }

void Rf_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_msg, sizeof err_msg, fmt, ap);
    va_end(ap);
    if (!err_jmp) { fprintf(stderr, "r_stub: Rf_error outside a call: %s\n", err_msg); abort(); }
    longjmp(*err_jmp, 1);
}

char *R_alloc(size_t n, int size)
{
    if (ralloc_n == ralloc_cap) {
        ralloc_cap = ralloc_cap ? 2 * ralloc_cap : 64;
        ralloc_list = (char **)realloc(ralloc_list, ralloc_cap * sizeof(char *));
    }
    char *p = (char *)calloc(n ? n : 1, (size_t)size);
    ralloc_list[ralloc_n++] = p;
    return p;
}

void *R_ExternalPtrAddr(SEXP s) { return s->type == EXTPTRSXP ? s->ext_addr : NULL; }
SEXP R_ExternalPtrTag(SEXP s) { return s->ext_tag ? s->ext_tag : R_NilValue; }
void R_ClearExternalPtr(SEXP s) { s->ext_addr = NULL; }
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot)
{
    (void)prot;
    SEXP s = new_obj(EXTPTRSXP, 0, 0);
    s->ext_addr = p; s->ext_tag = tag;
    return s;
}
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fin, Rboolean onexit) { (void)onexit; s->fin = fin; }
void R_PreserveObject(SEXP s) { s->preserved++; }
void R_ReleaseObject(SEXP s) { if (s->preserved <= 0) Rf_error("stub: R_ReleaseObject of an object that is not preserved"); s->preserved--; }
void (MARK_NOT_MUTABLE)(SEXP s) { s->not_mutable = 1; }

/* ---- registration --------------------------------------------------------------------------------------------------- */
static const R_CallMethodDef *registered = NULL;
static int dynamic_symbols = -1;
int R_registerRoutines(DllInfo *info, const R_CMethodDef *c, const R_CallMethodDef *call, const void *f, const void *e)
{
    (void)info; (void)c; (void)f; (void)e;
    registered = call;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean v) { (void)info; dynamic_symbols = (int)v; return TRUE; }

/* the glue is compiled with -Dgetpid=stub_getpid: the driver can make it believe it runs in another process (what a forked
 * worker of cocoOptim sees, R/optim.R:117-121) without forking a process that holds a HIP context */
#include <sys/syscall.h>
static int fake_pid = 0;
pid_t stub_getpid(void) { return fake_pid ? (pid_t)fake_pid : (pid_t)syscall(SYS_getpid); }
void stub_set_pid(int pid) { fake_pid = pid; }

/* ---- the driver's side (ctypes) -------------------------------------------------------------------------------------- */
void stub_init(void) { R_NaReal = NAN; }
SEXP stub_nil(void) { return R_NilValue; }
SEXP stub_real(long n) { return Rf_allocVector(REALSXP, n); }
SEXP stub_real_matrix(int nr, int nc) { return Rf_allocMatrix(REALSXP, nr, nc); }
SEXP stub_int(long n) { return Rf_allocVector(INTSXP, n); }
SEXP stub_list(long n) { return Rf_allocVector(VECSXP, n); }
void stub_list_set(SEXP lst, long i, SEXP v, const char *name)
{
    SET_VECTOR_ELT(lst, i, v);
    if (name) {
        if (!lst->names) lst->names = Rf_allocVector(STRSXP, lst->len);
        SEXP c = new_obj(CHARSXP, (R_xlen_t)strlen(name) + 1, 1);
        memcpy(c->data, name, strlen(name) + 1);
        ((SEXP *)lst->names->data)[i] = c;
    }
}
void *stub_data(SEXP s) { return s->data; }
long stub_len(SEXP s) { return (long)s->len; }
unsigned stub_type(SEXP s) { return s->type; }
int stub_nrow(SEXP s) { return Rf_nrows(s); }
int stub_ncol(SEXP s) { return Rf_ncols(s); }
SEXP stub_elt(SEXP s, long i) { return VECTOR_ELT(s, i); }
int stub_not_mutable(SEXP s) { return s->not_mutable; }
int stub_preserved(SEXP s) { return s->preserved; }
void *stub_extptr(SEXP s) { return R_ExternalPtrAddr(s); }
const char *stub_error(void) { return err_msg; }
int stub_protect_depth(void) { return protect_depth; }
int stub_dynamic_symbols(void) { return dynamic_symbols; }

/* what R does for `x[i] <- v` at top level: an object that may be shared (NAMED at its maximum / not mutable) is duplicated
 * first and the copy modified; one with a single reference is modified in place.  Returns the object that now holds the
 * modified value. */
SEXP stub_r_assign_real(SEXP x, long i, double v)
{
    SEXP t = x;
    if (x->not_mutable) {
        t = x->is_matrix ? Rf_allocMatrix(REALSXP, x->nrow, x->ncol) : Rf_allocVector(REALSXP, x->len);
        memcpy(t->data, x->data, (size_t)x->len * sizeof(double));
    }
    REAL(t)[i] = v;
    return t;
}

int stub_registered_arity(const char *name)
{
    for (const R_CallMethodDef *d = registered; d && d->name; ++d)
        if (strcmp(d->name, name) == 0) return d->numArgs;
    return -1;
}

/* .Call: looks the symbol up in the table R_init_cocons registered, checks the arity like R does, runs it under the error
 * handler.  Returns NULL when the callee raised an R error (message: stub_error()). */
SEXP stub_dot_call(const char *name, int nargs, SEXP *a)
{
    const R_CallMethodDef *volatile d = registered;
    for (; d && d->name; ++d)
        if (strcmp(d->name, name) == 0) break;
    if (!d || !d->name) { snprintf(err_msg, sizeof err_msg, "no such registered routine: %s", name); return NULL; }
    if (d->numArgs != nargs) { snprintf(err_msg, sizeof err_msg, "%s takes %d arguments, %d given", name, d->numArgs, nargs); return NULL; }
    jmp_buf jb;
    err_jmp = &jb;
    err_msg[0] = 0;
    const int depth0 = protect_depth;
    SEXP volatile out = NULL;
    if (setjmp(jb) == 0) {
        DL_FUNC f = d->fun;
        switch (nargs) {
        case 0: out = ((SEXP (*)(void))f)(); break;
        case 1: out = ((SEXP (*)(SEXP))f)(a[0]); break;
        case 2: out = ((SEXP (*)(SEXP, SEXP))f)(a[0], a[1]); break;
        case 3: out = ((SEXP (*)(SEXP, SEXP, SEXP))f)(a[0], a[1], a[2]); break;
        case 4: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP))f)(a[0], a[1], a[2], a[3]); break;
        case 5: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP))f)(a[0], a[1], a[2], a[3], a[4]); break;
        case 6: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))f)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
        case 8: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
        case 9: out = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))f)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
        default: snprintf(err_msg, sizeof err_msg, "stub: %d arguments not supported", nargs); out = NULL;
        }
        if (out && protect_depth != depth0) { snprintf(err_msg, sizeof err_msg, "%s left the protect stack unbalanced (%d)", name, protect_depth - depth0); out = NULL; }
    } else {
        protect_depth = depth0;           /* R unwinds the protect stack on an error */
        out = NULL;
    }
    err_jmp = NULL;
    for (size_t i = 0; i < ralloc_n; ++i) free(ralloc_list[i]);      /* R_alloc memory lives until the .Call returns */
    ralloc_n = 0;
    return out;
}

/* "garbage collection": every object that is neither preserved nor listed in keep[] is freed, external pointers through
 * their finalizers first.  Returns the number of finalizers run. */
int stub_gc(int nkeep, SEXP *keep)
{
    int ran = 0;
    for (SEXP s = all_objects; s; s = s->next) {
        int kept = s->preserved > 0;
        for (int k = 0; k < nkeep && !kept; ++k) kept = keep[k] == s;
        if (!kept && s->type == EXTPTRSXP && s->fin && s->ext_addr) { s->fin(s); ++ran; }
    }
    return ran;
}
