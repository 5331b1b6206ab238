/* syntax-check stand-in (tests/r_stub/README): the parts of <Rinternals.h> gpirt_shim.c uses */
#ifndef R_STUB_RINTERNALS_H
#define R_STUB_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
typedef enum { FALSE = 0, TRUE } Rboolean;
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NilValue, R_DimSymbol, R_NamesSymbol, R_GlobalEnv;
extern int R_NaInt;
#define NA_INTEGER R_NaInt
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
SEXP Rf_coerceVector(SEXP, SEXPTYPE);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_alloc3DArray(SEXPTYPE, int, int, int);
SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_install(const char*);
SEXP Rf_GetOption1(SEXP);
SEXP Rf_findVarInFrame(SEXP, SEXP);
SEXP Rf_mkChar(const char*);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
Rboolean Rf_isString(SEXP);
int TYPEOF(SEXP);
int LENGTH(SEXP);
R_xlen_t XLENGTH(SEXP);
int* INTEGER(SEXP);
double* REAL(SEXP);
const char* CHAR(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
void Rf_onintr(void);
#define coerceVector Rf_coerceVector
#define getAttrib Rf_getAttrib
#define setAttrib Rf_setAttrib
#define allocMatrix Rf_allocMatrix
#define alloc3DArray Rf_alloc3DArray
#define allocVector Rf_allocVector
#define install Rf_install
#define GetOption1 Rf_GetOption1
#define findVarInFrame Rf_findVarInFrame
#define mkChar Rf_mkChar
#define asInteger Rf_asInteger
#define asLogical Rf_asLogical
#define isString Rf_isString
#endif
