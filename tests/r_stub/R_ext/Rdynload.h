/* syntax-check stand-in (tests/r_stub/README) */
#ifndef R_STUB_RDYNLOAD_H
#define R_STUB_RDYNLOAD_H
#include <Rinternals.h>
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo*, const void*, const R_CallMethodDef*, const void*, const void*);
Rboolean R_useDynamicSymbols(DllInfo*, Rboolean);
#endif
