/* syntax-check stand-in (tests/r_stub/README) */
#ifndef R_STUB_UTILS_H
#define R_STUB_UTILS_H
#include <Rinternals.h>
void R_CheckUserInterrupt(void);
Rboolean R_ToplevelExec(void (*fun)(void*), void* data);
#endif
