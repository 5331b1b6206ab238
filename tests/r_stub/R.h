/* syntax-check stand-in (tests/r_stub/README): the parts of <R.h> gpirt_shim.c uses */
#ifndef R_STUB_R_H
#define R_STUB_R_H
#include <stddef.h>
#include <stdint.h>
void Rprintf(const char*, ...);
void Rf_error(const char*, ...);
#define error Rf_error
#endif
