/* syntax-check stand-in (tests/r_stub/README) */
#ifndef R_STUB_RANDOM_H
#define R_STUB_RANDOM_H
void GetRNGstate(void);
void PutRNGstate(void);
double unif_rand(void);
#endif
