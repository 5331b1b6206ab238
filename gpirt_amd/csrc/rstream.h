// rstream.h -- R's default RNG (Mersenne-Twister + inversion) on the host; product-side code.
#pragma once
#include "common.h"

namespace gpirt {

// ---------------------------------------------------------------- R's RNG (host) -----------
// Mersenne-Twister + inversion exactly as R's default generator (RNG.c: MT_genrand, fixup,
// Randomize; snorm.c INVERSION; qnorm.c AS241).  Call sites replaced: R/gpirtMCMC.R:96 and every
// R::rnorm / R::runif of the reference's src/.
struct RStream {
    uint32_t mt[624];
    int mti;
    uint32_t next32()
    {
        enum { N = 624, M = 397 };
        if (mti >= N) {
            int kk;
            uint32_t y;
            for (kk = 0; kk < N - M; ++kk) {
                y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            for (; kk < N - 1; ++kk) {
                y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
                mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            y = (mt[N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            mti = 0;
        }
        uint32_t y = mt[mti++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double unif()
    {
        const double i2_32m1 = 2.328306437080797e-10;
        double v = (double)next32() * 2.3283064365386963e-10;
        if (v <= 0.0) return 0.5 * i2_32m1;
        if ((1.0 - v) <= 0.0) return 1.0 - 0.5 * i2_32m1;
        return v;
    }
    double norm()
    {
        double u1 = unif();
        double u2 = unif();
        return rnorm_from_two(u1, u2);
    }
};

}  // namespace gpirt

struct gpirt_rstream_s { gpirt::RStream r; };
