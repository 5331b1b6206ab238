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
    void refresh()
    {
        enum { N = 624, M = 397 };
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
    static uint32_t temper(uint32_t y)
    {
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    uint32_t next32()
    {
        if (mti >= 624) refresh();
        return temper(mt[mti++]);
    }
    static double fixup(uint32_t y)
    {
        const double i2_32m1 = 2.328306437080797e-10;
        const double v = (double)y * 2.3283064365386963e-10;
        return v <= 0.0 ? 0.5 * i2_32m1 : ((1.0 - v) <= 0.0 ? 1.0 - 0.5 * i2_32m1 : v);
    }
    // `count` consecutive unif() values, a state block at a time (the tempering loop has no carried dependence: it vectorises)
    void fill_unif(double* dst, uint64_t count)
    {
        while (count) {
            if (mti >= 624) refresh();
            const uint64_t left = (uint64_t)(624 - mti);
            const int take = (int)(left < count ? left : count);
            const uint32_t* src = mt + mti;
            for (int q = 0; q < take; ++q) dst[q] = fixup(temper(src[q]));
            mti += take; dst += take; count -= (uint64_t)take;
        }
    }
    // the same draws as raw state words: unif() = fixup(temper(word)), left to the consumer (the device, rs_unpack_kernel)
    void fill_raw(uint32_t* dst, uint64_t count)
    {
        while (count) {
            if (mti >= 624) refresh();
            const uint64_t left = (uint64_t)(624 - mti);
            const uint64_t take = left < count ? left : count;
            memcpy(dst, mt + mti, (size_t)take * sizeof(uint32_t));
            mti += (int)take; dst += take; count -= take;
        }
    }
    // the state `count` draws further on, without producing them
    void skip(uint64_t count)
    {
        while (count) {
            if (mti >= 624) refresh();
            const uint64_t left = (uint64_t)(624 - mti);
            const uint64_t take = left < count ? left : count;
            mti += (int)take; count -= take;
        }
    }
    double unif() { return fixup(next32()); }
    double norm()
    {
        double u1 = unif();
        double u2 = unif();
        return rnorm_from_two(u1, u2);
    }
};

}  // namespace gpirt

// owner / resolve: a sampler that replays this stream keeps it AHEAD of what the chain has consumed (uniforms generated
// while the device works, sampler.hip: stream_begin / stream_end); whoever looks at the state, draws from it or destroys
// it calls rstream_sync first, which puts the state back to exactly the consumed position and detaches the sampler.
// attached / forget: EVERY sampler created on this stream, running ahead or not (before its first step, after an error
// handed the generator back, after another sampler took it over): gpirt_rstream_destroy tells each of them that the
// stream is gone, so that none is left with a dangling pointer (a later step then fails with "the R stream of this sampler
// has been destroyed" instead of reading freed memory).
struct gpirt_rstream_s {
    gpirt::RStream r;
    void* owner = nullptr;
    void (*resolve)(void* owner, bool gone) = nullptr;
    std::vector<void*> attached;
    void (*forget)(void* sampler) = nullptr;
};
inline void rstream_sync(gpirt_rstream_s* r, bool gone = false) { if (r && r->owner) r->resolve(r->owner, gone); }
inline void rstream_gone(gpirt_rstream_s* r)
{
    if (!r) return;
    rstream_sync(r, true);
    if (r->forget) for (void* s : r->attached) r->forget(s);
    r->attached.clear();
}
