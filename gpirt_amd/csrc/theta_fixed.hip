// theta_fixed.hip -- draw_theta's log-posterior product (src/draw-theta.cpp:15-19) in EXACT fixed point on the int8 matrix cores.
//
//   logpost[g, i] = sum_j [y_ij = +1] G+[g, j] + [y_ij = -1] G-[g, j],   G+- = -log(1 + exp(-+ f*[g, j]))          (N x n, N = 1001)
//
// is a product with a 0/1 operand: the only rounding an fp64 GEMM commits in it is in the ADDITIONS (2m of them per entry,
// each rounded at the size of the running sum).  Fixed point has none: every row g of G is scaled by a power of two
// 2^(54 - e_g) (2^e_g above everything the row holds) and rounded ONCE to a 54-bit integer, that integer is cut into seven
// balanced base-256 digits (signed bytes), and
//         sum_j Y[i, j] q[g, j] = sum_s 256^s * (sum_j Y[i, j] d_s[g, j])
// is seven int8 products with int32 accumulators, exact (|d| <= 128, <= 2^11 non-zero terms per sum: 2^18), recombined in
// two int64 halves and one fp64 addition.  What comes out is the correctly rounded sum of the once-rounded terms: the error
// per entry is <= m 2^(e_g - 55) in the worst case (4.5e-13 for m = 1024 and a row maximum below 16; ~1e-14 typical) against
// <= m^2 u max|G| for the fp64 chain -- tests/test_theta_fixed_scheme.py restates the scheme in Python integers,
// tests/test_gpu_theta_fixed.py measures both products against long double and holds the fixed-point form to the SMALLER error.  It is also independent of the order of the items, so a respondent block of a
// sharded run (do_theta_block) is bit-identical to the single-GPU product by construction.
//
// v_mfma_i32_32x32x32_i8 runs at 32x the fp64 MFMA's rate per clock, so seven digit planes cost less than a quarter of the
// one fp64 product (0.54 ms at 8192 x 1024 -- already 91% of the fp64 MFMA peak, nothing left to tune there).
//
// When a row cannot be scaled -- |f*| beyond exp()'s range (the formula as written then yields -inf, stages.hip
// loglik_terms_kernel) or a non-finite f* -- the quantiser raises a device-side flag: the int8 kernel returns at once and the
// fp64 product, launched behind it under that flag, runs instead (and only then): same results as before this file existed.
//
// Layouts (K = the 2m indicator columns, padded: plus part [0, mp), minus part [mp, 2 mp), mp = m rounded up to 16, then zeros
// up to a whole number of chunks).  The MFMA sums over k, exactly, so the order of k inside an instruction is free; both
// operands are stored in "fragment order", the 1 KiB one wave instruction consumes:
//     Y8 [i / 32][k / 32][lane = (i % 32) + 32 ((k % 32) / 16)][k % 16]           bytes 0 / 1      (built once: y is data)
//     Gq [digit s][g / 32][k / 32][lane = (g % 32) + 32 ((k % 32) / 16)][k % 16]   signed bytes     (every iteration)
// so global -> LDS is a lane-linear 16-byte copy and LDS -> operand one conflict-free ds_read_b128.  A work-group owns 256
// respondents x 32 grid points: wave w the respondent blocks 2 w, 2 w + 1 with all seven digit planes = 14 accumulator tiles
// (224 registers, one work-group per CU).  Only the digit planes are shared between the waves and go through the LDS (7 reads
// per k-step and wave for 14 MFMAs); a wave's indicators go from global memory straight into its registers.  Work-groups are
// dealt so that an XCD keeps four grid blocks (their 1.8 MB of Gq stay in its L2) and streams the respondents.
#include "kernels.h"

namespace gpirt {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int TF_BITS = 54;          // bits of a row's fixed-point terms (the top digit then stays below 65 + a carry)
constexpr int TF_DIGITS = 7;
constexpr int TF_KS = 4;             // k-steps (of 32) per LDS stage
constexpr int TF_STAGE = TF_DIGITS * TF_KS * 1024;                       // the seven digit planes of one grid block, bytes
constexpr int TF_TRACE_WGS = 1024;   // work-groups that leave stamps in a traced launch
static_assert((TF_DIGITS * TF_KS) % 4 == 0 && TF_KS % 2 == 0, "the pieces of a stage are dealt to four waves in whole rounds");
static_assert(2 * TF_STAGE <= 64 * 1024, "two stages in the static LDS allowance");

__device__ __forceinline__ uint32_t pack4(int b0, int b1, int b2, int b3)
{
    return (uint32_t)(b0 & 255) | ((uint32_t)(b1 & 255) << 8) | ((uint32_t)(b2 & 255) << 16) | ((uint32_t)(b3 & 255) << 24);
}

// the item and the sign an indicator column k stands for; false: padding
__device__ __forceinline__ bool tf_column(int64_t k, int64_t m, int64_t mp, int64_t* j, double* sign)
{
    if (k < mp) { *j = k; *sign = 1.0; return k < m; }
    if (k < 2 * mp) { *j = k - mp; *sign = -1.0; return k - mp < m; }
    return false;
}

// Y8: one thread per (respondent, half k-step) = 16 bytes
__global__ __launch_bounds__(256) void tf_y8_kernel(const double* __restrict__ y, int64_t n, int64_t ldy, int64_t m, TfDims d,
                                                    uint4* __restrict__ Y8)
{
    const int64_t total = d.iblocks * d.ksteps * 64;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int lane = (int)(t & 63);
        const int64_t ks = (t >> 6) % d.ksteps, ib = (t >> 6) / d.ksteps;
        const int64_t i = ib * 32 + (lane & 31);
        int b[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int64_t j; double sign;
            const bool real = tf_column(ks * 32 + 16 * (lane >> 5) + q, m, d.mp, &j, &sign);
            b[q] = (real && i < n && y[i + j * ldy] == sign) ? 1 : 0;                  // (NaN compares false: skipped, :16)
        }
        Y8[t] = make_uint4(pack4(b[0], b[1], b[2], b[3]), pack4(b[4], b[5], b[6], b[7]), pack4(b[8], b[9], b[10], b[11]),
                           pack4(b[12], b[13], b[14], b[15]));
    }
}

// max_j |f*[g, j]| as the bits of a non-negative double (they order like the values; NaN sorts above everything)
__global__ __launch_bounds__(256) void tf_rowmax_kernel(const double* __restrict__ fstar, int64_t N, int64_t m,
                                                        unsigned long long* __restrict__ amax)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= N) return;
    const int64_t j0 = (int64_t)blockIdx.y * 16, j1 = (j0 + 16 < m) ? j0 + 16 : m;
    unsigned long long best = 0;
    for (int64_t j = j0; j < j1; ++j) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(fabs(fstar[g + j * N]));
        best = bits > best ? bits : best;
    }
    atomicMax(amax + g, best);
}

// one thread per (grid point, half k-step): 16 terms -> 7 x 16 digit bytes
__global__ __launch_bounds__(256) void tf_quant_kernel(const double* __restrict__ fstar, int64_t N, int64_t m, TfDims d,
                                                       const unsigned long long* __restrict__ amax, uint4* __restrict__ Gq,
                                                       double* __restrict__ scale, int* __restrict__ ovf)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t kg = blockIdx.y;                          // half k-step
    if (g >= d.gblocks * 32) return;
    const int64_t ks = kg >> 1;
    const int lane = (int)(g & 31) + 32 * (int)(kg & 1);
    const int64_t piece = ((g >> 5) * d.ksteps + ks) * 64 + lane;           // within one digit plane
    const int64_t plane = d.gblocks * d.ksteps * 64;
    int64_t q[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) q[t] = 0;
    if (g < N) {
        const double fmax = __longlong_as_double((long long)amax[g]);
        // log(1 + exp(x)) <= x + log 2: 2^e is above every term of the row.  Beyond exp()'s range the formula as written
        // overflows to -inf (and NaN / inf are not numbers to scale): leave the product to the fp64 form
        const bool ok = fmax <= 709.0;
        if (!ok) { if (kg == 0) atomicOr(ovf, 1); }
        const int e = ok ? ilogb(fmax + 0.6931471805599453) + 1 : 0;
        if (kg == 0) scale[g] = ok ? ldexp(1.0, e - TF_BITS) : 0.0;
        if (ok) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                int64_t j; double sign;
                if (tf_column(kg * 16 + t, m, d.mp, &j, &sign)) {
                    const double term = ll_term(sign * fstar[g + j * N]);              // = -G+ / -G- (stages.hip), >= 0
                    q[t] = (int64_t)rint(ldexp(term, TF_BITS - e));
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < TF_DIGITS; ++s) {
        int b[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int dg = (s + 1 < TF_DIGITS) ? (int)(int8_t)(q[t] & 255) : (int)q[t];    // balanced digit; the last takes the rest
            b[t] = dg;
            q[t] = (q[t] - dg) >> 8;
        }
        Gq[(int64_t)s * plane + piece] = make_uint4(pack4(b[0], b[1], b[2], b[3]), pack4(b[4], b[5], b[6], b[7]),
                                                    pack4(b[8], b[9], b[10], b[11]), pack4(b[12], b[13], b[14], b[15]));
    }
}

struct TfMfmaArgs {
    const unsigned char* Y8; const unsigned char* Gq; const double* scale; const int* ovf;
    double* logpost; int64_t ldlp; int64_t n, N; int64_t ksteps, gblocks; int itiles;
    long long* trace;     // debug (gpirt_debug_theta_clock): six stamps per work-group, shader clock and 100 MHz wall clock
};

// A work-group owns 256 respondents x 32 grid points; wave w the respondent blocks 2 w, 2 w + 1 with all seven digit planes
// (14 accumulator tiles).  The digit planes of the grid block are shared by the four waves and go through the LDS, a
// respondent block's indicators are used by ONE wave and go from global memory straight into its registers; both one chunk
// (TF_KS k-steps) ahead.  Everything a chunk issues besides its 56 MFMAs sits in the gaps between them, in a pinned order
// (one wave per SIMD: nothing else hides a latency, and a burst of loads ahead of the MFMAs costs more than the MFMAs did --
// 143 us with LDS-DMA pieces issued at the top of each chunk, 105 us like this; tools/theta_clock.py: 2400 shader cycles
// per chunk against the MFMAs' 1792 at the 1.8 GHz the chip holds under this kernel -- taking out the digit-plane staging
// gives back 210 of them, the indicator loads 95, the barrier 50):
//     k-step 0   operands of k-step 1 (LDS)    the wave's 7 pieces of the next chunk's digit planes (global -> registers)
//     k-step 1   operands of k-step 2          the next chunk's indicators, 8 loads
//     k-step 2   operands of k-step 3          the 7 pieces registers -> the other LDS stage
//     k-step 3   6 MFMAs (the LDS traffic above completes under them), barrier (every wave's pieces are in, everyone is done
//                with this stage), the next chunk's first operands in one burst under the other 8 MFMAs
// Chunks are taken in pairs so that both register sets are indexed statically; past the last chunk the loads repeat it
// (into the stage and the registers nobody reads any more) instead of branching.
template <int P>
__device__ __forceinline__ void tf_chunk(const TfMfmaArgs& a, unsigned char* lds, int c, int nchunks, int lane, int wave,
                                         const unsigned char* srcA0, const unsigned char* srcA1, const unsigned char* srcB,
                                         v4i (&ya)[2][TF_KS][2], v4i (&fb)[2][TF_DIGITS], v16i (&acc)[2][TF_DIGITS])
{
    constexpr int KS = TF_KS;
    static_assert(KS == 4, "the schedule below is written out for four k-steps");
    constexpr int NP = (TF_DIGITS * KS) / 4;                // pieces per wave
    const int64_t ksteps = a.ksteps;
    const int cn = (c + 1 < nchunks) ? c + 1 : c;
    const int64_t adv = (int64_t)cn * KS * 1024;
    const unsigned char* sb = lds + P * TF_STAGE + lane * 16;
    unsigned char* sn = lds + (P ^ 1) * TF_STAGE + lane * 16;
    v4i bs[NP];
    auto mfma = [&](int ksl, int slot, int s0, int s1) {
#pragma unroll
        for (int s = s0; s < s1; ++s) {
            acc[0][s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ya[P][ksl][0], fb[slot][s], acc[0][s], 0, 0, 0);
            acc[1][s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ya[P][ksl][1], fb[slot][s], acc[1][s], 0, 0, 0);
        }
    };
    auto fetch = [&](const unsigned char* stage, int ksl, int slot) {
#pragma unroll
        for (int s = 0; s < TF_DIGITS; ++s) fb[slot][s] = *reinterpret_cast<const v4i*>(stage + (s * KS + ksl) * 1024);
    };
    // k-step 0
    fetch(sb, 1, 1);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pb = 4 * q + wave, ksl = pb % KS, s = pb / KS;
        bs[q] = *reinterpret_cast<const v4i*>(srcB + ((int64_t)s * a.gblocks * ksteps + ksl) * 1024 + adv);
    }
    mfma(0, 0, 0, TF_DIGITS);
#pragma unroll
    for (int r = 0; r < TF_DIGITS; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    // k-step 1
    fetch(sb, 2, 0);
#pragma unroll
    for (int ksl = 0; ksl < KS; ++ksl) {
        ya[P ^ 1][ksl][0] = *reinterpret_cast<const v4i*>(srcA0 + adv + ksl * 1024);
        ya[P ^ 1][ksl][1] = *reinterpret_cast<const v4i*>(srcA1 + adv + ksl * 1024);
    }
    mfma(1, 1, 0, TF_DIGITS);
#pragma unroll
    for (int r = 0; r < TF_DIGITS; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    // k-step 2
    fetch(sb, 3, 1);
#pragma unroll
    for (int q = 0; q < NP; ++q) *reinterpret_cast<v4i*>(sn + (4 * q + wave) * 1024) = bs[q];
    mfma(2, 0, 0, TF_DIGITS);
#pragma unroll
    for (int r = 0; r < TF_DIGITS; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int r = 0; r < TF_DIGITS; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    // k-step 3
    mfma(3, 1, 0, 3);
    __builtin_amdgcn_sched_barrier(0);                      // (the six MFMAs stay AHEAD of the barrier's wait for the LDS traffic)
    __syncthreads();
    fetch(sn, 0, 0);
    mfma(3, 1, 3, TF_DIGITS);
    __builtin_amdgcn_sched_group_barrier(0x100, TF_DIGITS, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
}

__global__ __launch_bounds__(256, 1) void tf_mfma_kernel(TfMfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_lds[];
    if (*a.ovf) return;
    constexpr int KS = TF_KS;
    // an XCD (work-group id mod 8) keeps the grid blocks 4 x .. 4 x + 3 (their digit planes, 1.8 MB at m = 1024, stay in its
    // L2) and walks the respondent tiles
    const int id = (int)blockIdx.x, xcd = id & 7, local = id >> 3;
    const int gb = 4 * xcd + (local & 3), it = local >> 2;
    const int lane = (int)(threadIdx.x & 63);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t ksteps = a.ksteps;
    const int nchunks = (int)(ksteps / KS);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const unsigned char* srcA0 = a.Y8 + ((int64_t)(it * 8 + 2 * wave) * ksteps) * 1024 + lane16;
    const unsigned char* srcA1 = srcA0 + ksteps * 1024;
    const unsigned char* srcB = a.Gq + ((int64_t)gb * ksteps) * 1024 + lane16;
    long long* tr = (a.trace && blockIdx.x < TF_TRACE_WGS && threadIdx.x == 0) ? a.trace + 6 * blockIdx.x : nullptr;
    if (tr) { tr[0] = (long long)__builtin_amdgcn_s_memtime(); tr[1] = (long long)__builtin_amdgcn_s_memrealtime(); }

    v16i acc[2][TF_DIGITS];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < TF_DIGITS; ++s)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][s][v] = 0;
    v4i ya[2][KS][2], fb[2][TF_DIGITS];

    // chunk 0: as tf_chunk sends chunk c + 1
#pragma unroll
    for (int q = 0; q < (TF_DIGITS * KS) / 4; ++q) {
        const int pb = 4 * q + wave, ksl = pb % KS, s = pb / KS;
        *reinterpret_cast<v4i*>(tf_lds + lane16 + pb * 1024) =
            *reinterpret_cast<const v4i*>(srcB + ((int64_t)s * a.gblocks * ksteps + ksl) * 1024);
    }
#pragma unroll
    for (int ksl = 0; ksl < KS; ++ksl) {
        ya[0][ksl][0] = *reinterpret_cast<const v4i*>(srcA0 + ksl * 1024);
        ya[0][ksl][1] = *reinterpret_cast<const v4i*>(srcA1 + ksl * 1024);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TF_DIGITS; ++s) fb[0][s] = *reinterpret_cast<const v4i*>(tf_lds + lane16 + (s * KS) * 1024);

    for (int c = 0; c < nchunks; c += 2) {                  // (tf_dims: an even number of chunks)
        tf_chunk<0>(a, tf_lds, c, nchunks, lane, wave, srcA0, srcA1, srcB, ya, fb, acc);
        tf_chunk<1>(a, tf_lds, c + 1, nchunks, lane, wave, srcA0, srcA1, srcB, ya, fb, acc);
    }
    if (tr) { tr[2] = (long long)__builtin_amdgcn_s_memtime(); tr[3] = (long long)__builtin_amdgcn_s_memrealtime(); }

    // register v of lane l: respondent (v & 3) + 8 (v >> 2) + 4 (l >> 5) of the block, grid point l & 31
    const int64_t g = (int64_t)gb * 32 + (lane & 31);
    const double sc = g < a.N ? a.scale[g] : 0.0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (g >= a.N) break;
        const int64_t i0 = (int64_t)it * 256 + (2 * wave + r) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int64_t i = i0 + (v & 3) + 8 * (v >> 2);
            if (i >= a.n) continue;
            const int64_t hi = (((int64_t)acc[r][6][v] * 256 + acc[r][5][v]) * 256 + acc[r][4][v]) * 256 + acc[r][3][v];
            const int64_t lo = ((int64_t)acc[r][2][v] * 256 + acc[r][1][v]) * 256 + acc[r][0][v];
            // |hi| < 2^45, |lo| < 2^36: both exact in fp64; one rounding in the addition, the scale is a power of two
            a.logpost[g + i * a.ldlp] = -(((double)hi * 16777216.0 + (double)lo) * sc);
        }
    }
    if (tr) { tr[4] = (long long)__builtin_amdgcn_s_memtime(); tr[5] = (long long)__builtin_amdgcn_s_memrealtime(); }
}

}  // namespace

TfDims tf_dims(int64_t n, int64_t m, int64_t N)
{
    TfDims d;
    d.mp = (m + 15) / 16 * 16;
    const int64_t steps = (2 * d.mp + 31) / 32;
    d.ksteps = (steps + 2 * TF_KS - 1) / (2 * TF_KS) * (2 * TF_KS);      // an even number of chunks (tf_mfma_kernel)
    if (d.ksteps == 0) d.ksteps = 2 * TF_KS;
    d.iblocks = (n + 255) / 256 * 8;
    d.gblocks = (N + 31) / 32;
    return d;
}
size_t tf_y8_bytes(const TfDims& d) { return (size_t)d.iblocks * d.ksteps * 1024; }
size_t tf_gq_bytes(const TfDims& d) { return (size_t)TF_DIGITS * d.gblocks * d.ksteps * 1024; }
// amax (gblocks * 32 u64) | scale (gblocks * 32 doubles) | ovf (int, padded to 16 bytes) | debug stamps
size_t tf_aux_bytes(const TfDims& d) { return (size_t)d.gblocks * 32 * 16 + 16 + (size_t)TF_TRACE_WGS * 6 * 8; }
long long* tf_trace(void* aux, const TfDims& d) { return reinterpret_cast<long long*>(reinterpret_cast<unsigned char*>(aux) + (size_t)d.gblocks * 32 * 16 + 16); }
int tf_trace_wgs() { return TF_TRACE_WGS; }

int launch_tf_indicators(hipStream_t stream, const double* y, int64_t n, int64_t ldy, int64_t m, const TfDims& d, void* Y8)
{
    const int64_t total = d.iblocks * d.ksteps * 64;
    if (total <= 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(tf_y8_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, y, n, ldy, m, d, reinterpret_cast<uint4*>(Y8));
    GP_HIP(hipGetLastError());
    return 0;
}

// logpost (N x n, leading dimension ldlp) from f* (N x m) and the prepared indicators; *tf_overflow(aux) != 0 afterwards means
// the product was NOT formed (see the header): the caller's fp64 product, launched under that flag, does it instead
int launch_theta_fixed(hipStream_t stream, const double* fstar, int64_t N, int64_t n, int64_t m, const TfDims& d,
                       const void* Y8, void* Gq, void* aux, double* logpost, int64_t ldlp, bool trace, gpirt_handle_t prof)
{
    if (n <= 0 || N <= 0) return 0;
    // Every refusal BEFORE anything is enqueued.  The recombination's int64 halves are exact doubles while a plane sum stays
    // below 2^27 (a million items); the quantiser's grid carries two entries per k-step in grid.y, whose limit is 65535
    // launches: 2 * ksteps <= 65535 (m <= ~524 000 items) is the bound that bites first.
    if (m > (1 << 20) || 2 * d.ksteps > 65535) { set_error("theta_fixed: m = %lld is beyond the fixed-point product's range (GPIRT_THETA_FIXED=2 selects the fp64 product)", (long long)m); return GPIRT_E_ARG; }
    if (d.gblocks != 32) { set_error("theta_fixed: the work-group map is laid out for the reference's 1001-point grid"); return GPIRT_E_ARG; }
    unsigned long long* amax = reinterpret_cast<unsigned long long*>(aux);
    double* scale = reinterpret_cast<double*>(amax + d.gblocks * 32);
    int* ovf = tf_overflow(aux, d);
    GP_HIP(hipMemsetAsync(aux, 0, trace ? tf_aux_bytes(d) : (size_t)d.gblocks * 32 * 16 + 16, stream));
    if (m > 0) {
        hipLaunchKernelGGL(tf_rowmax_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)((m + 15) / 16)), dim3(256), 0, stream, fstar, N, m, amax);
        GP_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(tf_quant_kernel, dim3((unsigned)((d.gblocks * 32 + 255) / 256), (unsigned)(d.ksteps * 2)), dim3(256), 0, stream,
                       fstar, N, m, d, amax, reinterpret_cast<uint4*>(Gq), scale, ovf);
    GP_HIP(hipGetLastError());
    TfMfmaArgs a;
    a.Y8 = reinterpret_cast<const unsigned char*>(Y8); a.Gq = reinterpret_cast<const unsigned char*>(Gq); a.scale = scale; a.ovf = ovf;
    a.logpost = logpost; a.ldlp = ldlp; a.n = n; a.N = N; a.ksteps = d.ksteps; a.gblocks = d.gblocks; a.itiles = (int)(d.iblocks / 8);
    a.trace = trace ? tf_trace(aux, d) : nullptr;
    ProfPair pp;
    if (prof) GP_TRY(prof_pair_begin(prof, stream, pp));
    hipLaunchKernelGGL(tf_mfma_kernel, dim3((unsigned)(32 * a.itiles)), dim3(256), 2 * TF_STAGE, stream, a);
    GP_HIP(hipGetLastError());
    // (class 5; bytes: both operands and the result once)
    if (prof) GP_TRY(prof_pair_end(prof, stream, pp, 5, (double)TF_DIGITS * 2.0 * (double)N * (double)n * 2.0 * (double)m,
                                   (double)tf_y8_bytes(d) + (double)tf_gq_bytes(d) + 8.0 * (double)N * (double)n));
    return 0;
}

int* tf_overflow(void* aux, const TfDims& d) { return reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(aux) + (size_t)d.gblocks * 32 * 16); }

}  // namespace gpirt
