// gemm_f64.hip -- the fp64 MFMA GEMM core every stage is built on (gfx950, v_mfma_f64_16x16x4_f64).
//
//   C (M x N, column-major) = alpha * op(A) * op(B) + beta * C
//
// Work-group = 256 threads = 4 waves (2 x 2), block tile 128 x 128, K-step 16, each wave owns a
// 64 x 64 sub-tile = 4 x 4 MFMA tiles (16 accumulators x 4 fp64 = 128 VGPRs).  Operands are
// staged global -> registers -> LDS (double-buffered, one barrier per K-step); an operand whose
// M/N index is contiguous in memory is kept [k][mn] with a 16-double row pad, one whose K index is
// contiguous is kept [mn][k] with a 2-double pad -- both read conflict-free by ds_read_b64.
// The MFMA is issued with the roles swapped (A_mfma <- B fragment, B_mfma <- A fragment) so the
// accumulator's lane index runs along M, the contiguous dimension of C: every store instruction
// writes 128-byte row segments.
//
// Triangular modes let the callers skip structural zeros without reshaping anything:
//   TRI_SYRK_LOWER  only blocks on/below the block diagonal (potrf trailing update, C -= P P^T)
//   TRI_A_LOWER     op(A) is lower triangular: K loop stops at the block's last row  (trmm L Z)
//   TRI_A_UPPER     op(A) is upper triangular: K loop starts at the block's first row (L^T X)
//
// Reference call sites this core serves: cholS * res (src/mvnormal.h:10), the dtrsm/dgemv chain of
// src/draw-fstar.cpp:7,19,25, arma::chol's trailing update (src/gpirtMCMC.cpp:17,78,97) and the
// log-likelihood sums of src/draw-theta.cpp:15-19 restated as a GEMM.
#include "common.h"
#include "kernels.h"
#include "potf2.h"

#include <type_traits>

namespace gpirt {

namespace {

constexpr int BK = 16;
constexpr int T128_MIN = 448;        // number of 128 x 128 tiles from which a product uses the 128-tile kernel (2 per CU)
constexpr int LDS_K = BK + 2;        // [mn][k] row stride (doubles)
// Block tile T x T (T = 128: 4 x 4 MFMA tiles per wave, the throughput configuration;
// T = 64: 2 x 2 tiles per wave, 4x as many work-groups and ~4x shorter k-steps -- used when the
// 128-tile grid would leave most of the 256 CUs idle: panel updates, trsm levels, potrf tail).
template <int T> struct Cfg {
    static constexpr int LDS_MN = T + 16;                 // [k][mn] row stride (doubles)
    static constexpr int TILE = (BK * (T + 16) > T * LDS_K) ? BK * (T + 16) : T * LDS_K;
    static constexpr int NT = T / 32;                     // MFMA tiles per wave per dimension
    static constexpr int PASSES = T / 32;                 // double2 loads per thread per operand tile
    static constexpr int DEPTH = (T == 64) ? 4 : 1;       // K-steps in flight between global memory and LDS
    static constexpr int FDEPTH = (T == 64) ? 2 : 1;      // the same for the branch-free loop of interior tiles
};

struct GemmParams {
    const double* A; const double* B; double* C;
    int64_t lda, ldb, ldc;
    int M, N, K;
    int Mr;               // rows of op(A) that may be READ (>= M: padded operand), 0 = M
    double alpha, beta;
    int tri;
    int mblocks, nblocks;
    int fastA, fastB;     // operand base/ld are 16-byte friendly
    int64_t sA, sB, sC;   // batch strides (elements): blockIdx.y selects the problem (0 for a single GEMM)
    int ksplit;           // > 0: blockIdx.y selects the K range [y * ksplit, (y + 1) * ksplit) of ONE product (split-K)
    // fused epilogue (FUSE kernels only): work-group 0 factors this 64 x 64 block after its tile
    double* fz_A; int64_t fz_lda; int fz_nb; int fz_k0; int* fz_info;
    const int* run_if = nullptr;   // when set: the launch does nothing unless *run_if != 0 (a fallback product kept behind a device-side flag)
};

template <bool KCONTIG, int T>
__device__ __forceinline__ void load_tile(const double* __restrict__ G, int64_t ld, int mn0,
                                          int mnmax, int k0, int kmax, bool fast,
                                          double2 (&reg)[Cfg<T>::PASSES])
{
    const int t = threadIdx.x;
    constexpr int P = Cfg<T>::PASSES;
    constexpr int TPC = T / 2;              // threads per column of the [k][mn] image
    constexpr int CPP = 256 / TPC;          // columns per pass
    if (!KCONTIG) {
        const int r2 = (t % TPC) * 2;
#pragma unroll
        for (int pass = 0; pass < P; ++pass) {
            const int c = (t / TPC) + CPP * pass;
            const double* p = G + (int64_t)(mn0 + r2) + (int64_t)(k0 + c) * ld;
            if (fast) {
                reg[pass] = *reinterpret_cast<const double2*>(p);
            } else {
                const bool kin = (k0 + c) < kmax;
                reg[pass].x = (kin && (mn0 + r2) < mnmax) ? p[0] : 0.0;
                reg[pass].y = (kin && (mn0 + r2 + 1) < mnmax) ? p[1] : 0.0;
            }
        }
    } else {
        const int k2 = (t & 7) * 2;
#pragma unroll
        for (int pass = 0; pass < P; ++pass) {
            const int mn = (t >> 3) + 32 * pass;
            const double* p = G + (int64_t)(k0 + k2) + (int64_t)(mn0 + mn) * ld;
            if (fast) {
                reg[pass] = *reinterpret_cast<const double2*>(p);
            } else {
                const bool in = (mn0 + mn) < mnmax;
                reg[pass].x = (in && (k0 + k2) < kmax) ? p[0] : 0.0;
                reg[pass].y = (in && (k0 + k2 + 1) < kmax) ? p[1] : 0.0;
            }
        }
    }
}

template <bool KCONTIG, int T>
__device__ __forceinline__ void store_tile(double* __restrict__ s, const double2 (&reg)[Cfg<T>::PASSES])
{
    const int t = threadIdx.x;
    constexpr int P = Cfg<T>::PASSES;
    constexpr int TPC = T / 2, CPP = 256 / TPC, LDS_MN = Cfg<T>::LDS_MN;
    if (!KCONTIG) {
        const int r2 = (t % TPC) * 2;
#pragma unroll
        for (int pass = 0; pass < P; ++pass) {
            const int c = (t / TPC) + CPP * pass;
            *reinterpret_cast<double2*>(&s[c * LDS_MN + r2]) = reg[pass];
        }
    } else {
        const int k2 = (t & 7) * 2;
#pragma unroll
        for (int pass = 0; pass < P; ++pass) {
            const int mn = (t >> 3) + 32 * pass;
            *reinterpret_cast<double2*>(&s[mn * LDS_K + k2]) = reg[pass];
        }
    }
}

// TA: A is stored K x M (op(A) = A^T)  -> K-contiguous.   !TA: stored M x K -> M-contiguous.
// TB: B is stored N x K (op(B) = B^T)  -> N-contiguous.   !TB: stored K x N -> K-contiguous.
// main loop of one block tile over the K range [kbeg, kend): acc += op(A)[tile rows, k] op(B)[k, tile cols]
template <bool TA, bool TB, int T>
__device__ __forceinline__ void gemm_mainloop(const GemmParams& p, const int bi, const int bj, double* smem,
                                              const int kbeg, const int kend,
                                              d4 (&acc)[Cfg<T>::NT][Cfg<T>::NT])
{
    constexpr int BM = T, BN = T, LDS_MN = Cfg<T>::LDS_MN, TILE_DOUBLES = Cfg<T>::TILE, NT = Cfg<T>::NT;
    constexpr int WT = T / 2;                // wave tile edge
    double* sA = smem;                       // [2][TILE]
    double* sB = smem + 2 * TILE_DOUBLES;    // [2][TILE]
    const int i0 = bi * BM, j0 = bj * BN;

    constexpr bool A_KC = TA, B_KC = !TB;
    const bool fullA = p.fastA && (i0 + BM <= p.M);
    const bool fullB = p.fastB && (j0 + BN <= p.N);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    const int nk = (kend - kbeg + BK - 1) / BK;
    // Register ring of D K-steps between global memory and LDS: the tile of K-step kt + D is requested while
    // K-step kt computes.  D = 1 for the 128-tile (its 128 accumulator registers leave room for one stage and
    // two work-groups per CU hide the rest); D = 4 for the 64-tile, whose launches are often too small to put
    // more than one work-group on a CU -- its K-step (16 MFMAs per wave, 0.43 us) is then far shorter than an
    // L2 / MALL round trip and the loop ran at memory latency, 1.2 us per K-step.
    constexpr int D = Cfg<T>::DEPTH;
    double2 ra[D][Cfg<T>::PASSES], rb[D][Cfg<T>::PASSES];
    auto request = [&](int kt, double2 (&qa)[Cfg<T>::PASSES], double2 (&qb)[Cfg<T>::PASSES]) {
        const int k0 = kbeg + kt * BK;
        const bool kfull = (k0 + BK <= p.K);
        load_tile<A_KC, T>(p.A, p.lda, i0, p.M, k0, p.K, fullA && kfull, qa);
        load_tile<B_KC, T>(p.B, p.ldb, j0, p.N, k0, p.K, fullB && kfull, qb);
    };
    if (nk > 0) {
        request(0, ra[0], rb[0]);
        store_tile<A_KC, T>(sA, ra[0]);
        store_tile<B_KC, T>(sB, rb[0]);
#pragma unroll
        for (int d = 1; d < D; ++d)
            if (d < nk) request(d, ra[d % D], rb[d % D]);      // slots 1 .. D-1 hold K-steps 1 .. D-1
    }
    __syncthreads();

    for (int kt0 = 0; kt0 < nk; kt0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int kt = kt0 + d;                            // ring slot of K-step j is j % D; kt % D == d here
            if (kt < nk) {
                const int buf = kt & 1;
                const bool more = (kt + 1 < nk);
                if (kt + D < nk) request(kt + D, ra[d], rb[d]);     // slot d held K-step kt: already in LDS
                const double* cA = sA + buf * TILE_DOUBLES;
                const double* cB = sB + buf * TILE_DOUBLES;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    double a[NT], b[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int mi = wm * WT + t * 16 + l15;
                        const int ni = wn * WT + t * 16 + l15;
                        const int kq = kk * 4 + l4;
                        a[t] = A_KC ? cA[mi * LDS_K + kq] : cA[kq * LDS_MN + mi];
                        b[t] = B_KC ? cB[ni * LDS_K + kq] : cB[kq * LDS_MN + ni];
                    }
#pragma unroll
                    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                        for (int tm = 0; tm < NT; ++tm)
                            acc[tn][tm] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[tn], a[tm], acc[tn][tm], 0, 0, 0);
                }
                if (more) {   // (writing LDS ahead of this step's MFMAs was measured: 20 % slower)
                    store_tile<A_KC, T>(sA + (buf ^ 1) * TILE_DOUBLES, ra[(d + 1) % D]);
                    store_tile<B_KC, T>(sB + (buf ^ 1) * TILE_DOUBLES, rb[(d + 1) % D]);
                }
                __syncthreads();
            }
        }
    }
}

// ---- interior tiles: the branch-free main loop ------------------------------------------------------------
// Every operand tile of the K range is complete and 16-byte aligned: loads are `uniform base + per-lane
// 32-bit offset` (the base advances on the scalar unit), nothing is predicated, and the K-step is ordered so
// that the wave's non-MFMA work sits in the shadow of its own MFMAs:
//   fragments of sub-step 0  ->  tile kt+1: registers -> LDS (other buffer)  ->  tile kt+2: global -> registers
//   ->  64 MFMAs  ->  barrier.
// One register stage gives a full K-step of lead because it is emptied at the START of the step.
template <bool KCONTIG, int T>
__device__ __forceinline__ uint32_t fast_lane_offset(int64_t ld)
{
    const int t = threadIdx.x;
    constexpr int TPC = T / 2;
    return KCONTIG ? (uint32_t)((((t & 7) * 2) + (int64_t)(t >> 3) * ld) * 8)
                   : (uint32_t)((((t % TPC) * 2) + (int64_t)(t / TPC) * ld) * 8);
}
template <bool KCONTIG, int T>
__device__ __forceinline__ void load_tile_fast(const double* __restrict__ G, int64_t ld, uint32_t voff,
                                               double2 (&reg)[Cfg<T>::PASSES])
{
    constexpr int CPP = 256 / (T / 2);
#pragma unroll
    for (int pass = 0; pass < Cfg<T>::PASSES; ++pass) {
        const double* base = G + (int64_t)((KCONTIG ? 32 : CPP) * pass) * ld;     // uniform
        typedef double d2v __attribute__((ext_vector_type(2)));
        const d2v v = *reinterpret_cast<const d2v*>(reinterpret_cast<const char*>(base) + voff);
        reg[pass].x = v.x; reg[pass].y = v.y;
    }
}

template <bool KCONTIG, int T>
__device__ __forceinline__ void load_pass_fast(const double* __restrict__ G, int64_t ld, uint32_t voff, double2& reg,
                                               const int pass)
{
    constexpr int CPP = 256 / (T / 2);
    typedef double d2v __attribute__((ext_vector_type(2)));
    const double* base = G + (int64_t)((KCONTIG ? 32 : CPP) * pass) * ld;     // uniform
    const d2v v = *reinterpret_cast<const d2v*>(reinterpret_cast<const char*>(base) + voff);
    reg.x = v.x; reg.y = v.y;
}
template <bool KCONTIG, int T>
__device__ __forceinline__ void store_pass(double* __restrict__ s, const double2& reg, const int pass)
{
    const int t = threadIdx.x;
    constexpr int TPC = T / 2, CPP = 256 / TPC, LDS_MN = Cfg<T>::LDS_MN;
    if (!KCONTIG) *reinterpret_cast<double2*>(&s[((t / TPC) + CPP * pass) * LDS_MN + (t % TPC) * 2]) = reg;
    else          *reinterpret_cast<double2*>(&s[((t >> 3) + 32 * pass) * LDS_K + (t & 7) * 2]) = reg;
}

template <bool TA, bool TB, int T>
__device__ __forceinline__ void gemm_mainloop_fast(const GemmParams& p, const int bi, const int bj, double* smem,
                                                   const int kbeg, const int nk, d4 (&acc)[Cfg<T>::NT][Cfg<T>::NT])
{
    constexpr int LDS_MN = Cfg<T>::LDS_MN, TILE_DOUBLES = Cfg<T>::TILE, NT = Cfg<T>::NT;
    constexpr int WT = T / 2;
    constexpr bool A_KC = TA, B_KC = !TB;
    double* sA = smem;
    double* sB = smem + 2 * TILE_DOUBLES;
    const int i0 = bi * T, j0 = bj * T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    const double* gA = A_KC ? p.A + kbeg + (int64_t)i0 * p.lda : p.A + i0 + (int64_t)kbeg * p.lda;
    const double* gB = B_KC ? p.B + kbeg + (int64_t)j0 * p.ldb : p.B + j0 + (int64_t)kbeg * p.ldb;
    const int64_t stepA = A_KC ? (int64_t)BK : (int64_t)BK * p.lda;
    const int64_t stepB = B_KC ? (int64_t)BK : (int64_t)BK * p.ldb;
    const uint32_t voffA = fast_lane_offset<A_KC, T>(p.lda);
    const uint32_t voffB = fast_lane_offset<B_KC, T>(p.ldb);
    // fragment read offsets (doubles) of this lane inside a tile image, sub-step 0, MFMA tile 0
    const int fa = A_KC ? (wm * WT + l15) * LDS_K + l4 : l4 * LDS_MN + wm * WT + l15;
    const int fb = B_KC ? (wn * WT + l15) * LDS_K + l4 : l4 * LDS_MN + wn * WT + l15;
    constexpr int FA_T = A_KC ? 16 * LDS_K : 16, FA_K = A_KC ? 4 : 4 * LDS_MN;
    constexpr int FB_T = B_KC ? 16 * LDS_K : 16, FB_K = B_KC ? 4 : 4 * LDS_MN;

    // Register ring of D stages between global memory and LDS: tile j waits in slot j % D.  Step kt moves tile
    // kt+1 from its slot to the other LDS buffer and requests tile kt+1+D into the slot it just emptied, so a
    // tile has D K-steps to arrive (D = 1 for the 128-tile: 64 MFMAs = 1.7 us per step; D = 4 for the 64-tile,
    // whose step is 16 MFMAs).
    constexpr int D = Cfg<T>::FDEPTH, P = Cfg<T>::PASSES, NM = NT * NT;
    double2 ra[D][P], rb[D][P];
    load_tile_fast<A_KC, T>(gA, p.lda, voffA, ra[0]);
    load_tile_fast<B_KC, T>(gB, p.ldb, voffB, rb[0]);
    gA += stepA; gB += stepB;
    store_tile<A_KC, T>(sA, ra[0]);
    store_tile<B_KC, T>(sB, rb[0]);
#pragma unroll
    for (int j = 1; j <= D; ++j)
        if (j < nk) {
            load_tile_fast<A_KC, T>(gA, p.lda, voffA, ra[j % D]);
            load_tile_fast<B_KC, T>(gB, p.ldb, voffB, rb[j % D]);
            gA += stepA; gB += stepB;
        }
    __syncthreads();

    // One K-step, written in issue order with scheduling fences so the compiler keeps it.  Four groups of NM
    // MFMAs: the PREVIOUS step's sub-step 3 (its fragments were read before the barrier), then sub-steps 0, 1, 2;
    // ahead of each group the fragments of the following sub-step are requested; behind the first 4P MFMAs sits
    // one memory instruction each -- 2P LDS writes (tile kt+1), then 2P global loads (tile kt+1+D); barrier.
    // So the LDS latency after the barrier, the LDS writes and the global loads all sit behind MFMAs of the same
    // wave.  FIRST: no previous step.  STEADY: tiles kt+1 and kt+1+D exist (no conditions: one basic block).
    double a3[NT], b3[NT];
    auto mfma = [&](const double (&fa_)[NT], const double (&fb_)[NT], const int i) {
        acc[i / NT][i % NT] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb_[i / NT], fa_[i % NT], acc[i / NT][i % NT], 0, 0, 0);
    };
    auto frags = [&](const double* cA, const double* cB, const int kk, double (&fa_)[NT], double (&fb_)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) { fa_[t] = cA[kk * FA_K + t * FA_T]; fb_[t] = cB[kk * FB_K + t * FB_T]; }
    };
    static_assert(4 * P <= 4 * NM, "one memory instruction per MFMA of the step at most");
    auto kstep = [&](const int buf, auto slot_c, auto first, auto steady, const bool st, const bool ld) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr bool FIRST = decltype(first)::value, STEADY = decltype(steady)::value;
        const double* cA = sA + buf * TILE_DOUBLES + fa;
        const double* cB = sB + buf * TILE_DOUBLES + fb;
        double* nA = sA + (buf ^ 1) * TILE_DOUBLES;
        double* nB = sB + (buf ^ 1) * TILE_DOUBLES;
        double f0a[NT], f0b[NT], f1a[NT], f1b[NT], f2a[NT], f2b[NT];
        // memory instruction number c of the step (behind MFMA number c)
        auto memop = [&](const int c) {
            if (c < P)          { if (STEADY || st) store_pass<A_KC, T>(nA, ra[SLOT][c], c); }
            else if (c < 2 * P) { if (STEADY || st) store_pass<B_KC, T>(nB, rb[SLOT][c - P], c - P); }
            else if (c < 3 * P) { if (STEADY || ld) load_pass_fast<A_KC, T>(gA, p.lda, voffA, ra[SLOT][c - 2 * P], c - 2 * P); }
            else if (c < 4 * P) { if (STEADY || ld) load_pass_fast<B_KC, T>(gB, p.ldb, voffB, rb[SLOT][c - 3 * P], c - 3 * P); }
        };
        frags(cA, cB, 0, f0a, f0b);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (!FIRST) mfma(a3, b3, i);
            memop(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        frags(cA, cB, 1, f1a, f1b);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma(f0a, f0b, i);
            if (NM + i < 4 * P) { memop(NM + i); __builtin_amdgcn_sched_barrier(0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        frags(cA, cB, 2, f2a, f2b);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mfma(f1a, f1b, i);
            if (2 * NM + i < 4 * P) { memop(2 * NM + i); __builtin_amdgcn_sched_barrier(0); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (STEADY || ld) { gA += stepA; gB += stepB; }
        frags(cA, cB, 3, a3, b3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) mfma(f2a, f2b, i);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    // step kt uses ring slot (kt + 1) % D; a step outside the steady state picks its instantiation by slot
    auto edge_step = [&](const int kt) {
        const bool st = kt + 1 < nk, ld = kt + 1 + D < nk;
        const int slot = (kt + 1) % D;
        if (D == 1 || slot == 0)      kstep(kt & 1, std::integral_constant<int, 0>{}, no{}, no{}, st, ld);
        else if (slot == 1)           kstep(kt & 1, std::integral_constant<int, 1 % D>{}, no{}, no{}, st, ld);
        else if (slot == 2)           kstep(kt & 1, std::integral_constant<int, 2 % D>{}, no{}, no{}, st, ld);
        else                          kstep(kt & 1, std::integral_constant<int, 3 % D>{}, no{}, no{}, st, ld);
    };
    static_assert(D >= 1 && D <= 4, "edge_step enumerates the slots of a ring of 1 or 4");
    kstep(0, std::integral_constant<int, 1 % D>{}, yes{}, no{}, 1 < nk, 1 + D < nk);
    int kt = 1;
    // steady state, unrolled over the ring: steps kt .. kt+D-1 all have tiles kt+1 .. kt+2D in range
    for (; kt + 2 * D < nk; kt += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            if (u == 0)      kstep((kt + 0) & 1, std::integral_constant<int, (2 + 0) % D>{}, no{}, yes{}, true, true);
            else if (u == 1) kstep((kt + 1) & 1, std::integral_constant<int, (2 + 1) % D>{}, no{}, yes{}, true, true);
            else if (u == 2) kstep((kt + 2) & 1, std::integral_constant<int, (2 + 2) % D>{}, no{}, yes{}, true, true);
            else             kstep((kt + 3) & 1, std::integral_constant<int, (2 + 3) % D>{}, no{}, yes{}, true, true);
        }
    }
    for (; kt < nk; ++kt) edge_step(kt);
#pragma unroll
    for (int i = 0; i < NM; ++i) mfma(a3, b3, i);
}

// C tile <- alpha * acc + beta * C (masked at the matrix edge and, for syrk, above the diagonal)
template <int T>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const int bi, const int bj,
                                              const d4 (&acc)[Cfg<T>::NT][Cfg<T>::NT])
{
    constexpr int BM = T, BN = T, NT = Cfg<T>::NT;
    constexpr int WT = T / 2;
    const int i0 = bi * BM, j0 = bj * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    // epilogue: acc[tn][tm][r] = C[m = i0+wm*64+tm*16+l15][n = j0+wn*64+tn*16+l4+4r]
    // C is read in batches of 16 independent loads before anything is stored: a store followed by
    // a load of the same array would otherwise serialise 64 L2 round trips per thread.
    const double alpha = p.alpha, beta = p.beta;
    const bool interior = (i0 + BM <= p.M) && (j0 + BN <= p.N);
    const bool lower_mask = (p.tri == TRI_SYRK_LOWER) && (bi == bj);
    const bool use_c = (beta != 0.0);
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
        double cv[4][NT];
        if (use_c) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = j0 + wn * WT + tn * 16 + l4 + 4 * r;
#pragma unroll
                for (int tm = 0; tm < NT; ++tm) {
                    const int m = i0 + wm * WT + tm * 16 + l15;
                    cv[r][tm] = (interior || (m < p.M && n < p.N)) ? p.C[(int64_t)m + (int64_t)n * p.ldc] : 0.0;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = j0 + wn * WT + tn * 16 + l4 + 4 * r;
#pragma unroll
            for (int tm = 0; tm < NT; ++tm) {
                const int m = i0 + wm * WT + tm * 16 + l15;
                if ((interior || (m < p.M && n < p.N)) && !(lower_mask && m < n)) {
                    double v = alpha * acc[tn][tm][r];
                    if (use_c) v += beta * cv[r][tm];
                    p.C[(int64_t)m + (int64_t)n * p.ldc] = v;
                }
            }
        }
    }
}

template <bool TA, bool TB, int T>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, const int bi, const int bj, double* smem)
{
    constexpr int NT = Cfg<T>::NT;
    int kbeg = 0, kend = p.K;
    if (p.tri == TRI_A_LOWER) kend = min(p.K, bi * T + T);
    if (p.tri == TRI_A_UPPER) kbeg = ((bi * T) / BK) * BK;
    if (p.ksplit > 0) { kbeg = (int)blockIdx.y * p.ksplit; kend = min(p.K, kbeg + p.ksplit); }
    if (p.ksplit < 0) {
        // triangular A with few work-groups: blockIdx.y takes one of -ksplit equal parts of THIS tile's K range
        // (an empty part writes zeros into its partial result)
        const int S = -p.ksplit;
        const int part = (((kend - kbeg + S - 1) / S) + BK - 1) / BK * BK;
        kbeg += (int)blockIdx.y * part;
        kend = min(kend, kbeg + part);
        if (kend < kbeg) kend = kbeg;
    }
    d4 acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    const bool fast = p.fastA && p.fastB && (bi * T + T <= (p.Mr > p.M ? p.Mr : p.M)) && (bj * T + T <= p.N) && kend <= p.K &&
                      kend > kbeg && ((kend - kbeg) % BK) == 0;
    if (fast) gemm_mainloop_fast<TA, TB, T>(p, bi, bj, smem, kbeg, (kend - kbeg) / BK, acc);
    else gemm_mainloop<TA, TB, T>(p, bi, bj, smem, kbeg, kend, acc);
    gemm_epilogue<T>(p, bi, bj, acc);
}

// PAD only distinguishes instantiations by name (see launch_gemm_trailing); it adds PAD doubles of LDS.
// FUSE: the work-group that owns work item 0 (the tile holding the next diagonal block of the Cholesky
// panel) factors that block right after producing it -- the potf2 launch and its kernel boundary
// disappear from the panel chain (potrf.hip).
template <bool TA, bool TB, int T, int PAD = 0, bool FUSE = false>
__global__ __launch_bounds__(256, (T == 64 ? 3 : 2)) void gemm_f64_kernel(GemmParams p)
{
    __shared__ __attribute__((aligned(16))) double smem[4 * Cfg<T>::TILE + PAD];
    if (p.run_if != nullptr && *p.run_if == 0) return;
    p.A += (int64_t)blockIdx.y * p.sA;
    p.B += (int64_t)blockIdx.y * p.sB;
    p.C += (int64_t)blockIdx.y * p.sC;
    int bi, bj;
    if (p.tri == TRI_SYRK_LOWER) {
        // Lower trapezoid (M >= N): block columns bj = 0..nblocks-1, each holding block rows bj..mblocks-1.
        // The order the tiles are handed out in is a pure SPEED choice (bijective for any grid):
        //  * work-groups are dealt round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md), so ids with equal
        //    id % 8 share an L2: each such group gets a CONTIGUOUS range of the logical tile order;
        //  * the logical order runs through super-columns of SW block columns, row by row inside a super-column:
        //    the ~64-96 tiles resident on an XCD at a time are then ~8 block rows x 8 block columns and stream
        //    8 + 8 operand blocks through that L2 instead of 64 + 1 (a plain column-major order) or, dealt
        //    round-robin, the same block column on all eight L2s at once.
        constexpr int SW = 8;
        const int mb = p.mblocks, nb = p.nblocks;
        const int G = (int)gridDim.x;
        const int id = (int)blockIdx.x;
        const int xq = G >> 3, xr = G & 7, xcd = id & 7;
        int t = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
        int c0 = 0, w = nb < SW ? nb : SW;
        for (;;) {                                     // <= nb / SW iterations, scalar
            const int cnt = w * (w + 1) / 2 + (mb - c0 - w) * w;
            if (t < cnt || c0 + w >= nb) break;
            t -= cnt;
            c0 += w;
            w = (nb - c0) < SW ? (nb - c0) : SW;
        }
        const int tri_cnt = w * (w + 1) / 2;
        if (t < tri_cnt) {
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= t) ++i;    // row i of the triangle holds i + 1 tiles
            bi = c0 + i;
            bj = c0 + (t - i * (i + 1) / 2);
        } else {
            const int v = t - tri_cnt;
            bi = c0 + w + v / w;
            bj = c0 + v % w;
        }
    } else if (p.tri == TRI_A_LOWER || p.tri == TRI_A_UPPER) {
        // triangular A: the K range of block row bi grows (or shrinks) linearly with bi, so each
        // work-group takes the pair (mblocks-1-q, q): every work-group then carries the same
        // number of k-steps instead of the last block rows setting the kernel's duration.
        const int half = (p.mblocks + 1) / 2;
        const int q = blockIdx.x % half;
        bj = blockIdx.x / half;
        bi = p.mblocks - 1 - q;
        gemm_tile<TA, TB, T>(p, bi, bj, smem);
        if (q == bi) return;
        bi = q;
    } else {
        bi = blockIdx.x % p.mblocks;
        bj = blockIdx.x / p.mblocks;
    }
    gemm_tile<TA, TB, T>(p, bi, bj, smem);
    if (FUSE) {
        __shared__ int sfail;
        if (blockIdx.x == 0 && p.fz_A != nullptr) {
            // the tile was written by this work-group's own lanes: make the stores visible to all of them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            potf2_64_body(p.fz_A, p.fz_lda, p.fz_nb, p.fz_k0, p.fz_info, smem, &sfail);
        }
    }
}

}  // namespace

template <int T>
static int launch_gemm_t(hipStream_t stream, bool ta, bool tb, GemmParams p, int batch = 1);

// The Cholesky trailing update gets its own instantiations of the NT / syrk-lower 128-tile kernel
// (PAD = 8: same code, +64 B of LDS) so kernel traces and PMC profiles can tell it apart by name
// from the other NT products (draw_theta's GEMM).
static int launch_gemm_trailing(hipStream_t stream, GemmParams p)
{
    constexpr int T = 128;
    p.mblocks = (p.M + T - 1) / T;
    p.nblocks = (p.N + T - 1) / T;
    const int64_t grid = (int64_t)p.nblocks * p.mblocks - (int64_t)p.nblocks * (p.nblocks - 1) / 2;
    hipLaunchKernelGGL((gemm_f64_kernel<false, true, 128, 8>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    GP_HIP(hipGetLastError());
    return 0;
}

// ... and so do the factorisation's 64-tile updates (trailing updates too small for 128-tiles, the K = 512 update
// between the sub-panels of an outer panel).
static int launch_gemm_syrk64(hipStream_t stream, GemmParams p)
{
    constexpr int T = 64;
    p.mblocks = (p.M + T - 1) / T;
    p.nblocks = (p.N + T - 1) / T;
    const int64_t grid = (int64_t)p.nblocks * p.mblocks - (int64_t)p.nblocks * (p.nblocks - 1) / 2;
    hipLaunchKernelGGL((gemm_f64_kernel<false, true, 64, 8>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    GP_HIP(hipGetLastError());
    return 0;
}

// ... and the K = 512 update BETWEEN the two sub-panels of an outer panel (potrf.hip, panel_update: on the pivot chain, beside
// the deferred updates) runs under a name of its own (PAD = 16), so that rocprofv3's per-kernel averages show the classes
// bench.py's roofline reports -- trailing / deferred updates (PAD = 8) and in-panel updates -- without bench.py's own timers.
static int launch_gemm_inpanel(hipStream_t stream, GemmParams p, bool t128)
{
    const int T = t128 ? 128 : 64;
    p.mblocks = (p.M + T - 1) / T;
    p.nblocks = (p.N + T - 1) / T;
    const int64_t grid = (int64_t)p.nblocks * p.mblocks - (int64_t)p.nblocks * (p.nblocks - 1) / 2;
    if (t128) hipLaunchKernelGGL((gemm_f64_kernel<false, true, 128, 16>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    else      hipLaunchKernelGGL((gemm_f64_kernel<false, true, 64, 16>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    GP_HIP(hipGetLastError());
    return 0;
}

// Panel update of the blocked Cholesky (K <= 64, lower trapezoid) with the factorisation of the next
// diagonal block fused into work-group 0: C = A[r0:, r0:c1] and the block is C's leading 64 x 64 tile.
int launch_gemm_update_potf2(hipStream_t stream, int64_t M, int64_t N, int64_t K, const double* P, int64_t ldp,
                             double* C, int64_t ldc, int nb_next, int k0_next, int* info)
{
    if (M <= 0 || N <= 0) return 0;
    constexpr int T = 64;
    GemmParams p;
    p.A = P; p.B = P; p.C = C;
    p.lda = ldp; p.ldb = ldp; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.Mr = 0;
    p.alpha = -1.0; p.beta = 1.0; p.tri = TRI_SYRK_LOWER;
    p.fastA = p.fastB = (((uintptr_t)P & 15) == 0) && (ldp % 2 == 0);
    p.mblocks = (p.M + T - 1) / T;
    p.nblocks = (p.N + T - 1) / T;
    p.fz_A = C; p.fz_lda = ldc; p.fz_nb = nb_next; p.fz_k0 = k0_next; p.fz_info = info;
    p.sA = p.sB = p.sC = 0; p.ksplit = 0;
    const int64_t grid = (int64_t)p.nblocks * p.mblocks - (int64_t)p.nblocks * (p.nblocks - 1) / 2;
    hipLaunchKernelGGL((gemm_f64_kernel<false, true, 64, 0, true>), dim3((unsigned)grid), dim3(256), 0, stream, p);
    GP_HIP(hipGetLastError());
    return 0;
}

// Several trailing updates of ONE block column by consecutive panels, as one launch: the panels' columns are contiguous in
// memory, so  C -= sum_q P_q P_q^T  is a single product over K = parts * kpart cut at the panel boundaries (blockIdx.y picks
// the panel).  Each part lands in its own slab of `work` as alpha * P_q P_q^T, and the sum kernel then applies them to C ONE
// AFTER THE OTHER in panel order -- c = part_q + c, exactly what the epilogue of a separate launch per panel computes
// (v = alpha * acc; v += beta * c with beta = 1) -- so the result is bit-identical to the sequential launches, while
// the products themselves run side by side (potrf.hip, deferred updates: 2-7 dependent launches of 150-900 tiles each,
// every one a partial round of the chip, become one grid).
__global__ __launch_bounds__(256) void apply_parts_seq_lower_kernel(const double* __restrict__ part, int64_t M, int64_t N, int64_t stride,
                                                                    int nparts, double* __restrict__ out, int64_t ldo)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int64_t r = i % M, c = i / M;
    if (r < c) return;
    double* o = out + r + c * ldo;
    double v = *o;
    for (int q = 0; q < nparts; ++q) v = part[i + q * stride] + v;
    *o = v;
}

int launch_syrk_panels(hipStream_t stream, int64_t M, int64_t N, int64_t kpart, int nparts, double alpha, const double* A,
                       int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, double* work)
{
    if (M <= 0 || N <= 0 || nparts <= 0) return 0;
    if (M < N || (kpart % BK) != 0) { set_error("launch_syrk_panels: needs M >= N and a part depth that is a multiple of 16"); return GPIRT_E_ARG; }
    constexpr int T = 64;
    GemmParams p;
    p.A = A; p.B = B; p.C = work;
    p.lda = lda; p.ldb = ldb; p.ldc = M;
    p.M = (int)M; p.N = (int)N; p.K = (int)(kpart * nparts); p.Mr = 0;
    p.alpha = alpha; p.beta = 0.0; p.tri = TRI_SYRK_LOWER;
    p.fz_A = nullptr; p.fz_lda = 0; p.fz_nb = 0; p.fz_k0 = 0; p.fz_info = nullptr;
    p.sA = p.sB = 0; p.sC = M * N;
    p.ksplit = (int)kpart;
    p.fastA = (((uintptr_t)A & 15) == 0) && (lda % 2 == 0);
    p.fastB = (((uintptr_t)B & 15) == 0) && (ldb % 2 == 0);
    p.mblocks = (p.M + T - 1) / T;
    p.nblocks = (p.N + T - 1) / T;
    const int64_t grid = (int64_t)p.nblocks * p.mblocks - (int64_t)p.nblocks * (p.nblocks - 1) / 2;
    hipLaunchKernelGGL((gemm_f64_kernel<false, true, 64, 8>), dim3((unsigned)grid, (unsigned)nparts), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(apply_parts_seq_lower_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, stream, work, M, N, M * N,
                       nparts, C, ldc);
    GP_HIP(hipGetLastError());
    return 0;
}

bool gemm_trailing_uses_128(int64_t M, int64_t N, bool background)
{
    (void)background;      // (one threshold for both since round 3: 150-320 measured 4.88-5.24 ms against 4.85)
    const int64_t mb = (M + 127) / 128, nb = (N + 127) / 128;
    return nb * mb - nb * (nb - 1) / 2 >= T128_MIN;
}

template <int T>
static int launch_gemm_t(hipStream_t stream, bool ta, bool tb, GemmParams p, int batch)
{
    p.mblocks = (p.M + T - 1) / T;
    p.nblocks = (p.N + T - 1) / T;
    int64_t grid;
    if (p.tri == TRI_SYRK_LOWER) grid = (int64_t)p.nblocks * p.mblocks - (int64_t)p.nblocks * (p.nblocks - 1) / 2;
    else if (p.tri == TRI_A_LOWER || p.tri == TRI_A_UPPER) grid = (int64_t)((p.mblocks + 1) / 2) * p.nblocks;
    else grid = (int64_t)p.mblocks * p.nblocks;
    dim3 g((unsigned)grid, (unsigned)batch), b(256);
    if (!ta && tb)       hipLaunchKernelGGL((gemm_f64_kernel<false, true, T>),  g, b, 0, stream, p);
    else if (!ta && !tb) hipLaunchKernelGGL((gemm_f64_kernel<false, false, T>), g, b, 0, stream, p);
    else if (ta && !tb)  hipLaunchKernelGGL((gemm_f64_kernel<true, false, T>),  g, b, 0, stream, p);
    else                 hipLaunchKernelGGL((gemm_f64_kernel<true, true, T>),   g, b, 0, stream, p);
    GP_HIP(hipGetLastError());
    return 0;
}

// `batch` independent products of one shape (TRI_NONE / TRI_A_* only), operands `stride*` elements apart
int launch_gemm_batched(hipStream_t stream, bool ta, bool tb, int tri, int64_t M, int64_t N, int64_t K, double alpha,
                        const double* A, int64_t lda, int64_t strideA, const double* B, int64_t ldb, int64_t strideB,
                        double beta, double* C, int64_t ldc, int64_t strideC, int batch)
{
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (tri == TRI_SYRK_LOWER || tri == TRI_SYRK_LOWER_TRAILING || tri == TRI_SYRK_LOWER_BACKGROUND) { set_error("batched gemm: no syrk mode"); return GPIRT_E_ARG; }
    GemmParams p;
    p.A = A; p.B = B; p.C = C;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.Mr = 0;
    p.alpha = alpha; p.beta = beta; p.tri = tri;
    p.mblocks = p.nblocks = 0;
    p.fz_A = nullptr; p.fz_lda = 0; p.fz_nb = 0; p.fz_k0 = 0; p.fz_info = nullptr;
    p.sA = strideA; p.sB = strideB; p.sC = strideC; p.ksplit = 0;
    p.fastA = (((uintptr_t)A & 15) == 0) && (lda % 2 == 0) && (strideA % 2 == 0);
    p.fastB = (((uintptr_t)B & 15) == 0) && (ldb % 2 == 0) && (strideB % 2 == 0);
    return launch_gemm_t<64>(stream, ta, tb, p, batch);
}

// Split-K: C_q = alpha * op(A)[:, Kq] op(B)[Kq, :] for the `nsplit` consecutive K ranges Kq, written to
// Cpart + q * strideC.  For products whose M x N is far too small to fill the chip but whose K is long
// (draw_fstar's B^T [B | W], 64 x 1088 x 8192); the caller adds the parts in a fixed order.
// out (ldo) = beta * out + sum of the parts (each M x N, leading dimension M), added in index order
__global__ __launch_bounds__(256) void sum_parts_kernel(const double* __restrict__ part, int64_t M, int64_t N,
                                                        int64_t stride, int nparts, double beta, double* __restrict__ out,
                                                        int64_t ldo)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    double acc = 0.0;
    for (int q = 0; q < nparts; ++q) acc += part[i + q * stride];
    const int64_t r = i % M, c = i / M;
    double* o = out + r + c * ldo;
    *o = (beta != 0.0) ? beta * (*o) + acc : acc;
}

int launch_gemm_splitk(hipStream_t stream, bool ta, bool tb, int64_t M, int64_t N, int64_t K, double alpha,
                       const double* A, int64_t lda, const double* B, int64_t ldb, double* Cpart, int64_t ldc,
                       int64_t strideC, int nsplit, double* Cout, int64_t ldout, double beta_out)
{
    if (M <= 0 || N <= 0 || nsplit <= 0) return 0;
    GemmParams p;
    p.A = A; p.B = B; p.C = Cpart;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.Mr = 0;
    p.alpha = alpha; p.beta = 0.0; p.tri = TRI_NONE;
    p.mblocks = p.nblocks = 0;
    p.fz_A = nullptr; p.fz_lda = 0; p.fz_nb = 0; p.fz_k0 = 0; p.fz_info = nullptr;
    p.sA = p.sB = 0; p.sC = strideC;
    p.ksplit = (int)((((K + nsplit - 1) / nsplit) + BK - 1) / BK * BK);
    const int parts = (int)((K + p.ksplit - 1) / p.ksplit);
    p.fastA = (((uintptr_t)A & 15) == 0) && (lda % 2 == 0);
    p.fastB = (((uintptr_t)B & 15) == 0) && (ldb % 2 == 0);
    GP_TRY(launch_gemm_t<64>(stream, ta, tb, p, parts));
    if (ldc != M) { set_error("split-K parts must be dense (ldc == M)"); return GPIRT_E_ARG; }
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, stream, Cpart, M, N,
                       strideC, parts, beta_out, Cout, ldout);
    GP_HIP(hipGetLastError());
    return 0;
}

// number of K ranges launch_gemm will cut this product into (1 = not split).  A split product reads A and B
// completely before its result is written by the summing kernel, so C may then alias an operand.
int gemm_split_count(gpirt_handle_t h, hipStream_t stream, int tri, int64_t M, int64_t N, int64_t K)
{
    constexpr bool splitk_on = true;
    constexpr int t128_min = T128_MIN;
    const int64_t tiles64 = ((M + 63) / 64) * ((N + 63) / 64);
    if (!splitk_on || h == nullptr || stream != h->stream || tri != TRI_NONE || tiles64 > 320 || K < 256) return 1;
    if (((M + 127) / 128) * ((N + 127) / 128) >= t128_min) return 1;
    int64_t split = 640 / tiles64;
    if (split > K / 64) split = K / 64;
    if (split > 16) split = 16;
    return split >= 2 ? (int)split : 1;
}

// ... for PLAIN products from 256 tiles on (one work-group per CU): the trsm recursion's large updates (4096 x 1024 x 4096 at
// the metric size) run 712 -> ~600 us that way, draw_fstar as written 3.50 -> 3.36 ms (round 6; 128 tiles: 3.88)
#ifndef T128_PLAIN_MIN
#define T128_PLAIN_MIN 256
#endif
static int t128_min_or_default() { return T128_PLAIN_MIN; }

int launch_gemm(gpirt_handle_t h, hipStream_t stream, bool ta, bool tb, int tri, int64_t M,
                int64_t N, int64_t K, double alpha, const double* A, int64_t lda, const double* B,
                int64_t ldb, double beta, double* C, int64_t ldc, int64_t Mread, const int* run_if)
{
    if (M <= 0 || N <= 0) return 0;
    if (run_if && tri != TRI_NONE) { set_error("a conditional product has to be a plain one"); return GPIRT_E_ARG; }
    GemmParams p;
    p.run_if = run_if;
    p.A = A; p.B = B; p.C = C;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.Mr = (!ta && Mread > M && Mread <= lda) ? (int)Mread : 0;
    p.alpha = alpha; p.beta = beta; p.tri = tri;
    p.mblocks = p.nblocks = 0;
    p.fz_A = nullptr; p.fz_lda = 0; p.fz_nb = 0; p.fz_k0 = 0; p.fz_info = nullptr;
    p.sA = p.sB = p.sC = 0; p.ksplit = 0;
    p.fastA = (((uintptr_t)A & 15) == 0) && (lda % 2 == 0);
    p.fastB = (((uintptr_t)B & 15) == 0) && (ldb % 2 == 0);
    if (tri == TRI_SYRK_LOWER_TRAILING || tri == TRI_SYRK_LOWER_BACKGROUND) {
        const bool bg = (tri == TRI_SYRK_LOWER_BACKGROUND);
        p.tri = TRI_SYRK_LOWER;
        if (M < N || ta || !tb) { set_error("trailing syrk mode needs the NT form and M >= N"); return GPIRT_E_ARG; }
        if (gemm_trailing_uses_128(M, N, bg)) return launch_gemm_trailing(stream, p);
        return launch_gemm_syrk64(stream, p);
    }
    if (tri == TRI_SYRK_LOWER && M < N) { set_error("syrk mode needs M >= N"); return GPIRT_E_ARG; }
    if (tri == TRI_SYRK_LOWER && !ta && tb) {
        // the update between the sub-panels of an outer panel (only potrf.hip's panel_update asks for this mode): same split
        // between the tile sizes as every other product, under the PAD = 16 names (launch_gemm_inpanel)
        const int64_t mb_ = (M + 127) / 128, nb_ = (N + 127) / 128;
        constexpr int t128_min_ = T128_MIN;
        return launch_gemm_inpanel(stream, p, nb_ * mb_ - nb_ * (nb_ - 1) / 2 >= t128_min_);
    }
    // 128-tiles when they already give every CU >= 2 work-groups, 64-tiles otherwise
    const int64_t mb = (M + 127) / 128, nb = (N + 127) / 128;
    int64_t blocks128 = (tri == TRI_SYRK_LOWER) ? nb * mb - nb * (nb - 1) / 2 : mb * nb;
    // triangular A: the work-groups take block rows in pairs (long + short K range), so there are mb * nb / 2 of them,
    // each with the full K range -- 128-tiles only when those pairs fill the chip (8192 x 1024: 256 of them; 8192 x 512,
    // the per-rank shape on two GPUs, has 128 and ran as long as 8192 x 1024: 1043 us instead of 540 with 64-tiles)
    if (tri == TRI_A_LOWER || tri == TRI_A_UPPER) blocks128 = (blocks128 / 2 >= 224) ? t128_min_or_default() : 0;
    const int t128_min = t128_min_or_default();
    if (blocks128 >= t128_min) return launch_gemm_t<128>(stream, ta, tb, p);
    // Triangular A and few column tiles (nu = L z for the item columns of one rank of several): the paired work-groups
    // -- each with the full K range -- do not fill the chip (8192 x 128: 128 of them, 290 us).  The K range of every
    // tile is cut into S parts computed side by side and added in a fixed order, like the split below.
    if ((tri == TRI_A_LOWER || tri == TRI_A_UPPER) && h != nullptr && stream == h->stream && K >= 1024) {
        constexpr bool tri_split_on = true;
        const int64_t pairs = (((M + 63) / 64 + 1) / 2) * ((N + 63) / 64);
        // (slots = 512: two work-groups on every CU; 384 left half the CUs with one -- 8192 x 128: 183 -> 156 us, x 256: 304 -> 289, round 6)
        int S = (int)(512 / (pairs > 0 ? pairs : 1));
        if (S > 4) S = 4;
        if (tri_split_on && S >= 2) {
            const size_t need = (size_t)S * (size_t)M * (size_t)N * sizeof(double);
            if (h->splitk_bytes < need) {
                GP_HIP(hipStreamSynchronize(stream));
                if (h->side) GP_HIP(hipStreamSynchronize(h->side));
                if (h->d_splitk) GP_HIP(hipFree(h->d_splitk));
                h->d_splitk = nullptr; h->splitk_bytes = 0;
                GP_HIP(hipMalloc(&h->d_splitk, need));
                h->splitk_bytes = need;
            }
            GemmParams q = p;
            q.C = h->d_splitk; q.ldc = M; q.sC = M * N; q.beta = 0.0; q.ksplit = -S;
            GP_TRY(launch_gemm_t<64>(stream, ta, tb, q, S));
            hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, stream, h->d_splitk, M, N,
                               M * N, S, beta, C, ldc);
            GP_HIP(hipGetLastError());
            return 0;
        }
    }
    // Few tiles and a long K: a lone work-group per CU runs its K loop at LDS / barrier latency (~1.1 us per
    // K-step against 0.43 us of MFMA), so the K range is cut into `split` parts computed side by side and added
    // in a fixed order .  Main stream only: the parts share one workspace.
    {
        // (a conditional product stays one launch of the kernel that reads its flag)
        const int split = run_if ? 0 : gemm_split_count(h, stream, tri, M, N, K);
        if (split >= 2) {
            const size_t need = (size_t)split * (size_t)M * (size_t)N * sizeof(double);
            if (h->splitk_bytes < need) {
                GP_HIP(hipStreamSynchronize(stream));
                if (h->side) GP_HIP(hipStreamSynchronize(h->side));
                if (h->d_splitk) GP_HIP(hipFree(h->d_splitk));
                h->d_splitk = nullptr; h->splitk_bytes = 0;
                GP_HIP(hipMalloc(&h->d_splitk, need));
                h->splitk_bytes = need;
            }
            return launch_gemm_splitk(stream, ta, tb, M, N, K, alpha, A, lda, B, ldb, h->d_splitk, M, M * N, split,
                                      C, ldc, beta);
        }
    }
    return launch_gemm_t<64>(stream, ta, tb, p);
}

}  // namespace gpirt
