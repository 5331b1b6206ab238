// rng_ess.hip -- random-number fills, ll_bar() and the elliptical slice sampler.
//
//   item_fill        GPIRT_RNG_ITEM uniforms / normals (Philox4x32-10 sub-streams)
//   rstream_normals  R::rnorm(0,1) replay (src/mvnormal.h:7-9) from a pre-generated uniform stream
//   ll_bar_kernel    src/log-likelihood.cpp:25-37 for every column
//   ess_kernel       ess(), src/draw-f.cpp:21-60: ONE work-group per item column; the (2+k) passes
//                    of ll_bar are block reductions with a fixed tree, every lane carries the same
//                    bracket state, so control flow is work-group uniform.  HBM/L2 bound.
#include "common.h"
#include "kernels.h"
#include "ll_fast.h"

namespace gpirt {

namespace {

// one term of ll(): as written (library exp and log) or the 70-instruction form of ll_fast.h (FAST; the slice kernels of
// the item-keyed RNG -- an R-stream replay keeps the reference's formula to the letter)
template <bool FAST>
__device__ __forceinline__ double ll_t(double a) { return FAST ? ll_term_fast(a) : ll_term(a); }

__global__ void ll_term_probe_kernel(const double* __restrict__ a, int64_t n, double* __restrict__ out, int fast)
{
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x)
        out[g] = fast == 2 ? ll_term_screen(a[g]) : fast ? ll_term_fast(a[g]) : ll_term(a[g]);
}

__global__ void item_fill_kernel(uint64_t seed, uint32_t iter, uint32_t stage, uint32_t item0,
                                 int64_t n_items, int64_t n_index, double* __restrict__ out,
                                 bool normal)
{
    const int64_t total = n_items * n_index;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t it = g / n_index, ix = g - it * n_index;
        const double u = item_uniform(seed, iter, stage, item0 + (uint32_t)it, (uint32_t)ix);
        out[g] = normal ? qnorm_as241(u) : u;
    }
}

__global__ void rstream_normals_kernel(const double* __restrict__ U, const uint64_t* __restrict__ pos,
                                       int64_t col_stride, int64_t n, int64_t m,
                                       double* __restrict__ out)
{
    const uint64_t p = *pos;
    const int64_t total = n * m;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = g / n, i = g - j * n;
        const uint64_t q = p + (uint64_t)(j * col_stride + 2 * i);
        out[g] = rnorm_from_two(U[q], U[q + 1]);
    }
}

__global__ __launch_bounds__(256) void ll_bar_kernel(const double* __restrict__ f,
                                                     const double* __restrict__ y,
                                                     const double* __restrict__ mu, int64_t n,
                                                     double* __restrict__ out)
{
    __shared__ double red[4];
    const int64_t j = blockIdx.x;
    const double* fj = f + j * n;
    const double* yj = y + j * n;
    const double* mj = mu ? mu + j * n : nullptr;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double yy = yj[i];
        if (yy != yy) continue;
        const double g = mj ? fj[i] + mj[i] : fj[i];
        acc += ll_term(yy * g);
    }
    const double s = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[j] = -s;
}

constexpr int ESS_MAX_TRIALS = 100000;

template <bool FAST>
__global__ __launch_bounds__(256) void ess_kernel(EssArgs a)
{
    __shared__ double red[4];
    const int64_t j = blockIdx.x;
    const int64_t n = a.n;
    double* fj = a.f + j * n;
    const double* nj = a.nu + j * n;
    const double* yj = a.y + j * n;
    const double* mj = a.mu + j * n;
    const bool stream = (a.U != nullptr);
    const uint64_t p0 = stream ? (*a.pos + 2ull * (uint64_t)n) : 0ull;   // after the n normals
    const uint32_t item = a.item0 + (uint32_t)j;
    uint32_t uidx = 0;
    bool overflow = false, nan_state = false;
    auto next_u = [&]() -> double {
        double u;
        if (stream) {
            const uint64_t q = p0 + uidx;
            if (q >= a.cap) { overflow = true; u = 0.5; } else u = a.U[q];
        } else {
            u = item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx);
        }
        ++uidx;
        return u;
    };

    // log_y = ll_bar(f, y, mu) + log(u)                                   draw-f.cpp:28-29
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double yy = yj[i];
        if (yy != yy) continue;
        acc += ll_t<FAST>(yy * (fj[i] + mj[i]));
    }
    const double ll0 = -block_sum_256(acc, red);
    const double u = next_u();
    const double log_y = ll0 + log(u);
    double eps_min = 0.0, eps_max = GP_2PI;                                // :33-34
    double eps = eps_min + (eps_max - eps_min) * next_u();                 // :35
    eps_min = eps - GP_2PI;                                                // :36
    int k = 0;
    double c, s;
    for (;;) {
        c = cos(eps);
        s = sin(eps);
        acc = 0.0;
        for (int64_t i = threadIdx.x; i < n; i += 256) {
            const double yy = yj[i];
            if (yy != yy) continue;
            const double fp = fj[i] * c + nj[i] * s;                       // :43
            acc += ll_t<FAST>(yy * (fp + mj[i]));
        }
        const double llp = -block_sum_256(acc, red);
        if (llp > log_y) break;                                            // :45-47
        if (llp != llp) { nan_state = true; break; }                       // NaN state: never accepts
        if (eps < 0.0) eps_min = eps; else eps_max = eps;                  // :50-55
        if (eps_min == eps_max) eps = eps_min;                             // R::runif(a,a) = a
        else eps = eps_min + (eps_max - eps_min) * next_u();               // :56
        ++k;
        if (k >= ESS_MAX_TRIALS || overflow) { overflow = true; break; }
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += 256) fj[i] = fj[i] * c + nj[i] * s;
    if (threadIdx.x == 0) {
        if (a.k_out) a.k_out[j] = k;
        if (nan_state && a.err) atomicCAS(a.err, 0, (int)GPIRT_E_NUMERIC);
        else if (overflow && a.err) atomicCAS(a.err, 0, stream ? GPIRT_E_RNG : GPIRT_E_NUMERIC);
        if (stream) *a.pos = p0 + uidx;
    }
}

// ---- R-stream replay: three items per pass over L ---------------------------------------------------------------------
// Item j's normals start where item j - 1's slice loop stopped consuming (src/draw-f.cpp:26,56), so nu_j = L z_j cannot be
// STARTED before that loop ends -- but it has few possible values, and a pass over L (268 MB at n = 8192) costs little more
// for 32 right-hand sides than for one.  R's inversion normal takes two consecutive uniforms (src/mvnormal.h:8), so with
// Nrm[r] = rnorm(U[r], U[r + 1])  computed ONCE per iteration for every position r of the window (rs3_begin_kernel), the
// normals of an item that starts at position p are simply Nrm[p + 2 i]: every candidate is a strided window of one array,
// nothing per candidate is ever materialised.
// A pass is anchored at an item a whose start posv[a] is known exactly and serves THREE items (kernels.h, RS3_*):
//   slot 0  item a          1 candidate            start  posv[a]
//   slot 1  item a + 1     15 candidates  c = used(a) in 0..14               start  posv[a] + (2n + 2) + c
//   slot 2  item a + 2     16 candidates  c = used(a) + used(a+1) in 0..15   start  posv[a] + 2 (2n + 2) + c
// (used = uniforms the slice loop consumed behind its first two = its rejection count, src/draw-f.cpp:56): 32 columns = two
// 16-wide MFMA tiles per step of L.  rs3_products_kernel computes the 32 products, rs3_slice_kernel then runs the three
// slice loops one after the other, each on the column its predecessors' counts select, and leaves the next anchor.  A count
// beyond a slot's candidates just ends the pass early -- the next pass is anchored at the first unresolved item -- so there is
// no host round trip and no other path: the host enqueues ceil(m / 3) passes + a few spare ones (a pass that finds every
// item done leaves at once) and looks at the item counter once at the end.
// Why 32 columns and not 48 (slot 2 with sums up to 31): the products kernel takes 23 us + 17 us per tile at n = 8192 (the
// tiles' MFMAs at the pipes' rate: 49 / 57 / 74 us for 2 / 2 with the old epilogue / 3 tiles), and the third tile bought 0.03
// items per pass -- two consecutive slice loops reject 16 times or more between them in 3 % of the cases.
//
// The slice loop itself is evaluated RS3_TRIALS points at a time: a rejected point only moves the bracket end of its own
// sign (src/draw-f.cpp:50-55), so the sequence of trial points is a function of the stream alone, not of the data -- the
// likelihoods of the next eight points are one pass over the rows and ONE meeting of the work-groups instead of eight.

// debug stamps (gpirt_debug_rs_trace): 100 MHz wall clock, one writer per slot
__device__ __forceinline__ void rs_stamp(long long* trace, int idx) { if (trace) trace[idx] = (long long)wall_clock64(); }

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// raw Mersenne-Twister words -> unif_rand() values (R's MT_genrand + fixup, RNG.c): the host only runs the recurrence
__global__ void rs_unpack_kernel(const uint32_t* __restrict__ raw, int64_t count, double* __restrict__ out)
{
    const double i2_32m1 = 2.328306437080797e-10;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)mt_temper(raw[g]) * 2.3283064365386963e-10;
        out[g] = v <= 0.0 ? 0.5 * i2_32m1 : ((1.0 - v) <= 0.0 ? 1.0 - 0.5 * i2_32m1 : v);
    }
}

// Nrm[r] for the positions draw_f can reach from the cursor, the first anchor, the item counter
__global__ __launch_bounds__(256) void rs3_begin_kernel(Rs3Args a, uint64_t span)
{
    const uint64_t p = *a.pos;
    uint64_t end = p + span;
    if (end + 1 > a.cap) end = a.cap > 0 ? a.cap - 1 : 0;      // Nrm[r] needs U[r + 1]
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.anchor[0] = 0; a.anchor[1] = p; a.anchor[2] = end; a.posv[0] = p; }
    for (uint64_t r = p + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < end; r += (uint64_t)gridDim.x * blockDim.x)
        a.Nrm[r] = rnorm_from_two(a.U[r], a.U[r + 1]);
}

// L in the order the candidate products read it: tile (row group rg of 32 rows, column quad kb) = 1 KiB, lane l's two rows of
// column 4 kb + (l >> 4) at doubles 2 l, 2 l + 1 -- the MFMA A operand of one step, so a wave's step is ONE contiguous
// kilobyte and its steps follow each other in memory.  From column-major L the same step touches four 256-byte pieces
// 64 KiB apart (a quarter of a DRAM page each): 3.55 TB/s.  Built once per iteration (a pass over the triangle: ~0.1 ms
// against the ~345 passes that read it); rows and columns past the matrix are zeros; only the tiles a product reads exist.
__global__ __launch_bounds__(256) void rs_tile_kernel(const double* __restrict__ L, int64_t n, int64_t ldl, int64_t nkb, double* __restrict__ Lt)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int64_t rg = blockIdx.y;
    const int64_t r0 = rg * RS_ROWS;
    const int64_t kall = (r0 + RS_ROWS < n) ? r0 + RS_ROWS : n;
    const int64_t kb_end = (kall + 3) / 4;
    const int64_t rp = r0 + 2 * (4 * (i & 3) + (i >> 2));
    for (int64_t kb = (int64_t)blockIdx.x * 4 + wave; kb < kb_end; kb += (int64_t)gridDim.x * 4) {
        const int64_t k = 4 * kb + g;
        double2 v; v.x = 0.0; v.y = 0.0;
        if (k < n) {
            if (rp < n) v.x = L[rp + k * ldl];
            if (rp + 1 < n) v.y = L[rp + 1 + k * ldl];
        }
        *reinterpret_cast<double2*>(Lt + (rg * nkb + kb) * 128 + 2 * lane) = v;
    }
}

// part[by][c][row] = sum over columns [by KC, (by + 1) KC) of L[row][k] z_c[k] for RS_ROWS = 32 rows and the pass's candidate
// columns c (NT tiles of 16): fp64 MFMA 16x16x4 with A = 16 rows x 4 columns of L -- one 16-byte load per lane brings two
// rows (tiles t = 0, 1) -- and B = 4 x 16 candidate normals, lane (i, g) reading  Nrm[start_c + 2 (k + g)]  for its
// candidate c = 16 ct + i: the lanes of a tile touch ~22 consecutive doubles per step, all of them L2 / L1 hits (a pass reads
// 3 (2n + 32) distinct normals).  The four waves of a work-group take a quarter of the part's columns each (a wave's loads are
// a dependent chain of round trips: short chains and many waves are what fills the memory system), eight steps' loads in
// flight; the quarters meet in LDS and are added in order.  The strict upper triangle of L holds zeros (gpirt_sampler_create).
template <int NT>
__device__ __forceinline__ void rs3_product_block(const Rs3Args& a, const uint64_t base, const int bx, const int by, double* red /* 8 NT x 64 doubles */)
{
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int64_t n = a.n;
    const int64_t r0 = (int64_t)bx * RS_ROWS;
    const int64_t kall = (r0 + RS_ROWS < n) ? r0 + RS_ROWS : n;       // columns that can be non-zero in these rows
    if ((int64_t)by * RS_KC >= kall) return;                  // (uniform over the work-group)
    const int64_t k_beg = (int64_t)by * RS_KC + (int64_t)kq * (RS_KC / 4);
    int64_t k_end = k_beg + RS_KC / 4;
    if (k_end > kall) k_end = kall;
    // The instruction returns row (g + 4 r) of the 16-row tile in element r of lane (i, g): A-lane i carries row pair
    // pi(i) = 4 (i & 3) + (i >> 2) (rs_tile_kernel), so that a lane ends up with EIGHT CONSECUTIVE rows, r0 + 8 g + 2 r + t
    // (solve64.h uses the same permutation).  Rows >= n are zeros in the tiles and are not stored.
    const double* Lp = a.Lt + ((int64_t)bx * a.nkb) * 128 + 2 * lane;      // tile (bx, kb) at + 128 kb (rs_tile_kernel)
    const uint64_t item_step = 2ull * (uint64_t)n + 2ull;
    const double* Zp[NT];
    Zp[0] = a.Nrm + (i == 0 ? base : base + item_step + (uint64_t)(i - 1)) + 2 * g;
    if (NT > 1) {
#pragma unroll
        for (int ct = 1; ct < NT; ++ct) Zp[ct] = a.Nrm + base + 2ull * item_step + (uint64_t)(16 * (ct - 1) + i) + 2 * g;
    }
    d4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[t][ct] = d4{ 0.0, 0.0, 0.0, 0.0 };
    int64_t k = k_beg;
    for (; k + 16 <= k_end; k += 16) {
        double2 av[4]; double b[4][NT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = *reinterpret_cast<const double2*>(Lp + ((k >> 2) + u) * 128);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) b[u][ct] = Zp[ct][2 * (k + 4 * u)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                acc[0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].x, b[u][ct], acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].y, b[u][ct], acc[1][ct], 0, 0, 0);
            }
    }
    for (; k < k_end; k += 4) {
        const double2 av = *reinterpret_cast<const double2*>(Lp + (k >> 2) * 128);      // (past the matrix: zeros in the tile)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const double bb = Zp[ct][2 * k];
            acc[0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bb, acc[0][ct], 0, 0, 0);
            acc[1][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bb, acc[1][ct], 0, 0, 0);
        }
    }
    // quarters 1, 2, 3 are added to quarter 0 in that order, one at a time through LDS
    double* mine = red + lane;
    for (int q = 1; q < 4; ++q) {
        __syncthreads();
        if (kq == q) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((t * NT + ct) * 4 + r) * 64] = acc[t][ct][r];
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][ct][r] += mine[((t * NT + ct) * 4 + r) * 64];
        }
    }
    if (kq != 0) return;
    // lane (i, g): acc[t][ct][r] = row r0 + 2 pi(g + 4 r) + t = r0 + 8 g + 2 r + t, candidate 16 ct + i
    double* out = a.part + ((int64_t)by * RS3_CAND) * n;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        double* oc = out + (int64_t)(16 * ct + i) * n + r0 + 8 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t row = r0 + 8 * g + 2 * r + t;
                if (row < n) oc[2 * r + t] = acc[t][ct][r];
            }
    }
}

// The same product for a FULL part (every wave has RS_KC / 16 steps), the form 9 of 10 work-groups run.  The candidates'
// normals come from LDS: the part's columns need three windows of Nrm -- slot 0: every other normal from the anchor on,
// slots 1 / 2: 2 RS_KC + 14 / + 15 consecutive normals -- which the work-group stages once (21 KB); the B operand of a step is
// then one conflict-free ds_read_b64 per tile (the lanes of a tile touch 22 consecutive doubles) instead of gathers through
// the vector memory path that had to be held in registers a batch ahead.  That leaves the registers to L: a ring of
// RS3_RING steps (1 KiB each) per wave, refilled as it is consumed -- every load address is known up front, nothing is
// conditional, so the compiler's vmcnt bookkeeping lets step s start the moment ITS kilobyte has arrived while 11 more are in
// flight behind it (the first form issued eight, waited for all, and only then computed).
constexpr int RS3_RING = 10;       // (8 / 12 / 16 measure the same at four work-groups per CU; the four waves taking the part's steps in
                                   //  turn, one 128 KB stream per work-group instead of four of 32 KB, too: gpurun_out/r5h, r5i.  10 = what
                                   //  fits the 96 registers of FIVE work-groups per CU: 29.7 ms of draw_f at 8192 x 1024 against 30.8 with
                                   //  four; six / seven / eight with rings of 8 / 6 / 4: 30.2-30.3 / 30.8 / 30.2)
template <int NT>
__device__ __forceinline__ void rs3_product_full(const Rs3Args& a, const uint64_t base, const int bx, const int by, double* lds,
                                                 const double* Lp, long long* tr)
{
    constexpr int STEPS = RS_KC / 16;                         // per wave
    constexpr int TS = 128;                                   // doubles between a wave's consecutive steps
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int64_t n = a.n;
    const int64_t r0 = (int64_t)bx * RS_ROWS;
    const int64_t k0 = (int64_t)by * RS_KC;
    // the windows: W0[kk] = z_slot0[k0 + kk];  W1[x] / W2[x] = the normals from slot 1's / slot 2's first candidate's z[k0] on.
    // Their loads go out together ...
    const uint64_t item_step = 2ull * (uint64_t)n + 2ull;
    double* W0 = lds; double* W1 = lds + RS_KC; double* W2 = W1 + 2 * RS_KC + 16;
    static_assert(RS_KC % 256 == 0, "the slot-0 window is staged 256 entries at a time");
    constexpr int C0 = RS_KC / 256, C1 = (2 * RS_KC + 16 + 255) / 256, C2 = (2 * RS_KC + RS3_C2 + 255) / 256;
    const double* N0 = a.Nrm + base + 2ull * (uint64_t)k0;
    const double* N1 = N0 + item_step;
    const double* N2 = N1 + item_step;
    double w0[C0], w1[C1], w2[NT > 1 ? C2 : 1];
#pragma unroll
    for (int q = 0; q < C0; ++q) w0[q] = N0[2 * (threadIdx.x + 256 * q)];
#pragma unroll
    for (int q = 0; q < C1; ++q) { const int x = threadIdx.x + 256 * q; w1[q] = N1[x < 2 * RS_KC + 16 ? x : 0]; }
    if (NT > 1) {
#pragma unroll
        for (int q = 0; q < C2; ++q) { const int x = threadIdx.x + 256 * q; w2[q] = N2[x < 2 * RS_KC + RS3_C2 ? x : 0]; }
    }
    // ... and the ring's first turn straight behind them: loads return in order, so the windows (L2 hits) are in LDS while
    // the kilobytes of L are still on their way (the other way round the windows queued behind 64 KB from HBM: 3.6-6 us)
    double2 av[RS3_RING];
#pragma unroll
    for (int u = 0; u < RS3_RING; ++u) av[u] = *reinterpret_cast<const double2*>(Lp + u * TS);
#pragma unroll
    for (int q = 0; q < C0; ++q) W0[threadIdx.x + 256 * q] = w0[q];
#pragma unroll
    for (int q = 0; q < C1; ++q) { const int x = threadIdx.x + 256 * q; if (x < 2 * RS_KC + 16) W1[x] = w1[q]; }
    if (NT > 1) {
#pragma unroll
        for (int q = 0; q < C2; ++q) { const int x = threadIdx.x + 256 * q; if (x < 2 * RS_KC + RS3_C2) W2[x] = w2[q]; }
    }
    __syncthreads();
    rs_stamp(tr, 2);                                          // windows staged
    // lane (i, g) at the wave's step s reads column kk = kq * RS_KC / 4 + 4 s + g of the part
    const int kk0 = kq * (RS_KC / 4) + g;
    const double* B0 = (i == 0) ? W0 + kk0 : W1 + 2 * kk0 + (i - 1);
    const int st0 = (i == 0) ? 4 : 8;                         // doubles per step
    constexpr int st1 = 8;
    const double* B1 = W2 + 2 * kk0 + i;
    d4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[t][ct] = d4{ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int u = s % RS3_RING;
        const double2 al = av[u];
        if (s + RS3_RING < STEPS) av[u] = *reinterpret_cast<const double2*>(Lp + (s + RS3_RING) * TS);
        double b[NT];
        b[0] = B0[s * st0];
        if (NT > 1) {
#pragma unroll
            for (int ct = 1; ct < NT; ++ct) b[ct] = B1[s * st1 + 16 * (ct - 1)];
        }
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            acc[0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(al.x, b[ct], acc[0][ct], 0, 0, 0);
            acc[1][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(al.y, b[ct], acc[1][ct], 0, 0, 0);
        }
    }
    // The four quarters meet in LDS (the windows' space is free behind the barrier) and ALL 256 threads add them, in the order
    // ((q0 + q1) + q2) + q3, and store: thread x takes (one row of a parity, candidate x >> 4 + 16 p): coalesced down to
    // every other row of a candidate's 256-byte segment, the other half follows in the second phase.  (Wave 0 alone adding its own 24 accumulators and storing them lane by lane -- 64 lines
    // touched per store instruction, 43 registers spilled around the additions -- took 5 to 16 us per work-group, as long
    // as its MFMAs: in-kernel stamps, tools/rs_trace.py.)
    rs_stamp(tr, 3);                                          // this wave's MFMAs issued
    // ... in two phases, the even rows of the group (tile t = 0), then the odd ones: half the LDS (16.6 KB, less than the
    // windows' 20.7), which is what lets a fifth work-group onto every compute unit
    double* out = a.part + ((int64_t)by * RS3_CAND) * n + r0;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        __syncthreads();
        if (tt == 0) rs_stamp(tr, 4);
        double* mine = lds + (size_t)kq * RS3_QSTRIDE + lane;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(ct * 4 + r) * 65] = acc[tt][ct][r];      // (65: the readers' bank spread)
        __syncthreads();
        // lane (i, g) of a wave held acc[tt][ct][r] = row 8 g + 2 r + tt of the group, candidate 16 ct + i
#pragma unroll
        for (int p = 0; p < (16 * NT) / 16; ++p) {
            const int x = threadIdx.x + 256 * p;
            const int row = 2 * (x & 15) + tt, c = x >> 4;
            const int gg = row >> 3, rr = (row >> 1) & 3, ii = c & 15, cc = c >> 4;
            const double* src = lds + (cc * 4 + rr) * 65 + ii + 16 * gg;
            const double v = ((src[0] + src[RS3_QSTRIDE]) + src[2 * RS3_QSTRIDE]) + src[3 * RS3_QSTRIDE];
            if (r0 + row < n) out[(int64_t)c * n + row] = v;
        }
    }
    rs_stamp(tr, 5);
}

// one work-group per unit of the table the sampler built (rs3_unit_table): full parts first, then the ragged ones.  The
// anchor is ONE 32-byte record (item, start, end of Nrm): one scalar load, in flight together with the unit's.
__global__ __launch_bounds__(256, 5) void rs3_products_kernel(Rs3Args a)
{
    __shared__ double lds[RS3_LDS_DOUBLES];
    const uint64_t item0 = a.anchor[0], base = a.anchor[1];   // (both scalar loads of the prologue go out together)
    const uint32_t unit = a.units[blockIdx.x];
    const int bx = (int)(unit & 0xffffu), by = (int)(unit >> 16);
    const bool full = (int)blockIdx.x < a.nfull;
    const double* Lp = a.Lt + ((int64_t)bx * a.nkb + (((int64_t)by * RS_KC) >> 2) + (int64_t)(threadIdx.x >> 6) * (RS_KC / 16)) * 128 + 2 * (threadIdx.x & 63);
    long long* tr = (a.trace && threadIdx.x == 0 && (blockIdx.x == 0 || (int)blockIdx.x == a.nfull / 2 || (int)blockIdx.x == a.nfull - 1))
                        ? a.trace + 64 + 8 * (blockIdx.x == 0 ? 0 : (int)blockIdx.x == a.nfull / 2 ? 1 : 2) : nullptr;
    rs_stamp(tr, 0);
    if (item0 >= (uint64_t)a.m) return;                       // every item is done (or the draw has failed): a spare pass
    rs_stamp(tr, 1);
    const bool three = (uint64_t)a.m - item0 >= 3;            // (the last items: slots 0 and 1 alone)
    if (full) {
        if (three) rs3_product_full<RS3_NT>(a, base, bx, by, lds, Lp, tr);
        else       rs3_product_full<1>(a, base, bx, by, lds, Lp, tr);
    } else {
        if (three) rs3_product_block<RS3_NT>(a, base, bx, by, lds);
        else       rs3_product_block<1>(a, base, bx, by, lds);
    }
}

// ess() (src/draw-f.cpp:21-60) for the items of a pass, one after the other, the formula as written (no ll_fast: an R-stream
// replay keeps the reference's arithmetic).  What bounds it is latency -- a slot is a chain of memory round trips, one
// sin / cos, one likelihood term and one meeting of the work-groups -- so the work is spread WIDE: E <= 256 work-groups of
// 32 R rows (R = 2 at n = 8192: 128 work-groups), five waves each:
//   waves 0..3  lane (row rl = lane & 31, half h = lane >> 5): trial point t = 2 wave + h of the round for its R rows -- the
//               eight points of a round are evaluated side by side, ONE term per thread and row;
//   wave 0      (h = 0) also the current state's ll_bar (first round);     wave 4  lanes 0..7: cos / sin of the eight points.
// The next item's f, mu, y are loaded while the current one is evaluated (they depend on nothing); nu = the parts of the
// selected column, eight groups of parts summed side by side (fixed order); the uniforms a round can consume are fetched
// TOGETHER into LDS before the bracket sequence is walked (read one by one from the 150 MB window, each at an address the one
// before decides -- :56 consumes nothing when the bracket has closed -- they were ten dependent HBM misses per item).
// Meeting (sync): every work-group stores its <= 9 partial sums write-through at agent scope, waits for the stores, raises
// ITS OWN flag word to the meeting's tag (tags grow monotonically over passes and draws: nothing is ever reset); thread q
// polls work-group q's flag (bounded, like every in-kernel wait of this library: on expiry the draw fails with GPIRT_E_HIP)
// and reads its sums behind it; then all partial sums are added in one fixed order -- every work-group sees the same bits
// and takes the same branches.  No atomics: 256 adds to one counter would queue for longer than the rest of the meeting.
// (Measured and not kept: each sum as ONE 16-byte record {value, tag ^ bits(value)} polled directly, no flag and no wait
// for the stores -- fewer round trips on paper, 5.6 instead of 3.8 us per meeting: gpurun_out/r5o.)
// The work-groups are the whole grid of a launch on an otherwise idle stream, <= one per CU: resident together.
template <int R>
__global__ __launch_bounds__(320) void rs3_slice_kernel(Rs3Args a)
{
    constexpr int T = RS3_TRIALS, V = T + 1, RW = 32 * R;
    __shared__ double ul[T + 2];
    __shared__ double cs[2 * T];
    __shared__ double psum[8][RW];
    __shared__ double vals[RS3_MAX_WGS * V];
    __shared__ double csum[V * 16];
    __shared__ double tots[V];
    __shared__ int expired;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, rl = lane & 31, hh = lane >> 5;
    const int w = blockIdx.x, E = gridDim.x;
    const int64_t n = a.n;
    if (a.anchor[0] >= (uint64_t)a.m) return;                 // every item is done (or the draw has failed)
    const int item0 = (int)a.anchor[0];
    const int ns = ((int)a.m - item0 < RS3_SLOTS) ? (int)a.m - item0 : RS3_SLOTS;
    const uint64_t nrm_end = a.anchor[2];
    uint64_t start = a.anchor[1];
    const int64_t i0 = (int64_t)w * RW + rl;                  // this thread's rows: i0 + 32 e
    // parts of this work-group's rows: group gq = 2 wave + h (waves 0..3) adds parts [gq pp, (gq + 1) pp)
    const int64_t last_row = ((int64_t)(w + 1) * RW < n ? (int64_t)(w + 1) * RW : n) - 1;
    const int maxparts = (int)((((last_row / RS_ROWS) + 1) * RS_ROWS < n ? ((last_row / RS_ROWS) + 1) * RS_ROWS : n) + RS_KC - 1) / RS_KC;
    const int pp = (maxparts + 7) / 8;
    int usum = 0, resolved = 0, sync = 0, fail = 0;
    if (tid == 0) expired = 0;
    long long* tr = (a.trace && tid == 0 && (w == 0 || w == E - 1)) ? a.trace + (w == 0 ? 0 : 32) : nullptr;      // first and last work-group
    int ti = 0;
    rs_stamp(tr, ti++);
    double Fn[R], Mn[R], Yn[R];                               // the NEXT slot's rows
    auto fetch_rows = [&](const int64_t j) {
#pragma unroll
        for (int e = 0; e < R; ++e) {
            const int64_t i = i0 + 32 * e;
            Fn[e] = 0.0; Mn[e] = 0.0; Yn[e] = __builtin_nan("");             // NaN = skipped, like a missing response
            if (i < n) { Fn[e] = a.f[j * n + i]; Mn[e] = a.mu[j * n + i]; Yn[e] = a.y[j * n + i]; }
        }
    };
    fetch_rows(item0);
    for (int g = 0; g < ns; ++g) {
        int col = 0;
        if (g == 1) { if (usum >= a.lim1) break; col = 1 + usum; }
        if (g == 2) { if (usum >= a.lim2) break; col = 16 + usum; }
        const int64_t j = item0 + g;
        // the item's 2n uniforms (all its normals were built) and its first two slice uniforms lie inside the window
        if (start + 2ull * (uint64_t)n + 2ull > a.cap || start + 2ull * (uint64_t)(n - 1) >= nrm_end) { fail = GPIRT_E_RNG; break; }
        const uint64_t p0 = start + 2ull * (uint64_t)n;                       // behind the n normals
        uint32_t uidx = 0, ubase = 0;
        rs_stamp(tr, ti++);                                                    // slot start
        __syncthreads();                                                       // (ul, psum, cs are still being read from the slot before)
        // the uniforms of the first round: u, the first point and one per rejection
        if (tid < T + 2) { const uint64_t q = p0 + tid; ul[tid] = q < a.cap ? a.U[q] : __builtin_nan(""); }
        if (wv < 4) {
            const int gq = 2 * wv + hh;
#pragma unroll
            for (int e = 0; e < R; ++e) {
                const int64_t i = i0 + 32 * e;
                double v = 0.0;
                if (i < n) {
                    const int64_t grp = (i / RS_ROWS) * RS_ROWS;
                    const int64_t kall = (grp + RS_ROWS < n) ? grp + RS_ROWS : n;
                    const int parts = (int)((kall + RS_KC - 1) / RS_KC);
                    int q1 = (gq + 1) * pp; if (q1 > parts) q1 = parts;
                    for (int q0 = gq * pp; q0 < q1; q0 += 4) {    // (four loads in flight, not four round trips)
                        double t[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int64_t at = ((int64_t)(q0 + u) * RS3_CAND + col) * n + i;
                            t[u] = (q0 + u < q1) ? a.part[at] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) v += t[u];
                    }
                }
                psum[gq][32 * e + rl] = v;
            }
        }
        double F[R], Vn[R], M[R], Y[R];
#pragma unroll
        for (int e = 0; e < R; ++e) { F[e] = Fn[e]; M[e] = Mn[e]; Y[e] = Yn[e]; }
        if (g + 1 < ns) fetch_rows(j + 1);
        __syncthreads();
        rs_stamp(tr, ti++);                                                    // loads in
#pragma unroll
        for (int e = 0; e < R; ++e) {
            double v = psum[0][32 * e + rl];
#pragma unroll
            for (int q = 1; q < 8; ++q) v += psum[q][32 * e + rl];
            Vn[e] = v;
        }
        bool bad_u = false;
        auto next_u = [&]() -> double {
            const double x = ul[uidx - ubase];
            ++uidx;
            if (x != x) { bad_u = true; return 0.5; }        // (past the window: reported if the draw gets that far)
            return x;
        };
        const double u = next_u();                                             // draw-f.cpp:28
        double eps_min = 0.0, eps_max = GP_2PI;                                // :33-34
        double eps = eps_min + (eps_max - eps_min) * next_u();                 // :35
        eps_min = eps - GP_2PI;                                                // :36
        int k = 0;
        double log_y = 0.0, c = 1.0, sn = 0.0;
        uint32_t uacc = 0;
        bool first = true, done = false;
        while (!done) {
            if (!first) {
                // a further round: the T uniforms it can consume
                ubase = uidx;
                __syncthreads();
                if (tid < T) { const uint64_t q = p0 + ubase + tid; ul[tid] = q < a.cap ? a.U[q] : __builtin_nan(""); }
                __syncthreads();
            }
            // the next T trial points: each is what :50-56 makes of the one before, were it rejected
            // (the uniforms are read from LDS all at once, as if every rejection consumed one -- true unless the bracket has
            // closed, eps_min == eps_max, where :56 consumes nothing: then the walk is redone with dependent reads)
            uint32_t Ut[T]; bool Bt[T];
            double my_eps = 0.0;
            {
                double uu[T];
#pragma unroll
                for (int t = 0; t < T; ++t) uu[t] = ul[uidx - ubase + t];
                double e0 = eps, emin = eps_min, emax = eps_max;
                bool closed = false, badf = bad_u;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if (lane == t) my_eps = e0;
                    Ut[t] = uidx + t; Bt[t] = badf;
                    if (e0 < 0.0) emin = e0; else emax = e0;                   // :50-55
                    if (emin == emax) closed = true;
                    double x = uu[t];
                    if (x != x) { badf = true; x = 0.5; }
                    e0 = emin + (emax - emin) * x;                             // :56
                }
                if (!closed) { eps = e0; eps_min = emin; eps_max = emax; uidx += T; bad_u = badf; }
                else {
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if (lane == t) my_eps = eps;
                        Ut[t] = uidx; Bt[t] = bad_u;
                        if (eps < 0.0) eps_min = eps; else eps_max = eps;      // :50-55
                        if (eps_min == eps_max) eps = eps_min;                 // R::runif(a, a) = a, nothing consumed
                        else eps = eps_min + (eps_max - eps_min) * next_u();   // :56
                    }
                }
            }
            double mine = 0.0, mine0 = 0.0;                                    // this thread's term sums: its trial point / ll_bar(f)
            // cos and sin of the eight points on two waves, side by side; ll_bar(f) on a third meanwhile
            // (measured and not kept: both taken from a table that one more work-group of the products kernel fills for every
            //  candidate's first eight points -- 0.65 us less in front of the terms and as much more waiting in the meeting,
            //  30.6 ms of draw_f either way: the meeting waits for the slowest of 256 work-groups, not for this phase)
            if (wv == 4) { if (lane < T) cs[lane] = cos(my_eps); }
            else if (wv == 3) { if (lane < T) cs[T + lane] = sin(my_eps); }
            else if (first && wv == 0 && hh == 0) {
#pragma unroll
                for (int e = 0; e < R; ++e) if (Y[e] == Y[e]) mine0 += ll_term(Y[e] * (F[e] + M[e]));          // :29
            }
            __syncthreads();
            rs_stamp(tr, ti++);                                                // sequence, cos / sin, ll_bar(f)
            // The meeting: partial sums stored write-through at agent scope and waited for, the work-group's barrier, its flag
            // word raised to the meeting's tag; thread q polls work-group q's flag and reads its nine sums straight behind it.
            const uint64_t tag = a.tag + (uint64_t)sync;
            double* rec = a.partial + (size_t)(sync & 1) * RS3_MAX_WGS * V;
            unsigned long long* flg = a.flags + (size_t)(sync & 1) * RS3_MAX_WGS;
            if (wv < 4) {
                const int t = 2 * wv + hh;
                const double ct = cs[t], st = cs[T + t];
#pragma unroll
                for (int e = 0; e < R; ++e) if (Y[e] == Y[e]) mine += ll_term(Y[e] * ((F[e] * ct + Vn[e] * st) + M[e]));   // :43
                // sums over the 32 lanes of a half (fixed tree)
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) { mine += __shfl_down(mine, off, 64); mine0 += __shfl_down(mine0, off, 64); }
                if (rl == 0) __hip_atomic_store(rec + w * V + 1 + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (first && tid == 0) __hip_atomic_store(rec + w * V, mine0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            rs_stamp(tr, ti++);                                                // terms, partial sums stored
#ifdef GPIRT_PANEL_FENCES
            // the fenced reference form of the meeting (`make fences`, tests/test_gpu_fences.py): an agent-scope release in
            // front of the flag, an acquire behind the poll -- the draws must be the default build's bit for bit
            if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
            if (tid == 0) __hip_atomic_store(flg + w, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < E) {
                int spins = 0;
                unsigned long long seen = __hip_atomic_load(flg + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (seen < tag && ++spins < (1 << 22)) {
                    __builtin_amdgcn_s_sleep(1);
                    seen = __hip_atomic_load(flg + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (seen < tag) expired = 1;
#ifdef GPIRT_PANEL_FENCES
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                // work-group tid's sums, straight behind its flag (nine loads in flight)
                double t9[V];
#pragma unroll
                for (int v = 0; v < V; ++v) t9[v] = __hip_atomic_load(rec + tid * V + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int v = 0; v < V; ++v) vals[tid * V + v] = t9[v];
            }
            __syncthreads();
            rs_stamp(tr, ti++);                                                // every flag seen, every sum read
            if (expired) { fail = GPIRT_E_HIP; break; }
            if (tid < V * 16) {
                const int v = tid >> 4, ch = tid & 15, per = (E + 15) / 16;
                int q1 = (ch + 1) * per; if (q1 > E) q1 = E;
                double r = 0.0;
                for (int q = ch * per; q < q1; ++q) r += vals[q * V + v];
                csum[tid] = r;
            }
            __syncthreads();
            if (tid < V) {
                double r = csum[tid * 16];
                for (int ch = 1; ch < 16; ++ch) r += csum[tid * 16 + ch];
                tots[tid] = r;
            }
            __syncthreads();
            double tot[V];
#pragma unroll
            for (int v = 0; v < V; ++v) tot[v] = tots[v];
            rs_stamp(tr, ti++);                                                // sums read and added
            ++sync;
            if (first) { log_y = -tot[0] + log(u); first = false; }            // :29
            int hit = -1;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (hit < 0 && fail == 0) {
                    const double llp = -tot[1 + t];
                    if (llp > log_y) hit = t;                                  // :45-47
                    else if (llp != llp) fail = GPIRT_E_NUMERIC;               // NaN state: never accepts
                }
            }
            if (fail) break;
            if (hit >= 0) {
                bool bad = false;
#pragma unroll
                for (int t = 0; t < T; ++t) if (t == hit) { c = cs[t]; sn = cs[T + t]; uacc = Ut[t]; bad = Bt[t]; }
                if (bad) { fail = GPIRT_E_RNG; break; }
                k += hit; done = true;
            } else {
                k += T;
                if (bad_u) { fail = GPIRT_E_RNG; break; }
                if (k >= ESS_MAX_TRIALS) { fail = GPIRT_E_NUMERIC; break; }
            }
        }
        if (fail) break;
        if (wv == 0 && hh == 0) {
#pragma unroll
            for (int e = 0; e < R; ++e) {
                const int64_t i = i0 + 32 * e;
                if (i < n) a.f[j * n + i] = F[e] * c + Vn[e] * sn;
            }
        }
        start = p0 + uacc;
        usum += (int)uacc - 2;
        ++resolved;
        if (w == 0 && tid == 0) { a.k_out[j] = k; a.posv[j + 1] = start; }
    }
    rs_stamp(tr, ti++);
    if (tr && w == 0) a.trace[63] = ti;
    if (tid == 0) {
        if (fail) { atomicCAS(a.err, 0, fail); a.anchor[0] = (uint64_t)a.m; }        // every later kernel of the draw leaves at once
        else if (w == 0) { a.anchor[0] = (uint64_t)(item0 + resolved); a.anchor[1] = start; *a.pos = start; }
    }
}

// Register-resident variant for n <= NTH * EPT: each lane keeps its EPT entries of f, nu, mu and y in
// registers, so the (2 + k) likelihood passes of a column touch memory once; arithmetic is identical
// to ess_kernel (same per-element expression, same reduction tree).
// FOLD (n up to 16384: 16 entries per lane of 1024, four wavefronts per SIMD at 128 registers): y is +-1 or NaN, so y ((f c + nu s) + mu) = ((y f) c + (y nu) s) + (y mu)
// bit for bit (a sign change is exact and rounding is symmetric) -- the lane keeps THREE arrays, y f, y nu, y mu (NaN in
// y mu = a missing response), 96 registers instead of 128, and reads f and nu once more at the end to write f'.
template <int EPT, int NTH, bool FAST, bool FOLD = false>
__global__ __launch_bounds__(NTH) void ess_kernel_reg(EssArgs a)
{
    __shared__ double red[16];
    auto block_sum = [&](double v) {
        if (NTH == 256) return block_sum_256(v, red);
        if (NTH == 512) return block_sum_512(v, red);
        // 1024 threads: sixteen wavefronts, fixed tree
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) +
               (((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15])));
    };
    const int64_t j = blockIdx.x;
    const int64_t n = a.n;
    double* fj = a.f + j * n;
    const double* nj = a.nu + j * n;
    const double* yj = a.y + j * n;
    const double* mj = a.mu + j * n;
    const uint32_t item = a.item0 + (uint32_t)j;
    double F[EPT], V[EPT], M[EPT], Y[EPT];          // (FOLD: Y stays unused and costs no register)
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = threadIdx.x + NTH * e;
        const bool in = i < n;
        const double yy = in ? yj[i] : __builtin_nan("");      // NaN = skipped, like a missing response
        F[e] = in ? fj[i] : 0.0;
        V[e] = in ? nj[i] : 0.0;
        M[e] = in ? mj[i] : 0.0;
        Y[e] = FOLD ? 0.0 : yy;
        if (FOLD) { F[e] *= yy; V[e] *= yy; M[e] *= yy; }
    }
    // the argument of one term, and whether the row counts.  (Macros, not lambdas: a lambda that captures the arrays by
    // reference put them into scratch memory -- 164 registers + 432 bytes of scratch per lane instead of 237 registers, and the
    // kernel ran 330-450 instead of 140-205 us at 8192 x 1024.)
#define ESS_ARG0(e) (FOLD ? F[e] + M[e] : Y[e] * (F[e] + M[e]))
#define ESS_ARGP(e, c_, s_) (FOLD ? (F[e] * (c_) + V[e] * (s_)) + M[e] : Y[e] * ((F[e] * (c_) + V[e] * (s_)) + M[e]))
#define ESS_LIVE(e) (FOLD ? M[e] == M[e] : Y[e] == Y[e])
    uint32_t uidx = 0;
    double acc = 0.0;
    // ll(f) of the current state, the other half of the slice level (:28-29) -- in full precision when it is needed: with the
    // screen on, a trial point is decided by the DIFFERENCE of two screened sums wherever that difference is further from
    // log(u) than both error bands together, and only a trial point inside that band costs the two full-precision passes
    // (this one at most once per item).  The decision taken there is the expression below, bit for bit.
    auto exact_ll0 = [&]() {
        double t = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (ESS_LIVE(e)) t += ll_t<FAST>(ESS_ARG0(e));
        return -block_sum(t);
    };
    // the screen's error bound for this item's sums (ll_fast.h): every row could be off by LL_SCREEN_ERR
    const double band = LL_SCREEN_ERR * (double)n;
    double ll0 = 0.0, lls0 = 0.0;
    bool have_ll0 = false;
    // (the written form of the term overflows to +inf beyond |a| = 709.8 where the screen does not: with it, ll(f) is always
    // taken in full precision, so that an infinite slice level stays one)
    const bool lazy = a.screen && FAST;
    if (lazy) {
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (ESS_LIVE(e)) acc += ll_term_screen(ESS_ARG0(e));
        lls0 = -block_sum(acc);
    } else {
        ll0 = exact_ll0();
        have_ll0 = true;
    }
    const double u = item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx++);
    const double log_u = log(u);
    double eps_min = 0.0, eps_max = GP_2PI;
    double eps = eps_min + (eps_max - eps_min) * item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx++);
    eps_min = eps - GP_2PI;                                                // :36
    int k = 0;
    bool bad = false;
    double c, s;
    for (;;) {
        c = cos(eps);
        s = sin(eps);
        // :45 needs the sign of ll_bar(f') - log_y only
        int verdict = 0;                                                   // +1 accept, -1 reject, 0 undecided
        if (a.screen) {
            acc = 0.0;
            // (the written form of the term, !FAST, is +inf below a = -709.78 where the screen stays finite: a trial point
            // with such a row is rejected by the formula as written, so the screen may not decide it)
            int overflows = 0;
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (ESS_LIVE(e)) {
                    const double arg = ESS_ARGP(e, c, s);
                    acc += ll_term_screen(arg);
                    if (!FAST && arg < -709.0) overflows = 1;
                }
            const double lls = -block_sum(acc);
            if (!have_ll0) {
                const double d = lls - lls0;
                if (d - 2.0 * band > log_u) verdict = 1;
                else if (d + 2.0 * band < log_u) verdict = -1;
            } else {
                const double log_y = ll0 + log_u;
                if (lls - band > log_y) verdict = 1;
                else if (lls + band < log_y) verdict = -1;
            }
            if (!FAST) { if (__syncthreads_or(overflows)) verdict = 0; }
        }
        if (verdict == 0) {
            if (!have_ll0) { ll0 = exact_ll0(); have_ll0 = true; }
            const double log_y = ll0 + log_u;                              // draw-f.cpp:28-29
            acc = 0.0;
#pragma unroll
            for (int e = 0; e < EPT; ++e)
                if (ESS_LIVE(e)) acc += ll_t<FAST>(ESS_ARGP(e, c, s));   // :43
            const double llp = -block_sum(acc);
            if (llp > log_y) verdict = 1;                                  // :45-47
            else if (llp != llp || ll0 != ll0) { bad = true; break; }
        }
        if (verdict > 0) break;
        if (eps < 0.0) eps_min = eps; else eps_max = eps;                  // :50-55
        if (eps_min == eps_max) eps = eps_min;
        else eps = eps_min + (eps_max - eps_min) * item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx++);
        ++k;
        if (k >= ESS_MAX_TRIALS) { bad = true; break; }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = threadIdx.x + NTH * e;
        if (FOLD) { if (i < n) fj[i] = fj[i] * c + nj[i] * s; }       // (the lane's own entries: read back, the folded copies may be NaN)
        else if (i < n) fj[i] = F[e] * c + V[e] * s;
    }
    if (threadIdx.x == 0) {
        if (a.k_out) a.k_out[j] = k;
        if (bad && a.err) atomicCAS(a.err, 0, (int)GPIRT_E_NUMERIC);
    }
}

#undef ESS_ARG0
#undef ESS_ARGP
#undef ESS_LIVE

__global__ void advance_pos_kernel(uint64_t* pos, uint64_t delta) { *pos += delta; }

}  // namespace

int launch_item_uniforms(hipStream_t stream, uint64_t seed, uint32_t iter, uint32_t stage,
                         uint32_t item0, int64_t n_items, int64_t n_index, double* out, bool normal)
{
    const int64_t total = n_items * n_index;
    if (total <= 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(item_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed, iter,
                       stage, item0, n_items, n_index, out, normal);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rstream_normals(hipStream_t stream, const double* U, const uint64_t* d_pos,
                           int64_t col_stride, int64_t n, int64_t m, double* out)
{
    const int64_t total = n * m;
    if (total <= 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rstream_normals_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, U, d_pos,
                       col_stride, n, m, out);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_ll_bar(hipStream_t stream, const double* f, const double* y, const double* mu, int64_t n,
                  int64_t m, double* out)
{
    if (m <= 0) return 0;
    hipLaunchKernelGGL(ll_bar_kernel, dim3((unsigned)m), dim3(256), 0, stream, f, y, mu, n, out);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_ess(hipStream_t stream, const EssArgs& a)
{
    if (a.m <= 0) return 0;
    // a.ll_exact (GPIRT_LL_EXACT=1): log(1 + exp(-a)) through the library in every mode
    const bool fast = a.U == nullptr && a.ll_exact != 1;
    if (a.U == nullptr && a.n <= 256 * 8) {
        if (fast) hipLaunchKernelGGL((ess_kernel_reg<8, 256, true>), dim3((unsigned)a.m), dim3(256), 0, stream, a);
        else      hipLaunchKernelGGL((ess_kernel_reg<8, 256, false>), dim3((unsigned)a.m), dim3(256), 0, stream, a);
    } else if (a.U == nullptr && a.n <= 256 * 32) {
        if (fast) hipLaunchKernelGGL((ess_kernel_reg<16, 512, true>), dim3((unsigned)a.m), dim3(512), 0, stream, a);
        else      hipLaunchKernelGGL((ess_kernel_reg<16, 512, false>), dim3((unsigned)a.m), dim3(512), 0, stream, a);
    } else if (a.U == nullptr && a.n <= 1024 * 16) {
        if (fast) hipLaunchKernelGGL((ess_kernel_reg<16, 1024, true, true>), dim3((unsigned)a.m), dim3(1024), 0, stream, a);
        else      hipLaunchKernelGGL((ess_kernel_reg<16, 1024, false, true>), dim3((unsigned)a.m), dim3(1024), 0, stream, a);
    } else {
        if (fast) hipLaunchKernelGGL(ess_kernel<true>, dim3((unsigned)a.m), dim3(256), 0, stream, a);
        else      hipLaunchKernelGGL(ess_kernel<false>, dim3((unsigned)a.m), dim3(256), 0, stream, a);
    }
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs_tiles(hipStream_t stream, const double* L, int64_t n, int64_t ldl, double* Lt)
{
    const int64_t nrg = (n + RS_ROWS - 1) / RS_ROWS, nkb = rs_tile_quads(n);
    hipLaunchKernelGGL(rs_tile_kernel, dim3(8, (unsigned)nrg), dim3(256), 0, stream, L, n, ldl, nkb, Lt);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs_unpack(hipStream_t stream, const uint32_t* raw, int64_t count, double* out)
{
    if (count <= 0) return 0;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rs_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, raw, count, out);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs3_begin(hipStream_t stream, const Rs3Args& a, uint64_t span)
{
    hipLaunchKernelGGL(rs3_begin_kernel, dim3(2048), dim3(256), 0, stream, a, span);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs3_products(hipStream_t stream, const Rs3Args& a)
{
    if (a.nunits <= 0) return 0;
    hipLaunchKernelGGL(rs3_products_kernel, dim3((unsigned)a.nunits), dim3(256), 0, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

// the work-groups of a products launch: every (row group, part) that holds a non-zero of L, full parts first
void rs3_unit_table(int64_t n, std::vector<uint32_t>& units, int* nfull)
{
    std::vector<uint32_t> ragged;
    units.clear();
    const int64_t nbx = (n + RS_ROWS - 1) / RS_ROWS;
    for (int64_t bx = 0; bx < nbx; ++bx) {
        const int64_t kall = ((bx + 1) * RS_ROWS < n) ? (bx + 1) * RS_ROWS : n;
        for (int64_t by = 0; by * RS_KC < kall; ++by) {
            const uint32_t u = (uint32_t)bx | ((uint32_t)by << 16);
            if ((by + 1) * RS_KC <= kall) units.push_back(u); else ragged.push_back(u);
        }
    }
    *nfull = (int)units.size();
    units.insert(units.end(), ragged.begin(), ragged.end());
}

// rows per thread of the slice kernel (32 R rows per work-group): the fewest that keep the grid at <= 256 work-groups
int rs3_slice_rows(int64_t n) { const int64_t r = (n + 8191) / 8192; return r <= 1 ? 1 : r <= 2 ? 2 : r <= 4 ? 4 : 8; }
int rs3_slice_wgs(int64_t n) { const int64_t rw = 32 * (int64_t)rs3_slice_rows(n); return (int)((n + rw - 1) / rw); }

int launch_rs3_slice(hipStream_t stream, const Rs3Args& a)
{
    const int wgs = rs3_slice_wgs(a.n);
    if (wgs > RS3_MAX_WGS) { set_error("R-stream replay: n = %lld is beyond the slice kernel's %lld rows", (long long)a.n, (long long)RS3_MAX_N); return GPIRT_E_ARG; }
    switch (rs3_slice_rows(a.n)) {
    case 1: hipLaunchKernelGGL((rs3_slice_kernel<1>), dim3((unsigned)wgs), dim3(320), 0, stream, a); break;
    case 2: hipLaunchKernelGGL((rs3_slice_kernel<2>), dim3((unsigned)wgs), dim3(320), 0, stream, a); break;
    case 4: hipLaunchKernelGGL((rs3_slice_kernel<4>), dim3((unsigned)wgs), dim3(320), 0, stream, a); break;
    default: hipLaunchKernelGGL((rs3_slice_kernel<8>), dim3((unsigned)wgs), dim3(320), 0, stream, a); break;
    }
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_ll_term_probe(hipStream_t stream, const double* a, int64_t n, double* out, int fast)
{
    if (n <= 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ll_term_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, n, out, fast);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_advance_pos(hipStream_t stream, uint64_t* pos, uint64_t delta)
{
    hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, stream, pos, delta);
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
