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
        out[g] = fast ? ll_term_fast(a[g]) : ll_term(a[g]);
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

// ---- R-stream replay, speculative form --------------------------------------------------------------------------------
// Item j's normals start where item j - 1's slice loop stopped consuming, so nu_j = L z_j cannot be STARTED before that
// loop ends -- but it has few possible values: z_j starts at  posv[j - 1] + 2n + 2 + k  for the rejection count k of item
// j - 1, and a pass over L (268 MB at n = 8192: HBM-bound) costs the same for 32 right-hand sides as for one.  So ONE grid
// per item runs item j's slice loop in its first work-groups and  L z  for the 32 candidates of item j + 1 in all the
// others (16 below n = 6144, kernels.h); the slice loop of item j + 1 then picks column k_j.  A count beyond the candidates raises
// `miss`: every later kernel of the
// pass leaves at once and the host redoes that item the plain way (do_draw_f).  Two launches per item on one stream (the
// candidate normals, then this grid) instead of four dependent ones; no events.
template <int RS_CAND>
__global__ __launch_bounds__(256) void rs_cand_normals_kernel(RsSpecArgs a)
{
    if (*(volatile int*)a.miss != 0) return;
    const uint64_t base = a.cand_first ? *a.pos : a.posv[a.cand_item - 1] + 2ull * (uint64_t)a.n + 2ull;
    if (a.cand_first && blockIdx.x == 0 && threadIdx.x == 0) a.posv[a.cand_item] = base;
    const int64_t total = a.n * RS_CAND;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = g / RS_CAND; const int c = (int)(g - i * RS_CAND);
        const uint64_t q = base + (uint64_t)c + 2ull * (uint64_t)i;
        a.cand_zc[g] = (q + 1 < a.cap) ? rnorm_from_two(a.U[q], a.U[q + 1]) : 0.0;     // (past the window: the slice loop reports it)
    }
}

// L in the order the candidate products read it: tile (row group rg of 32 rows, column quad kb) = 1 KiB, lane l's two rows of
// column 4 kb + (l >> 4) at doubles 2 l, 2 l + 1 -- the MFMA A operand of one step, so a wave's step is ONE contiguous
// kilobyte and its steps follow each other in memory.  From column-major L the same step touches four 256-byte pieces
// 64 KiB apart (a quarter of a DRAM page each): 3.55 TB/s.  Built once per iteration (a pass over the triangle: ~0.2 ms
// against 1024 passes that read it); rows and columns past the matrix are zeros; only the tiles a product reads exist.
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

// part[s][c][row] = sum over columns [s KC, (s + 1) KC) of L[row][k] zc[k][c] for RS_ROWS = 32 rows: fp64 MFMA 16x16x4 with
// A = 16 rows x 4 columns of L -- one 16-byte load per lane brings two rows (tiles t = 0, 1) -- and B = 4 x 16 candidates.
// The four waves of a work-group take a quarter of the part's columns each (a wave's loads are a dependent chain of ~2.5 us
// round trips: short chains and many waves are what fills the memory system -- 128 rows x 1024 columns per wave ran at
// 3 TB/s; 64 rows per wave with 32-byte loads, half as many waves, at 3.1), eight steps' loads in flight; the quarters meet
// in LDS and are added in order.  The strict upper triangle of L holds zeros (gpirt_sampler_create).
template <int RS_CAND>
__device__ __forceinline__ void rs_product_block(const RsSpecArgs& a, const int bx, const int by, double* red /* 16 x 64 doubles */)
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
    const double* Zp = a.cand_zc + i;
    constexpr int CT = RS_CAND / 16;                          // tiles of 16 candidates
    d4 acc[2][CT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[t][ct] = d4{ 0.0, 0.0, 0.0, 0.0 };
    int64_t k = k_beg;
    for (; k + 32 <= k_end; k += 32) {
        double2 av[8]; double b[8][CT];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t kk = k + 4 * u + g;
            av[u] = *reinterpret_cast<const double2*>(Lp + ((k >> 2) + u) * 128);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) b[u][ct] = Zp[kk * RS_CAND + 16 * ct];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                acc[0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].x, b[u][ct], acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].y, b[u][ct], acc[1][ct], 0, 0, 0);
            }
    }
    for (; k < k_end; k += 4) {
        const int64_t kk = k + g;
        const double2 av = *reinterpret_cast<const double2*>(Lp + (k >> 2) * 128);      // (past the matrix: zeros in the tile and in zc)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const double bb = Zp[kk * RS_CAND + 16 * ct];
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
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((t * CT + ct) * 4 + r) * 64] = acc[t][ct][r];
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][ct][r] += mine[((t * CT + ct) * 4 + r) * 64];
        }
    }
    if (kq != 0) return;
    // lane (i, g): acc[t][ct][r] = row r0 + 2 pi(g + 4 r) + t = r0 + 8 g + 2 r + t, candidate 16 ct + i
    double* out = a.cand_part + ((int64_t)by * RS_CAND) * n;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
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

// Sum of one value per thread over ALL the work-groups of the slice loop (pass = how many such sums came before): a block
// sum, then the work-groups' sums meet in memory and every work-group adds them in the same order -- so all of them see
// the same bits and take the same branches.  One lane publishes (agent-scope atomic store, then an add on the
// counter) and polls; the poll is bounded like every in-kernel wait of this library (flagsync.h): on expiry the pass is
// abandoned with GPIRT_E_HIP (as the panel kernel's guard is).  The work-groups are the FIRST of their grid on an otherwise idle stream: dispatched
// together, before any of the product work-groups.  Returns false when the wait expired (uniform).
__device__ __forceinline__ bool rs_sum_all(const RsSpecArgs& a, const int w, const int E, const int pass, const double v,
                                           double* red, double& out)
{
    const double bs = block_sum_256(v, red);
    if (E == 1) { out = bs; return true; }
    double* slot = a.ess_partial + (pass & 1) * RS_ESS_WGS;
    if (threadIdx.x == 0) {
        // flagsync.h's form: the value is stored write-through at agent scope, the store is waited for, then the counter;
        // the readers poll and read with agent-scope loads (served past the L1 and this XCD's L2) -- no release (it would
        // write back every dirty line the product work-groups of this XCD have produced) and no acquire per poll
        __hip_atomic_store(slot + w, bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(a.ess_cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long need = (unsigned long long)E * (unsigned long long)(pass + 1);
        int spins = 0;
        unsigned long long seen = __hip_atomic_load(a.ess_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (seen < need && ++spins < (1 << 22)) {
            __builtin_amdgcn_s_sleep(1);
            seen = __hip_atomic_load(a.ess_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        red[8] = (seen >= need) ? 1.0 : 0.0;
    }
    __syncthreads();
    const bool ok = red[8] != 0.0;
    double r = 0.0;
    for (int q = 0; q < E; ++q) r += __hip_atomic_load(slot + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();                                          // (red is reused by the next block sum)
    out = r;
    return ok;
}

// ess() of ONE item (src/draw-f.cpp:21-60) on the candidate its predecessor's count selects, the formula as written (no
// ll_fast: an R-stream replay keeps the reference's arithmetic).  On one compute unit the 2 + k likelihood passes over
// 8192 rows take ~85 us -- as long as the products beside them -- so the rows are spread over E work-groups (1024 each, four
// per thread, in registers) that meet once per pass (rs_sum_all).
template <int RS_CAND>
__device__ __forceinline__ void rs_ess_block(const RsSpecArgs& a, const int w, double* red)
{
    const int64_t n = a.n;
    const int j = a.ess_item, E = a.ess_wgs;
    const int kprev = a.ess_first ? 0 : a.k_out[j - 1];
    if (kprev >= a.cand_limit) {
        if (w == 0 && threadIdx.x == 0) *a.miss = j + 1;
        return;
    }
    double* fj = a.f; const double* yj = a.y; const double* mj = a.mu;
    const uint64_t p0 = a.posv[j] + 2ull * (uint64_t)n;                                // behind the n normals
    const int64_t per = ((n + E - 1) / E + 255) / 256 * 256;                           // rows per work-group
    const int64_t i0 = (int64_t)w * per, i1 = (i0 + per < n) ? i0 + per : n;
    const bool in_regs = per <= 4 * 256;
    // nu = the parts of candidate kprev, added in part order; kept in the first part's column (nobody else reads it)
    double* nj = a.ess_part + (int64_t)kprev * n;
    double F[4], V[4], M[4], Y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { F[e] = 0.0; V[e] = 0.0; M[e] = 0.0; Y[e] = __builtin_nan(""); }
    for (int64_t i = i0 + threadIdx.x, e = 0; i < i1; i += 256, e = (e + 1) & 3) {
        const int64_t grp = (i / RS_ROWS) * RS_ROWS;
        const int64_t kall = (grp + RS_ROWS < n) ? grp + RS_ROWS : n;
        const int parts = (int)((kall + RS_KC - 1) / RS_KC);
        double v = nj[i];
        for (int q = 1; q < parts; ++q) v += a.ess_part[((int64_t)q * RS_CAND + kprev) * n + i];
        if (in_regs) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) if (ee == e) { V[ee] = v; F[ee] = fj[i]; M[ee] = mj[i]; Y[ee] = yj[i]; }
        } else nj[i] = v;                                     // (own rows only: re-read by this work-group alone)
    }
    if (!in_regs) __syncthreads();
    uint32_t uidx = 0;
    int pass = 0;
    bool overflow = false, nan_state = false, expired = false;
    auto next_u = [&]() -> double {
        const uint64_t q = p0 + uidx;
        double u;
        if (q >= a.cap) { overflow = true; u = 0.5; } else u = a.U[q];
        ++uidx;
        return u;
    };
    // log_y = ll_bar(f, y, mu) + log(u)                                   draw-f.cpp:28-29
    double acc = 0.0, ll0, llp = 0.0;
    if (in_regs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (Y[e] == Y[e]) acc += ll_term(Y[e] * (F[e] + M[e]));
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const double yy = yj[i];
            if (yy != yy) continue;
            acc += ll_term(yy * (fj[i] + mj[i]));
        }
    }
    if (!rs_sum_all(a, w, E, pass++, acc, red, ll0)) expired = true;
    ll0 = -ll0;
    const double u = next_u();
    const double log_y = ll0 + log(u);
    double eps_min = 0.0, eps_max = GP_2PI;                                // :33-34
    double eps = eps_min + (eps_max - eps_min) * next_u();                 // :35
    eps_min = eps - GP_2PI;                                                // :36
    int k = 0;
    double c = 1.0, s = 0.0;
    while (!expired) {
        c = cos(eps);
        s = sin(eps);
        acc = 0.0;
        if (in_regs) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (Y[e] == Y[e]) acc += ll_term(Y[e] * ((F[e] * c + V[e] * s) + M[e]));     // :43
        } else {
            for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
                const double yy = yj[i];
                if (yy != yy) continue;
                acc += ll_term(yy * ((fj[i] * c + nj[i] * s) + mj[i]));
            }
        }
        if (!rs_sum_all(a, w, E, pass++, acc, red, llp)) { expired = true; break; }
        llp = -llp;
        if (llp > log_y) break;                                            // :45-47
        if (llp != llp) { nan_state = true; break; }                       // NaN state: never accepts
        if (eps < 0.0) eps_min = eps; else eps_max = eps;                  // :50-55
        if (eps_min == eps_max) eps = eps_min;                             // R::runif(a,a) = a
        else eps = eps_min + (eps_max - eps_min) * next_u();               // :56
        ++k;
        if (k >= ESS_MAX_TRIALS || overflow) { overflow = true; break; }
    }
    if (expired) {
        if (threadIdx.x == 0 && a.err) atomicCAS(a.err, 0, (int)GPIRT_E_HIP);          // (reported like the panel kernel's guard)
        return;
    }
    if (in_regs) {
        for (int64_t i = i0 + threadIdx.x, e = 0; i < i1; i += 256, e = (e + 1) & 3) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) if (ee == e) fj[i] = F[ee] * c + V[ee] * s;
        }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) fj[i] = fj[i] * c + nj[i] * s;
    }
    if (w == 0 && threadIdx.x == 0) {
        a.k_out[j] = k;
        if (nan_state && a.err) atomicCAS(a.err, 0, (int)GPIRT_E_NUMERIC);
        else if (overflow && a.err) atomicCAS(a.err, 0, (int)GPIRT_E_RNG);
        a.posv[j + 1] = p0 + uidx;
        *a.pos = p0 + uidx;
    }
}

// (forcing five work-groups per CU -- 96 registers, every wave of a pass at n = 8192 resident at once -- changed nothing at
// n = 8192 and cost the slice loop spills at small n)
template <int RS_CAND>
__global__ __launch_bounds__(256) void rs_item_kernel(RsSpecArgs a)
{
    __shared__ double red[16 * 64];
    if (*(volatile int*)a.miss != 0) return;
    const int E = a.ess_item >= 0 ? a.ess_wgs : 0;
    if ((int)blockIdx.x < E) { rs_ess_block<RS_CAND>(a, (int)blockIdx.x, red); return; }
    if (a.cand_item < 0) return;
    const int nbx = (int)((a.n + RS_ROWS - 1) / RS_ROWS);
    const int id = (int)blockIdx.x - E;
    rs_product_block<RS_CAND>(a, id % nbx, id / nbx, red);
}

// Register-resident variant for n <= NTH * EPT: each lane keeps its EPT entries of f, nu, mu and y in
// registers, so the (2 + k) likelihood passes of a column touch memory once; arithmetic is identical
// to ess_kernel (same per-element expression, same reduction tree).
template <int EPT, int NTH, bool FAST>
__global__ __launch_bounds__(NTH) void ess_kernel_reg(EssArgs a)
{
    __shared__ double red[8];
    auto block_sum = [&](double v) { return NTH == 256 ? block_sum_256(v, red) : block_sum_512(v, red); };
    const int64_t j = blockIdx.x;
    const int64_t n = a.n;
    double* fj = a.f + j * n;
    const double* nj = a.nu + j * n;
    const double* yj = a.y + j * n;
    const double* mj = a.mu + j * n;
    const uint32_t item = a.item0 + (uint32_t)j;
    double F[EPT], V[EPT], M[EPT], Y[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = threadIdx.x + NTH * e;
        const bool in = i < n;
        F[e] = in ? fj[i] : 0.0;
        V[e] = in ? nj[i] : 0.0;
        M[e] = in ? mj[i] : 0.0;
        Y[e] = in ? yj[i] : __builtin_nan("");      // NaN = skipped, like a missing response
    }
    uint32_t uidx = 0;
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (Y[e] == Y[e]) acc += ll_t<FAST>(Y[e] * (F[e] + M[e]));
    const double ll0 = -block_sum(acc);
    const double u = item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx++);
    const double log_y = ll0 + log(u);                                     // draw-f.cpp:28-29
    double eps_min = 0.0, eps_max = GP_2PI;
    double eps = eps_min + (eps_max - eps_min) * item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx++);
    eps_min = eps - GP_2PI;                                                // :36
    int k = 0;
    bool bad = false;
    double c, s;
    for (;;) {
        c = cos(eps);
        s = sin(eps);
        acc = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (Y[e] == Y[e]) acc += ll_t<FAST>(Y[e] * ((F[e] * c + V[e] * s) + M[e]));   // :43
        const double llp = -block_sum(acc);
        if (llp > log_y) break;                                            // :45-47
        if (llp != llp) { bad = true; break; }
        if (eps < 0.0) eps_min = eps; else eps_max = eps;                  // :50-55
        if (eps_min == eps_max) eps = eps_min;
        else eps = eps_min + (eps_max - eps_min) * item_uniform(a.seed, a.iter, GPIRT_ST_F_ESS, item, uidx++);
        ++k;
        if (k >= ESS_MAX_TRIALS) { bad = true; break; }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int64_t i = threadIdx.x + NTH * e;
        if (i < n) fj[i] = F[e] * c + V[e] * s;
    }
    if (threadIdx.x == 0) {
        if (a.k_out) a.k_out[j] = k;
        if (bad && a.err) atomicCAS(a.err, 0, (int)GPIRT_E_NUMERIC);
    }
}

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

int launch_rs_cand_normals(hipStream_t stream, const RsSpecArgs& a)
{
    int64_t blocks = (a.n * a.cand + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (a.cand == 16) hipLaunchKernelGGL(rs_cand_normals_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else              hipLaunchKernelGGL(rs_cand_normals_kernel<RS_CAND_MAX>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs_item(hipStream_t stream, const RsSpecArgs& a)
{
    const unsigned nbx = (unsigned)((a.n + RS_ROWS - 1) / RS_ROWS), parts = (unsigned)((a.n + RS_KC - 1) / RS_KC);
    const unsigned grid = (a.ess_item >= 0 ? (unsigned)a.ess_wgs : 0u) + (a.cand_item >= 0 ? nbx * parts : 0u);
    if (grid == 0) return 0;
    if (a.cand == 16) hipLaunchKernelGGL(rs_item_kernel<16>, dim3(grid), dim3(256), 0, stream, a);
    else              hipLaunchKernelGGL(rs_item_kernel<RS_CAND_MAX>, dim3(grid), dim3(256), 0, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_ll_term_probe(hipStream_t stream, const double* a, int64_t n, double* out, bool fast)
{
    if (n <= 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ll_term_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, n, out, fast ? 1 : 0);
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
