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

// ---- R-stream replay: three items per pass over L ---------------------------------------------------------------------
// Item j's normals start where item j - 1's slice loop stopped consuming (src/draw-f.cpp:26,56), so nu_j = L z_j cannot be
// STARTED before that loop ends -- but it has few possible values, and a pass over L (268 MB at n = 8192: HBM-bound) costs
// nearly the same for 48 right-hand sides as for one.  R's inversion normal takes two consecutive uniforms
// (src/mvnormal.h:8), so with  Nrm[r] = rnorm(U[r], U[r + 1])  computed ONCE per iteration for every position r of the
// window (rs3_begin_kernel), the normals of an item that starts at position p are simply Nrm[p + 2 i]: every candidate is a
// strided window of one array, nothing per candidate is ever materialised.
// A pass is anchored at an item a whose start posv[a] is known exactly and serves THREE items (kernels.h, RS3_*):
//   slot 0  item a          1 candidate            start  posv[a]
//   slot 1  item a + 1     15 candidates  c = used(a) in 0..14            start  posv[a] + (2n + 2) + c
//   slot 2  item a + 2     32 candidates  c = used(a) + used(a+1) in 0..31   start  posv[a] + 2 (2n + 2) + c
// (used = uniforms the slice loop consumed behind its first two = its rejection count, src/draw-f.cpp:56): 48 columns = three
// 16-wide MFMA tiles per step of L.  rs3_products_kernel computes the 48 products, rs3_slice_kernel then runs the three
// slice loops one after the other, each on the column its predecessors' counts select, and leaves the next anchor.  A count
// beyond a slot's candidates just ends the pass early -- the next pass is anchored at the first unresolved item -- so there is
// no host round trip and no other path: the host enqueues ceil(m / 3) passes + a few spare ones (a pass that finds every
// item done leaves at once) and looks at the item counter once at the end.
//
// The slice loop itself is evaluated RS3_TRIALS points at a time: a rejected point only moves the bracket end of its own
// sign (src/draw-f.cpp:50-55), so the sequence of trial points is a function of the stream alone, not of the data -- the
// likelihoods of the next eight points are one pass over the rows and ONE meeting of the work-groups instead of eight.

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
    if (blockIdx.x == 0 && threadIdx.x == 0) { *a.next_item = 0; a.posv[0] = p; *a.nrm_end = end; }
    for (uint64_t r = p + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < end; r += (uint64_t)gridDim.x * blockDim.x)
        a.Nrm[r] = rnorm_from_two(a.U[r], a.U[r + 1]);
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
    for (; k + 32 <= k_end; k += 32) {
        double2 av[8]; double b[8][NT];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = *reinterpret_cast<const double2*>(Lp + ((k >> 2) + u) * 128);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) b[u][ct] = Zp[ct][2 * (k + 4 * u)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
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

__global__ __launch_bounds__(256) void rs3_products_kernel(Rs3Args a)
{
    __shared__ double red[8 * 3 * 64];
    if (*(volatile int*)a.err != 0) return;
    const int item0 = *(volatile int*)a.next_item;
    if (item0 >= (int)a.m) return;                            // every item is done: a spare pass
    const uint64_t base = a.posv[item0];
    const int nbx = (int)((a.n + RS_ROWS - 1) / RS_ROWS);
    const int bx = (int)blockIdx.x % nbx, by = (int)blockIdx.x / nbx;
    if ((int)a.m - item0 >= 3 && a.lim2 > 0) rs3_product_block<3>(a, base, bx, by, red);
    else                                     rs3_product_block<1>(a, base, bx, by, red);     // the last items: slots 0 and 1 alone
}

// Sums of V values per thread over ALL the work-groups of the slice kernel (sync = how many such meetings came before in
// this launch): wave sums, block sums, then the work-groups' sums meet in memory and every work-group adds them in the same
// order -- so all of them see the same bits and take the same branches.  flagsync.h's form: the values are stored
// write-through at agent scope by lanes of ONE wave, that wave waits for its stores, one lane adds to the counter and polls
// it; readers use agent-scope loads (no release: nothing else of this kernel is shared, and no acquire per poll).  The poll is
// bounded like every in-kernel wait of this library; the work-groups are the whole grid of a launch on an otherwise idle
// stream, <= RS3_MAX_WGS of them, so they are resident together.  Returns false when the wait expired (uniform per block).
template <int V>
__device__ __forceinline__ bool rs3_sum_all(const Rs3Args& a, const int w, const int E, const int sync, const double* acc,
                                            double* sh /* 5 V + 1 doubles */, double* tot)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();                                          // (sh is still being read from the meeting before)
#pragma unroll
    for (int v = 0; v < V; ++v) {
        double x = acc[v];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) sh[wv * V + v] = x;
    }
    __syncthreads();
    double bs = 0.0;
    if (threadIdx.x < V) bs = (sh[threadIdx.x] + sh[V + threadIdx.x]) + (sh[2 * V + threadIdx.x] + sh[3 * V + threadIdx.x]);
    if (E == 1) {
        if (threadIdx.x < V) sh[4 * V + threadIdx.x] = bs;
        if (threadIdx.x == 0) sh[5 * V] = 1.0;
        __syncthreads();
    } else {
        double* slot = a.partial + (size_t)(sync & 1) * RS3_MAX_WGS * V;
        if (wv == 0) {
            if (lane < V) __hip_atomic_store(slot + w * V + lane, bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __hip_atomic_fetch_add(a.cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long need = (unsigned long long)E * (unsigned long long)(sync + 1);
                int spins = 0;
                unsigned long long seen = __hip_atomic_load(a.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (seen < need && ++spins < (1 << 22)) {
                    __builtin_amdgcn_s_sleep(1);
                    seen = __hip_atomic_load(a.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sh[5 * V] = (seen >= need) ? 1.0 : 0.0;
            }
        }
        __syncthreads();
        if (threadIdx.x < V && sh[5 * V] != 0.0) {
            double r = 0.0;
            for (int q = 0; q < E; ++q) r += __hip_atomic_load(slot + q * V + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh[4 * V + threadIdx.x] = r;
        }
        __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < V; ++v) tot[v] = sh[4 * V + v];
    return sh[5 * V] != 0.0;
}

// ess() (src/draw-f.cpp:21-60) for the items of a pass, one after the other, the formula as written (no ll_fast: an R-stream
// replay keeps the reference's arithmetic).  The rows are spread over the E work-groups of the launch, R per thread, in
// registers; every work-group carries the same bracket state and takes the same branches.
template <int R>
__global__ __launch_bounds__(256) void rs3_slice_kernel(Rs3Args a)
{
    constexpr int T = RS3_TRIALS, V = T + 1;
    __shared__ double sh[5 * V + 1];
    __shared__ double cs[2 * T];
    const int tid = threadIdx.x, w = blockIdx.x, E = a.wgs;
    const int64_t n = a.n;
    if (*(volatile int*)a.err != 0) return;
    const int item0 = *(volatile int*)a.next_item;
    if (item0 >= (int)a.m) return;
    const int ns = ((int)a.m - item0 < RS3_SLOTS) ? (int)a.m - item0 : RS3_SLOTS;
    const uint64_t nrm_end = *a.nrm_end;
    uint64_t start = a.posv[item0];
    const int64_t i0 = (int64_t)w * (R * 256) + tid;
    int usum = 0, resolved = 0, sync = 0, fail = 0;
    for (int g = 0; g < ns; ++g) {
        int col = 0;
        if (g == 1) { if (usum >= a.lim1) break; col = 1 + usum; }
        if (g == 2) { if (usum >= a.lim2) break; col = 16 + usum; }
        const int64_t j = item0 + g;
        // the item's 2n uniforms (all its normals were built) and its first two slice uniforms lie inside the window
        if (start + 2ull * (uint64_t)n + 2ull > a.cap || start + 2ull * (uint64_t)(n - 1) >= nrm_end) { fail = GPIRT_E_RNG; break; }
        double* fj = a.f + j * n; const double* yj = a.y + j * n; const double* mj = a.mu + j * n;
        double F[R], Vn[R], M[R], Y[R];
#pragma unroll
        for (int e = 0; e < R; ++e) {
            const int64_t i = i0 + 256 * e;
            F[e] = 0.0; Vn[e] = 0.0; M[e] = 0.0; Y[e] = __builtin_nan("");      // NaN = skipped, like a missing response
            if (i < n) {
                // nu = the parts of column `col`, added in part order
                const int64_t grp = (i / RS_ROWS) * RS_ROWS;
                const int64_t kall = (grp + RS_ROWS < n) ? grp + RS_ROWS : n;
                const int parts = (int)((kall + RS_KC - 1) / RS_KC);
                double v = a.part[(int64_t)col * n + i];
                for (int q = 1; q < parts; ++q) v += a.part[((int64_t)q * RS3_CAND + col) * n + i];
                Vn[e] = v; F[e] = fj[i]; M[e] = mj[i]; Y[e] = yj[i];
            }
        }
        const uint64_t p0 = start + 2ull * (uint64_t)n;                       // behind the n normals
        uint32_t uidx = 0;
        bool bad_u = false;
        auto next_u = [&]() -> double {
            const uint64_t q = p0 + uidx;
            ++uidx;
            if (q >= a.cap) { bad_u = true; return 0.5; }
            return a.U[q];
        };
        const double u = next_u();                                             // draw-f.cpp:28
        double eps_min = 0.0, eps_max = GP_2PI;                                // :33-34
        double eps = eps_min + (eps_max - eps_min) * next_u();                 // :35
        eps_min = eps - GP_2PI;                                                // :36
        int k = 0;
        double log_y = 0.0, c = 1.0, s = 0.0;
        uint32_t uacc = 0;
        bool first = true, done = false;
        while (!done) {
            // the next T trial points: each is what :50-56 makes of the one before, were it rejected
            uint32_t Ut[T]; bool Bt[T];
            double my_eps = 0.0;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (tid == t) my_eps = eps;
                Ut[t] = uidx; Bt[t] = bad_u;
                if (eps < 0.0) eps_min = eps; else eps_max = eps;              // :50-55
                if (eps_min == eps_max) eps = eps_min;                         // R::runif(a, a) = a, nothing consumed
                else eps = eps_min + (eps_max - eps_min) * next_u();           // :56
            }
            __syncthreads();                                                   // (cs is still being read from the round before)
            if (tid < T) { cs[tid] = cos(my_eps); cs[T + tid] = sin(my_eps); }
            __syncthreads();
            double acc[V], tot[V];
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = 0.0;
            if (first) {
#pragma unroll
                for (int e = 0; e < R; ++e) if (Y[e] == Y[e]) acc[0] += ll_term(Y[e] * (F[e] + M[e]));          // :29
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const double ct = cs[t], st = cs[T + t];
#pragma unroll
                for (int e = 0; e < R; ++e) if (Y[e] == Y[e]) acc[1 + t] += ll_term(Y[e] * ((F[e] * ct + Vn[e] * st) + M[e]));   // :43
            }
            if (!rs3_sum_all<V>(a, w, E, sync++, acc, sh, tot)) { fail = GPIRT_E_HIP; break; }
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
                for (int t = 0; t < T; ++t) if (t == hit) { c = cs[t]; s = cs[T + t]; uacc = Ut[t]; bad = Bt[t]; }
                if (bad) { fail = GPIRT_E_RNG; break; }
                k += hit; done = true;
            } else {
                k += T;
                if (bad_u) { fail = GPIRT_E_RNG; break; }
                if (k >= ESS_MAX_TRIALS) { fail = GPIRT_E_NUMERIC; break; }
            }
        }
        if (fail) break;
#pragma unroll
        for (int e = 0; e < R; ++e) {
            const int64_t i = i0 + 256 * e;
            if (i < n) fj[i] = F[e] * c + Vn[e] * s;
        }
        start = p0 + uacc;
        usum += (int)uacc - 2;
        ++resolved;
        if (w == 0 && tid == 0) { a.k_out[j] = k; a.posv[j + 1] = start; }
    }
    if (tid == 0) {
        if (fail) atomicCAS(a.err, 0, fail);                  // every later kernel of the pass leaves at once
        else if (w == 0) { *a.next_item = item0 + resolved; *a.pos = start; }
    }
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
    const unsigned nbx = (unsigned)((a.n + RS_ROWS - 1) / RS_ROWS), parts = (unsigned)((a.n + RS_KC - 1) / RS_KC);
    hipLaunchKernelGGL(rs3_products_kernel, dim3(nbx * parts), dim3(256), 0, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

int rs3_slice_wgs(int64_t n) { const int64_t e = (n + 255) / 256; return (int)(e < RS3_MAX_WGS ? e : RS3_MAX_WGS); }

int launch_rs3_slice(hipStream_t stream, const Rs3Args& a)
{
    const int rows = (int)((a.n + (int64_t)a.wgs * 256 - 1) / ((int64_t)a.wgs * 256));     // per thread
    switch (rows) {
    case 1: hipLaunchKernelGGL(rs3_slice_kernel<1>, dim3((unsigned)a.wgs), dim3(256), 0, stream, a); break;
    case 2: hipLaunchKernelGGL(rs3_slice_kernel<2>, dim3((unsigned)a.wgs), dim3(256), 0, stream, a); break;
    case 3: hipLaunchKernelGGL(rs3_slice_kernel<3>, dim3((unsigned)a.wgs), dim3(256), 0, stream, a); break;
    case 4: hipLaunchKernelGGL(rs3_slice_kernel<4>, dim3((unsigned)a.wgs), dim3(256), 0, stream, a); break;
    default: set_error("R-stream replay: n = %lld is beyond the slice kernel's %d rows", (long long)a.n, RS3_MAX_WGS * 1024); return GPIRT_E_ARG;
    }
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
