// rs_lr.hip -- the R-stream predictor's pass WITHOUT a pass over L (rs_predict.hip, phase A).
//
// The predictor needs nu = L z for 32 candidate vectors per pass, to single precision: 134 MB of L as floats at n = 8192, 30 of
// a pass's 44 us.  But S = K(theta, theta) + jitter I (src/gpirtMCMC.cpp:15-17) is a smooth kernel matrix plus a multiple of
// the identity, and the blocks of its Cholesky factor BELOW the diagonal blocks have the kernel's numerical rank: with the
// r = 64 Chebyshev nodes c on [-5, 5] and the Lagrange basis V[i][k] = l_k(theta_i),
//       K(theta_i, theta_j) = sum_k V[i][k] K(c_k, theta_j)                 (to 1.2e-15 per entry: unit squared-exponential),
// so for rows i below a block J of columns      L[i, J] = V[i, :] C_J,      C_J = D_J V_J^T L_JJ^-T   (r x |J|),
// where D_J is the posterior covariance of the kernel AT THE NODES given the rows in front of block J -- the Schur complement
// of the factorisation carried in the basis.  D_J has a closed form in the PREFIX GRAM G_J = sum_{i < J} V[i,:]^T V[i,:]:
//       D_J = eps (eps I + M G_J)^-1 M,    M = K(c, c),  eps = the jitter,
// so every block is independent of the others: no recurrence, no dependence on the dense factor, 64-column blocks side by
// side (one work-group each: Gauss-Jordan with partial pivoting on 64 x 64, then a forward substitution with the block's
// diagonal block of the dense factor, which equals the Cholesky factor of V_J D_J V_J^T + eps I to 5e-13 -- all in fp64).  NumPy at n = 4096: the factor assembled this way matches arma::chol's to
// 4e-12 (5e-10 with theta sorted); applied in single precision nu differs from the exact product by 5e-7 relative (the dense
// single-precision pass: 6e-8) -- a PREDICTION either way, verified exactly by phase B.
//
// A pass then is:  the predictor's products kernel on the DIAGONAL parts only (32 rows x the 512-column part that holds the
// row group's diagonal: 16 MB instead of 134) plus 2 x parts units that apply C (as two more "row groups" of tiles) to the
// part's normals: y_J = C_J z_J;  rs_lr_apply_kernel: nu[rows of part I] += V (sum_{J < I} y_J);  the decide kernel as before.
#include "common.h"
#include "kernels.h"
#include <utility>

namespace gpirt {

namespace {

constexpr int LR_R = RS_LR_RANK;         // nodes
constexpr int LR_B = 64;                 // columns per block of the construction
constexpr int LR_MS = LR_R + 1;          // row stride of the 64 x 64 matrices in LDS

// V[i][k] = l_k(theta_i): barycentric form, theta clamped to the nodes' interval (a theta outside it -- possible only for a
// theta_init the caller chose; draw_theta's values lie on the grid -5 .. 5 -- gets a wrong row, which costs mispredictions)
__global__ __launch_bounds__(256) void lr_basis_kernel(const double* __restrict__ theta, int64_t n, int64_t npad,
                                                       const double* __restrict__ nodes, const double* __restrict__ wts,
                                                       double* __restrict__ V64, float* __restrict__ V32t)
{
    __shared__ double c[LR_R], w[LR_R];
    if (threadIdx.x < LR_R) { c[threadIdx.x] = nodes[threadIdx.x]; w[threadIdx.x] = wts[threadIdx.x]; }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    double* out = V64 + i * LR_R;
    if (i >= n) {
        for (int k = 0; k < LR_R; ++k) out[k] = 0.0;
        return;
    }
    double t = theta[i];
    t = t < -5.0 ? -5.0 : (t > 5.0 ? 5.0 : t);
    if (!(t == t)) t = 0.0;
    int hit = -1;
    double s = 0.0;
    for (int k = 0; k < LR_R; ++k) {
        const double d = t - c[k];
        if (d == 0.0) hit = k; else s += w[k] / d;
    }
    for (int k = 0; k < LR_R; ++k) {
        const double v = (hit >= 0) ? (k == hit ? 1.0 : 0.0) : (w[k] / (t - c[k])) / s;
        out[k] = v;
        V32t[(int64_t)k * n + i] = (float)v;
    }
}

// Gb[b] = V_b^T V_b over the block's 64 rows
__global__ __launch_bounds__(256) void lr_gram_kernel(const double* __restrict__ V64, double* __restrict__ Gb)
{
    __shared__ double Vb[LR_B * LR_MS];
    const int tid = threadIdx.x;
    const double* src = V64 + (int64_t)blockIdx.x * LR_B * LR_R;
    for (int e = tid; e < LR_B * LR_R; e += 256) Vb[(e / LR_R) * LR_MS + (e % LR_R)] = src[e];
    __syncthreads();
    const int k = tid >> 2, l0 = (tid & 3) * 16;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int i = 0; i < LR_B; ++i) {
        const double a = Vb[i * LR_MS + k];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += a * Vb[i * LR_MS + l0 + q];
    }
    double* out = Gb + (int64_t)blockIdx.x * LR_R * LR_R + k * LR_R + l0;
#pragma unroll
    for (int q = 0; q < 16; ++q) out[q] = acc[q];
}

// exclusive prefix over the blocks, in place: Gb[b] <- sum_{b' < b} Gb[b']
__global__ __launch_bounds__(256) void lr_scan_kernel(double* __restrict__ Gb, int nb)
{
    const int e = blockIdx.x * 256 + threadIdx.x;          // one of the LR_R x LR_R entries
    double run = 0.0;
    int b = 0;
    for (; b + 8 <= nb; b += 8) {
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = Gb[(int64_t)(b + q) * LR_R * LR_R + e];
#pragma unroll
        for (int q = 0; q < 8; ++q) { Gb[(int64_t)(b + q) * LR_R * LR_R + e] = run; run += t[q]; }
    }
    for (; b < nb; ++b) { const double t = Gb[(int64_t)b * LR_R * LR_R + e]; Gb[(int64_t)b * LR_R * LR_R + e] = run; run += t; }
}

// One step of the Gauss-Jordan elimination of lr_coef_kernel, the column index a template parameter: every index into the
// thread's 32 entries is static (a loop the optimiser declines to unroll would index them dynamically: scratch)
template <int K>
__device__ __forceinline__ void gj_step(double (&a)[32], unsigned long long& used, int& myk, double (*colbuf)[LR_R], double* prow,
                                        const int ri, const int cs, const int lane, const int tid, int* bad)
{
    // the pivot row of column K: every wave for itself
    double v = ((used >> lane) & 1ull) ? -1.0 : fabs(colbuf[K & 1][lane]);
    int idx = lane;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double v2 = __shfl_xor(v, o);
        const int i2 = __shfl_xor(idx, o);
        if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
    }
    const int p = idx;
    if (tid == 0 && !(v > 0.0)) atomicOr(bad, 1);
    used |= 1ull << p;
    if (ri == p) {
        myk = K;
#pragma unroll
        for (int q = 0; q < 32; ++q) prow[32 * cs + q] = a[q];
    }
    __syncthreads();
    {
        const double pv = 1.0 / colbuf[K & 1][p];
        const double f = colbuf[K & 1][ri] * pv;
        if (ri == p) {
#pragma unroll
            for (int q = 0; q < 32; ++q) a[q] *= pv;
        } else {
#pragma unroll
            for (int q = 0; q < 32; q += 2) {
                const double2 pr = *reinterpret_cast<const double2*>(prow + 32 * cs + q);
                a[q] -= f * pr.x; a[q + 1] -= f * pr.y;
            }
        }
    }
    if (K + 1 < LR_R && cs == ((K + 1) >> 5)) colbuf[(K + 1) & 1][ri] = a[(K + 1) & 31];
    __syncthreads();
}
template <int... Ks>
__device__ __forceinline__ void gj_all(std::integer_sequence<int, Ks...>, double (&a)[32], unsigned long long& used, int& myk,
                                       double (*colbuf)[LR_R], double* prow, const int ri, const int cs, const int lane, const int tid, int* bad)
{
    (gj_step<Ks>(a, used, myk, colbuf, prow, ri, cs, lane, tid, bad), ...);
}

// One column of the substitution C L^T = T of lr_coef_kernel (four lanes to a row of T, lane cs holds the columns 4 i + cs)
template <int J>
__device__ __forceinline__ void subst_step(double (&t)[16], const double* Lb, const int cs, const int lane)
{
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (4 * i < J) sum += ((4 * i + cs) < J) ? t[i] * Lb[J * LR_MS + 4 * i + cs] : 0.0;
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    const double cj = (__shfl(t[J >> 2], (lane & ~3) | (J & 3)) - sum) / Lb[J * LR_MS + J];
    if (cs == (J & 3)) t[J >> 2] = cj;
}
template <int... Js>
__device__ __forceinline__ void subst_all(std::integer_sequence<int, Js...>, double (&t)[16], const double* Lb, const int cs, const int lane)
{
    (subst_step<Js>(t, Lb, cs, lane), ...);
}

// One 64-column block b:  D = eps (eps I + M G_b)^-1 M;  T = D V_b^T;  C = T L_bb^-T  (L_bb: the block's diagonal block of the
// dense factor -- V_b D V_b^T + eps I = L_bb L_bb^T to 5e-13)  -> float tiles in the layout of the predictor's products.
// 256 threads = 64 rows x 4 column segments.  The Gauss-Jordan elimination keeps every thread's 32 entries of the augmented
// row [eps I + M G | M] in REGISTERS (steps fully unrolled: static indices), pivots implicitly (the row with the largest
// entry of column k among the rows not used yet serves column k where it stands; every wave finds it by itself from the
// column in LDS) and costs two barriers a step.  The substitution C L^T = T runs four lanes to a row of T.
__global__ __launch_bounds__(256) void lr_coef_kernel(const double* __restrict__ V64, const double* __restrict__ Gp,
                                                      const double* __restrict__ Mn, const double* __restrict__ L, int64_t ldl,
                                                      double eps, int64_t n, int64_t nk8, float* __restrict__ Ct32, int* __restrict__ bad)
{
    __shared__ double R1[LR_R * LR_MS];                    // M, then D
    __shared__ double R2[LR_R * LR_MS];                    // G, then X, then L_bb
    __shared__ double Vb[LR_B * LR_MS];
    __shared__ double colbuf[2][LR_R];
    __shared__ __attribute__((aligned(16))) double prow[2 * LR_R];
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63;
    const int ri = tid >> 2, cs = tid & 3;
    const double* G = Gp + (int64_t)b * LR_R * LR_R;
    const double* vsrc = V64 + (int64_t)b * LR_B * LR_R;
    for (int e = tid; e < LR_R * LR_R; e += 256) {
        const int i = e / LR_R, j = e % LR_R;
        R1[i * LR_MS + j] = Mn[e]; R2[i * LR_MS + j] = G[e]; Vb[i * LR_MS + j] = vsrc[e];
    }
    __syncthreads();
    // this thread's 32 entries of row ri of [eps I + M G | M]: columns 32 cs .. 32 cs + 31
    double a[32];
    if (cs < 2) {
#pragma unroll
        for (int q = 0; q < 32; ++q) a[q] = (32 * cs + q == ri) ? eps : 0.0;
        for (int x = 0; x < LR_R; ++x) {
            const double mv = R1[ri * LR_MS + x];
#pragma unroll
            for (int q = 0; q < 32; ++q) a[q] += mv * R2[x * LR_MS + 32 * cs + q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 32; ++q) a[q] = R1[ri * LR_MS + 32 * (cs - 2) + q];
    }
    if (cs == 0) colbuf[0][ri] = a[0];
    __syncthreads();
    unsigned long long used = 0ull;
    int myk = 0;
    gj_all(std::make_integer_sequence<int, LR_R>{}, a, used, myk, colbuf, prow, ri, cs, lane, tid, bad);
    // X = (eps I + M G)^-1 M: the row that served column k holds X[k, :] in its right half
    if (cs >= 2) {
#pragma unroll
        for (int q = 0; q < 32; ++q) R2[myk * LR_MS + 32 * (cs - 2) + q] = a[q];
    }
    __syncthreads();
    for (int e = tid; e < LR_R * LR_R; e += 256) {
        const int i = e / LR_R, j = e % LR_R;
        R1[i * LR_MS + j] = 0.5 * eps * (R2[i * LR_MS + j] + R2[j * LR_MS + i]);          // D
    }
    __syncthreads();
    // L_bb into R2 (rows and columns past the matrix: the identity)
    for (int e = tid; e < LR_B * LR_B; e += 256) {
        const int i = e % LR_B, j = e / LR_B;                  // (column-major source: i runs fastest)
        const int64_t gi = (int64_t)b * LR_B + i, gj = (int64_t)b * LR_B + j;
        R2[i * LR_MS + j] = (gi < n && gj < n) ? ((i >= j) ? L[gi + gj * ldl] : 0.0) : (i == j ? 1.0 : 0.0);
    }
    // T[k = ri][j = 4 i + cs], i < 16:  T = D V_b^T
    double t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = 0.0;
    for (int x = 0; x < LR_R; ++x) {
        const double dv = R1[ri * LR_MS + x];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] += dv * Vb[(4 * i + cs) * LR_MS + x];
    }
    __syncthreads();
    // C L^T = T:  c_j = (t_j - sum_{x < j} c_x L[j][x]) / L[j][j], the four lanes of a row share the sum
    subst_all(std::make_integer_sequence<int, LR_B>{}, t, R2, cs, lane);
    // the tiles of the predictor's products (rs32_tile_kernel's layout): row group = k / 32, oct = column / 8,
    // lane = (k % 32) + 32 ((column % 8) / 4), element column % 4
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = 4 * i + cs;
        const int64_t col = (int64_t)b * LR_B + j;
        const double v = (col < n) ? t[i] : 0.0;
        if (!(fabs(v) < 1e6)) atomicOr(bad, 4);
        const int64_t ko = col >> 3;
        const int ln = (ri & 31) + 32 * (int)((col & 7) >> 2);
        Ct32[(((int64_t)(ri >> 5) * nk8 + ko) * 256) + 4 * ln + (int)(col & 3)] = (float)v;
    }
}

// nu[c][row] (part 0 of part32: the diagonal part's product) += V[row, :] (sum_{J < I} y_J[:, c]), I = the row's part.
// One work-group per 32 rows (the row groups of the products), thread = (row, four candidates); the prefix over the parts'
// records is recomputed by every work-group of a part (8 KB per part in front, out of the L2).
__global__ __launch_bounds__(256) void rs_lr_apply_kernel(Rs3Args a)
{
    __shared__ __attribute__((aligned(16))) float pref[LR_R * RS3_CAND];      // [k][c]
    const uint64_t item0 = a.anchor[0], stalled = a.anchor[3];
    if (item0 >= (uint64_t)a.m || stalled != 0) return;
    const int tid = threadIdx.x;
    const int64_t n = a.n;
    const int64_t row0 = (int64_t)blockIdx.x * RS_ROWS;
    const int I = (int)(row0 / RS3P_KC);
    if (I == 0) return;                                      // (uniform: the first part's rows have nothing in front)
    const int64_t row = row0 + (tid & 31) < n ? row0 + (tid & 31) : n - 1;
    const int c0 = (tid >> 5) * 4;
    // this thread's column of V and its four sums so far: issued before the prefix is built
    float v[LR_R], acc[4];
#pragma unroll
    for (int k = 0; k < LR_R; ++k) v[k] = a.V32t[(int64_t)k * n + row];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = a.part32[(int64_t)(c0 + q) * n + row];
    // pref[k][c]: entry e = c * 64 + k of the parts' records; eight entries per thread, the parts four at a time
    constexpr int NE = (LR_R * RS3_CAND) / 256;
    float s[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) s[q] = 0.0f;
    int J = 0;
    for (; J + 4 <= I; J += 4) {
        float t[4][NE];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < NE; ++q) t[u][q] = a.lrY[(int64_t)(J + u) * LR_R * RS3_CAND + tid + 256 * q];
#pragma unroll
        for (int q = 0; q < NE; ++q) s[q] += (t[0][q] + t[1][q]) + (t[2][q] + t[3][q]);
    }
    for (; J < I; ++J)
#pragma unroll
        for (int q = 0; q < NE; ++q) s[q] += a.lrY[(int64_t)J * LR_R * RS3_CAND + tid + 256 * q];
#pragma unroll
    for (int q = 0; q < NE; ++q) { const int e = tid + 256 * q; pref[(e % LR_R) * RS3_CAND + (e / LR_R)] = s[q]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LR_R; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(pref + k * RS3_CAND + c0);
        acc[0] = __builtin_fmaf(v[k], p.x, acc[0]); acc[1] = __builtin_fmaf(v[k], p.y, acc[1]);
        acc[2] = __builtin_fmaf(v[k], p.z, acc[2]); acc[3] = __builtin_fmaf(v[k], p.w, acc[3]);
    }
    if (row0 + (tid & 31) < n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a.part32[(int64_t)(c0 + q) * n + row] = acc[q];
    }
}

}  // namespace

// nodes, barycentric weights and M = K(c, c) of the Chebyshev nodes on [-5, 5] (host, once per sampler)
void rs_lr_nodes(std::vector<double>& nodes, std::vector<double>& wts, std::vector<double>& M)
{
    const int r = LR_R;
    nodes.resize(r); wts.resize(r); M.resize((size_t)r * r);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int k = 0; k < r; ++k) {
        const long double a = (2 * k + 1) * pi / (2 * r);
        nodes[(size_t)k] = (double)(5.0L * cosl(a));
        wts[(size_t)k] = (double)(((k & 1) ? -1.0L : 1.0L) * sinl(a));      // Chebyshev points of the first kind
    }
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < r; ++j) {
            const double d = nodes[(size_t)i] - nodes[(size_t)j];
            M[(size_t)i * r + j] = exp(-0.5 * d * d);
        }
}

// the structured form's work-groups: every row group with the part that holds its diagonal, then 2 row groups of C per part
void rs_lr_unit_table(int64_t n, std::vector<uint32_t>& units)
{
    units.clear();
    const int64_t nbx = (n + RS_ROWS - 1) / RS_ROWS, parts = (n + RS3P_KC - 1) / RS3P_KC;
    for (int64_t bx = 0; bx < nbx; ++bx) units.push_back((uint32_t)bx | ((uint32_t)((bx * RS_ROWS) / RS3P_KC) << 16));
    for (int64_t by = 0; by < parts; ++by)
        for (uint32_t g = 0; g < LR_R / RS_ROWS; ++g) units.push_back(0x80000000u | g | ((uint32_t)by << 16));
}

int launch_rs_lr_setup(hipStream_t stream, const RsLrSetup& q)
{
    const int64_t n = q.n;
    const int nb = (int)((n + LR_B - 1) / LR_B);
    const int64_t npad = (int64_t)nb * LR_B;
    hipLaunchKernelGGL(lr_basis_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, stream, q.theta, n, npad, q.nodes, q.wts, q.V64, q.V32t);
    hipLaunchKernelGGL(lr_gram_kernel, dim3((unsigned)nb), dim3(256), 0, stream, q.V64, q.Gb);
    hipLaunchKernelGGL(lr_scan_kernel, dim3(LR_R * LR_R / 256), dim3(256), 0, stream, q.Gb, nb);
    hipLaunchKernelGGL(lr_coef_kernel, dim3((unsigned)nb), dim3(256), 0, stream, q.V64, q.Gb, q.Mn, q.L, q.ldl, q.eps, n, q.nk8, q.Ct32, q.bad);
    GP_HIP(hipGetLastError());
    return 0;
}

int launch_rs_lr_apply(hipStream_t stream, const Rs3Args& a)
{
    hipLaunchKernelGGL(rs_lr_apply_kernel, dim3((unsigned)((a.n + RS_ROWS - 1) / RS_ROWS)), dim3(256), 0, stream, a);
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
