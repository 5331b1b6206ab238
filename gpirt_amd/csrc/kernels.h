// kernels.h -- internal launch functions of libgpirt_hip (device pointers, explicit stream).
#pragma once

#include "common.h"

namespace gpirt {

enum { TRI_NONE = 0, TRI_SYRK_LOWER = 1, TRI_A_LOWER = 2, TRI_A_UPPER = 3, TRI_SYRK_LOWER_TRAILING = 4,
       TRI_SYRK_LOWER_BACKGROUND = 5 /* deferred trailing update (GPIRT_DEFER): its own tile threshold GPIRT_BG128_MIN */ };
bool gemm_trailing_uses_128(int64_t M, int64_t N, bool background = false);
int launch_gemm_update_potf2(hipStream_t stream, int64_t M, int64_t N, int64_t K, const double* P, int64_t ldp,
                             double* C, int64_t ldc, int nb_next, int k0_next, int* info);   // does a trailing update of this shape run the 128-tile kernel?

// gemm_f64.hip
int launch_gemm(gpirt_handle_t h, hipStream_t stream, bool ta, bool tb, int tri, int64_t M,
                int64_t N, int64_t K, double alpha, const double* A, int64_t lda, const double* B,
                int64_t ldb, double beta, double* C, int64_t ldc, int64_t Mread = 0, const int* run_if = nullptr);
// run_if (plain products only): a device flag; the launch does nothing unless it is non-zero when the kernel starts.
// Mread (> M, !ta only): rows of A beyond M that exist in memory (padding up to a tile multiple) -- lets the
// last block row take the branch-free main loop; whatever those rows hold only reaches masked rows of C.

int launch_gemm_batched(hipStream_t stream, bool ta, bool tb, int tri, int64_t M, int64_t N, int64_t K, double alpha,
                        const double* A, int64_t lda, int64_t strideA, const double* B, int64_t ldb, int64_t strideB,
                        double beta, double* C, int64_t ldc, int64_t strideC, int batch);

// nparts consecutive trailing updates of one block column (panel q = columns [q * kpart, (q + 1) * kpart) of A / B) as one
// launch + an in-order application: bit-identical to nparts separate launches (work: nparts * M * N doubles)
int launch_syrk_panels(hipStream_t stream, int64_t M, int64_t N, int64_t kpart, int nparts, double alpha, const double* A,
                       int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, double* work);
int gemm_split_count(gpirt_handle_t h, hipStream_t stream, int tri, int64_t M, int64_t N, int64_t K);
// split-K product for small M x N with long K: parts land in Cpart (+ q * strideC, each M x N with ldc == M),
// then Cout (ldout) = beta_out * Cout + their sum
int launch_gemm_splitk(hipStream_t stream, bool ta, bool tb, int64_t M, int64_t N, int64_t K, double alpha,
                       const double* A, int64_t lda, const double* B, int64_t ldb, double* Cpart, int64_t ldc,
                       int64_t strideC, int nsplit, double* Cout, int64_t ldout, double beta_out);

// se_kernel.hip
int launch_se_kernel(hipStream_t stream, const double* x1, int64_t n1, const double* x2, int64_t n2,
                     double* out, int64_t ld, double jitter);
// lower-triangular blocks only (upper blocks are left untouched): the potrf input
int launch_se_kernel_lower(hipStream_t stream, const double* x, int64_t n, double* out, int64_t ld,
                           double jitter, bool fp32 = false);

// potrf.hip
int launch_potrf_lower(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda,
                       bool zero_upper, bool reset_info = true, int64_t extra_rows = 0);

int potrf_guard_reset(gpirt_handle_t h, hipStream_t stream);      // hang-guard fallback: see potrf.hip

// the same factorisation in pieces (distributed hosts): outer panel p = columns [p W, (p + 1) W)
int64_t potrf_panel_width();
int64_t potrf_subpanel_width(int64_t n);     // first sub-panel of an outer panel of an n x n factorisation (GPIRT_NBP, or by size)
// half: 0 = the panel's first sub-panel, 1 = the rest of it, 2 = the whole panel (persistent panel kernel for 0 / 1)
int potrf_panel_factor(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t p, int64_t extra_rows = 0,
                       int half = 2);
// part: 2 = the whole update of block column c by panel p; 0 = what needs only the panel's first sub-panel; 1 = the rest
int potrf_panel_update(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t p, int64_t c,
                       int64_t extra_rows = 0, int part = 2);
// half as for potrf_panel_factor: the columns of that part of the panel, rows from the part's first row down
// capacity: doubles the buffer holds (< 0: not checked) -- a part that does not fit is refused, never truncated
int potrf_panel_copy(hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t p, double* buf, bool to_buf,
                     int64_t extra_rows = 0, int half = 2, int64_t capacity = -1);

// panel.hip: columns [K0, c1) of the Cholesky factor, all rows below, one persistent kernel
int launch_panel_ll(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t K0, int64_t c1,
                    int64_t row_end = 0, unsigned long long* epoch_out = nullptr);
size_t panel_ll_smem_bytes();

// trsm.hip
int launch_trsm_lower(gpirt_handle_t h, hipStream_t stream, const double* L, int64_t n, int64_t ldl,
                      double* B, int64_t nrhs, int64_t ldb, bool trans, bool reuse_inverses = false);

// the block inverses launch_trsm_lower applies, exposed so that they can be built by ranges of 512-block pairs [p0, p1) as
// the factor's outer panels finish; *_mark tells the handle that its inverses now belong to (L, n, ldl)
int trsm_inverses_reserve(gpirt_handle_t h, hipStream_t stream, int64_t n, int64_t nrhs, bool thin);
int trsm_inverses_build(gpirt_handle_t h, hipStream_t stream, const double* L, int64_t n, int64_t ldl, bool thin,
                        int64_t p0, int64_t p1);
void trsm_inverses_mark(gpirt_handle_t h, const double* L, int64_t n, int64_t ldl, bool thin);

// rng.hip
int launch_item_uniforms(hipStream_t stream, uint64_t seed, uint32_t iter, uint32_t stage,
                         uint32_t item0, int64_t n_items, int64_t n_index, double* out, bool normal);
// z (n x m) from the R stream: column j, row i <- rnorm from U[pos0 + j*stride + 2i], U[.. + 1]
int launch_rstream_normals(hipStream_t stream, const double* U, const uint64_t* d_pos,
                           int64_t col_stride, int64_t n, int64_t m, double* out);

// ess.hip
struct EssArgs {
    double* f; const double* nu; const double* y; const double* mu;
    int64_t n, m;
    int* k_out;           // rejection counts per column (may be null)
    int* err;             // device error flag (set when the slice loop hits its cap)
    // item RNG
    uint64_t seed; uint32_t iter; uint32_t item0;
    // R-stream replay (U != null): one column per launch, uniforms from U[*pos + 2n ...]
    const double* U; uint64_t* pos; uint64_t cap;
    int ll_exact;         // 1: log(1 + exp(-a)) through the library's exp and log, as written (GPIRT_LL_EXACT)
    int screen;           // 1: the register kernels decide a trial point by the single-precision screen where it can (ll_fast.h)
};
int launch_ess(hipStream_t stream, const EssArgs& a);
// R-stream replay of draw_f, three items per pass over L (rng_ess.hip; sampler.hip, do_draw_f)
constexpr int RS_KC = 512;           // columns of L per part of a candidate product (one work-group: four waves x 128 columns)
constexpr int RS_ROWS = 32;          // rows of L per work-group of a candidate product
constexpr int RS_SPEC_MIN_N = 64;    // below: item by item, four launches each
constexpr int RS3_SLOTS = 3;         // items a pass can resolve
constexpr int RS3_C1 = 15;           // slot 1: the item before consumed 0 .. 14 uniforms behind its first two
constexpr int RS3_C2 = 16;           // slot 2: the two items before consumed 0 .. 15 together
constexpr int RS3_NT = 2;            // 16-wide MFMA tiles of a pass ...
constexpr int RS3_CAND = 16 * RS3_NT;   // ... = its columns: [slot 0 | slot 1 x 15 | slot 2 x 16]
static_assert(1 + RS3_C1 == 16 && RS3_C2 == 16 * (RS3_NT - 1), "tile 0 = slots 0 and 1, the other tiles = slot 2");
constexpr int RS3_TRIALS = 8;        // trial points of a slice loop evaluated per meeting of the work-groups
constexpr int RS3_MAX_WGS = 256;     // work-groups of the slice kernel (32 R rows each, R <= 8 rows per thread)
constexpr int64_t RS3_MAX_N = (int64_t)RS3_MAX_WGS * 256;
constexpr int RS3_QSTRIDE = 4 * RS3_NT * 65;         // products: one wave's accumulators of ONE tile parity per lane in LDS (rows of 65: bank spread)
constexpr int RS3_LDS_DOUBLES = (4 * RS3_QSTRIDE > 5 * RS_KC + 16 + RS3_C2) ? 4 * RS3_QSTRIDE : 5 * RS_KC + 16 + RS3_C2;    //   the three Nrm windows of a part (5 RS_KC + 16 + RS3_C2 doubles), then the four waves' accumulators
static_assert(5 * RS_KC + 16 + RS3_C2 <= RS3_LDS_DOUBLES, "the windows must fit");
struct Rs3Args {
    const double* U; uint64_t cap;   // the window of stream uniforms
    double* Nrm;                     // Nrm[r] = rnorm(U[r], U[r + 1]) for r in [cursor at the start of draw_f, *nrm_end)
    uint64_t* anchor;                // (8 words; [3] = the predictor has stalled, [4] = rounds of 16 trial points its anchor item has already lost, rs_predict.hip)
                                     // [0] the first item no pass has resolved yet (m: all done, or the draw has failed)  [1] where
                                     //   its normals start  [2] the end of Nrm -- one 32-byte record, read once per work-group
    uint64_t* pos;                   // the cursor (start of the next unconsumed uniform)
    uint64_t* posv;                  // [m + 1]: where item j's normals start
    int* k_out;                      // [m]: rejection counts
    int* err;                        // the draw's error flag (the failing kernel also closes the anchor)
    int64_t n, m;
    const double* Lt; int64_t nkb;   // L in 1 KiB tiles of 32 rows x 4 columns (launch_rs_tiles), nkb = rs_tile_quads(n) tiles per row group
    double* part;                    // [parts][RS3_CAND][n] parts of the pass's candidate products
    const uint32_t* units;           // the products' work-groups: (row group bx) | (part by) << 16, the nfull full parts first
    int nunits, nfull;
    int lim1, lim2;                  // slots 1 / 2 take counts below these (RS3_C1 / RS3_C2; smaller only through gpirt_debug_rs_cand_limit)
    double* f; const double* y; const double* mu;      // n x m
    // the slice kernel's work-groups meet through partial[2][RS3_MAX_WGS][RS3_TRIALS + 1] and flags[2][RS3_MAX_WGS]: a
    // work-group raises its flag to the meeting's tag = tag + (meetings before it in this launch); the host hands every
    // launch a range of 2^20 tags above all earlier ones, so nothing is ever reset
    double* partial; unsigned long long* flags; uint64_t tag;
    long long* trace;                // debug (gpirt_debug_rs_trace): in-kernel time stamps of this pass, or null
    // the predicted replay (rs_predict.hip): the predictor's passes work on single-precision tiles of L and leave
    // single-precision parts; anchor[3] != 0 = the predictor has stalled
    const float* Lt32; int64_t nk8;  // L in 1 KiB tiles of 32 rows x 8 columns of floats (launch_rs32_tiles), nk8 = rs32_tile_octs(n) per row group
    float* part32;                   // [parts][RS3_CAND][n]; non-null selects the predictor's form of the slice kernel
    int mispredict;                  // debug (gpirt_debug_rs_mispredict): the predictor is off by one at every mispredict-th item
    // rs3p_decide_kernel: partial sums [work-group][17], the candidates' walk records [32][18], the ticket (monotonic)
    double* dec_part; double* dec_rec; unsigned* dec_ticket;
    uint64_t* pass_count;            // += 1 per real pass of the predictor (rs_ctl[3])
    // the predictor's structured form (rs_lr.hip): the blocks of L below the diagonal parts as V C
    int lr;                          // != 0: the units are the diagonal parts + 2 row groups of C per part (bit 31 of the unit word);
                                     //       rs_lr_apply_kernel completes part 0 of part32, the decide kernel reads that one part
    const float* Ct32;               // C in the tile layout of Lt32: RS_LR_RANK rows
    float* lrY;                      // [parts][RS3_CAND][RS_LR_RANK]: y_J = C_J z_J of the pass
    const float* V32t;               // [RS_LR_RANK][n]: the Lagrange basis at theta
};
inline int64_t rs_tile_quads(int64_t n) { return (n + 3) / 4 + 1; }
inline size_t rs_tile_doubles(int64_t n) { return (size_t)((n + RS_ROWS - 1) / RS_ROWS) * (size_t)rs_tile_quads(n) * 128; }
int launch_rs_tiles(hipStream_t stream, const double* L, int64_t n, int64_t ldl, double* Lt);
int launch_rs_unpack(hipStream_t stream, const uint32_t* raw, int64_t count, double* out);   // MT words -> unif_rand() values
int launch_rs3_begin(hipStream_t stream, const Rs3Args& a, uint64_t span);   // Nrm over [cursor, cursor + span), anchor 0
int launch_rs3_products(hipStream_t stream, const Rs3Args& a);
void rs3_unit_table(int64_t n, std::vector<uint32_t>& units, int* nfull);
int rs3_slice_wgs(int64_t n);
int rs3_slice_rows(int64_t n);
int launch_rs3_slice(hipStream_t stream, const Rs3Args& a);
// rs_predict.hip: the predicted replay (phase A on single-precision tiles of L, phase B = one fp64 product + all slice loops
// side by side + an in-order commit)
constexpr int RS_LR_RANK = 64;       // Chebyshev nodes of the predictor's structured form (rs_lr.hip)
// (whole parts of 512 columns: the structured form's units read every oct of the part that holds a row group's diagonal)
inline int64_t rs32_tile_octs(int64_t n) { return ((n + 511) / 512) * 64 + 1; }
inline size_t rs32_tile_floats(int64_t n) { return (size_t)((n + RS_ROWS - 1) / RS_ROWS) * (size_t)rs32_tile_octs(n) * 256; }
#ifndef RS3P_KC_VALUE
#define RS3P_KC_VALUE 512
#endif
constexpr int RS3P_KC = RS3P_KC_VALUE;   // columns of L per unit of the PREDICTOR's products (its own unit table: rs3p_unit_table)
void rs3p_unit_table(int64_t n, std::vector<uint32_t>& units, int* nfull);
struct RsVerifyArgs {
    const double* f; double* nu; const double* y; const double* mu;    // n x m; nu = the product's columns for items j0 .. (n x (m - j0)), f' on return
    int64_t n, m, j0;
    const double* U; uint64_t cap;
    uint64_t* posv;                  // [m + 1] predicted starts; the commit corrects the entry behind a misprediction
    const uint64_t* anchorP;         // [0] = first item the predictor has NOT reached
    int* kv; int* used; int* ierr;   // [m]: rejections, uniforms consumed behind the normals, error code of each verified item
};
int launch_rs32_tiles(hipStream_t stream, const double* L, int64_t n, int64_t ldl, float* Lt, bool diag_only = false);
// rs_lr.hip: the structured form of the predictor's pass
struct RsLrSetup {
    const double* theta; int64_t n;
    const double* nodes; const double* wts; const double* Mn;     // RS_LR_RANK nodes, barycentric weights, K(c, c)
    double eps;                       // the jitter
    const double* L; int64_t ldl;     // the dense factor (its 64 x 64 diagonal blocks are read)
    double* V64; double* Gb;          // [64 ceil(n / 64)][RS_LR_RANK]; [ceil(n / 64)][RS_LR_RANK^2]
    float* V32t; float* Ct32; int64_t nk8;
    int* bad;                         // |= 1 zero pivot, 2 negative diagonal, 4 a coefficient beyond 1e6
};
void rs_lr_nodes(std::vector<double>& nodes, std::vector<double>& wts, std::vector<double>& M);
void rs_lr_unit_table(int64_t n, std::vector<uint32_t>& units);
int launch_rs_lr_setup(hipStream_t stream, const RsLrSetup& q);
int launch_rs_lr_apply(hipStream_t stream, const Rs3Args& a);
int launch_rs3p_products(hipStream_t stream, const Rs3Args& a);
int launch_rs3p_decide(hipStream_t stream, const Rs3Args& a);
int launch_rs_pred_start(hipStream_t stream, const uint64_t* anchor, uint64_t* anchorP, const int* k_last, int64_t m);
int launch_rs_gather(hipStream_t stream, const double* Nrm, const uint64_t* posv, const uint64_t* anchorP, int64_t n, int64_t j0,
                     int64_t m, double* Z);
int launch_rs_verify(hipStream_t stream, const RsVerifyArgs& a);
int launch_rs_commit(hipStream_t stream, const RsVerifyArgs& a, uint64_t* anchor, uint64_t* pos, uint64_t* ctl, int* err, double* f,
                     int* k_out);
int launch_ll_term_probe(hipStream_t stream, const double* a, int64_t n, double* out, int fast);     // 0 written, 1 ll_fast, 2 screen
int launch_ll_bar(hipStream_t stream, const double* f, const double* y, const double* mu, int64_t n,
                  int64_t m, double* out);

// fstar.hip
int launch_colnorm_s(hipStream_t stream, const double* tmp, int64_t n, int64_t N, int64_t ld, double* s);
int launch_lowrank_s(hipStream_t stream, const double* V, int64_t N, int r, const double* G, int64_t ldg, double* s);
struct FstarEpiArgs {
    const double* mean; const double* mu_star; const double* s; double* out;
    int64_t N, m;
    uint64_t seed; uint32_t iter; uint32_t item0;
    const double* U; uint64_t* pos; uint64_t cap; int* err;   // R-stream replay when U != null
    double* mean_out;     // optional copy of mean + mu_star
    int* off_scratch;     // N + 1 ints of device scratch for the R-stream consumption offsets
};
int launch_fstar_epilogue(hipStream_t stream, const FstarEpiArgs& a);

// theta.hip
int launch_indicators(hipStream_t stream, const double* y, int64_t n, int64_t m, double* Ypm /* n x 2m */);
int launch_loglik_terms(hipStream_t stream, const double* fstar, int64_t N, int64_t m, double* Gpm /* ldg x 2m */, int64_t ldg,
                        const int* run_if = nullptr);
// theta_fixed.hip: the same product in exact fixed point on the int8 matrix cores
struct TfDims { int64_t mp, ksteps, iblocks, gblocks; };
TfDims tf_dims(int64_t n, int64_t m, int64_t N);
size_t tf_y8_bytes(const TfDims& d);
size_t tf_gq_bytes(const TfDims& d);
size_t tf_aux_bytes(const TfDims& d);
int* tf_overflow(void* aux, const TfDims& d);
int launch_tf_indicators(hipStream_t stream, const double* y, int64_t n, int64_t ldy, int64_t m, const TfDims& d, void* Y8);
int launch_theta_fixed(hipStream_t stream, const double* fstar, int64_t N, int64_t n, int64_t m, const TfDims& d,
                       const void* Y8, void* Gq, void* aux, double* logpost, int64_t ldlp, bool trace = false,
                       gpirt_handle_t prof = nullptr);   // prof: the handle whose event-pair profiler brackets the int8 kernel (class 5)
long long* tf_trace(void* aux, const TfDims& d);            // (a traced launch: six stamps per work-group, tf_trace_wgs() of them)
int tf_trace_wgs();
struct ThetaArgs {
    const double* logpost;   // N x n (column i = respondent i0 + i), WITHOUT the prior
    int64_t N, n;
    int64_t i0;              // global index of the first respondent (a block of a sharded run; 0 otherwise):
                             // keys the RNG and offsets theta_out, so draws do not depend on the partition
    int stabilise;
    uint64_t seed; uint32_t iter;
    const double* U; uint64_t* pos; uint64_t cap;  // R-stream replay when U != null
    double* theta_out; int* degenerate; int* err;
};
int launch_theta_sample(hipStream_t stream, const ThetaArgs& a);

// beta.hip
struct BetaArgs {
    double* beta; const double* theta; const double* y; const double* f;
    const double* pm; const double* ps; const double* step;
    int64_t n, m, N;
    double* mu; double* mu_star;      // refreshed with the new beta (may be null)
    uint64_t seed; uint32_t iter; uint32_t item0;
    const double* U; uint64_t* pos; const uint64_t* item_off; uint64_t cap; int* err;  // R stream
};
int launch_draw_beta(hipStream_t stream, const BetaArgs& a);
int launch_linear_mean(hipStream_t stream, const double* x, int64_t n, const double* beta, int64_t m, double* mu);

// api.hip: an event pair around one launch while gpirt_prof_enable is on (classes: common.h); resolved by gpirt_prof_syrk
int prof_pair_begin(gpirt_handle_t h, hipStream_t stream, ProfPair& pp);
int prof_pair_end(gpirt_handle_t h, hipStream_t stream, ProfPair& pp, int cls, double flops, double bytes);

// api.hip: turns the hang-guard record of the panel kernel (info[1..7]) into the error message and clears it
int report_panel_guard(gpirt_handle_t h, const int* info_words, hipStream_t stream);

// api.hip
int create_side_handle(gpirt_handle_t* out, int device);

// misc
// out (cols x rows, ldo) = in^T, in is rows x cols with leading dimension ldi
int launch_transpose(hipStream_t stream, const double* in, int64_t rows, int64_t cols, int64_t ldi, double* out, int64_t ldo);
int launch_axpy_irf(hipStream_t stream, double* acc, const double* fstar, int64_t count);
int launch_advance_pos(hipStream_t stream, uint64_t* pos, uint64_t delta);

}  // namespace gpirt
