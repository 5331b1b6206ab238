// runtime.hip -- the dependency-driven schedule of the blocked Cholesky (GPIRT_RUNTIME=2; arma::chol(S, "lower"),
// src/gpirtMCMC.cpp:76-78, 95-97).
//
// The launch-ordered schedule of potrf.hip is ~70 dependent launches on two streams.  Its sub-panel kernels need WHOLE
// compute units (148 KB of LDS per work-group) and wait ~75 us per launch for a generation of resident update work-groups
// to leave some; its updates run in partial last rounds and share the chip with work-groups that hold a CU at 9 % MFMA use
// (every schedule of rounds 2 and 3 moved those waits around).  Here a factorisation is three things
// that run side by side for its whole length:
//
//   * ONE persistent update kernel (gemm_f64.hip, update_worker_kernel): three work-groups per CU on all compute units
//     but RT_RESERVED, working off every MFMA product of the factorisation as 64 x 64 x K tile tasks from two queues --
//     what the pivot chain waits for next, and everything else -- whose entries become ready through counters;
//   * the sub-panel kernels of panel.hip, enqueued back to back on one high-priority stream, each swept over a WINDOW of
//     rows only (its own outer panel's and the next two: <= 48 row blocks) plus 8 row blocks of identity below the matrix,
//     on the RT_RESERVED compute units nothing else can occupy: they are resident the moment their predecessor leaves and
//     start each row block when its input tiles are in (per-row-group counters);
//   * the rows BELOW the window are not swept at all: X = A W^T with W the inverse of the sub-panel's 512 x 512 diagonal
//     block -- which IS what the sweep leaves in the identity rows (E L^-T for E = I) -- as tile tasks of the update kernel,
//     at the update kernel's rate instead of 0.7 ms of whole compute units (the far rows are what made a sub-panel kernel
//     129 work-groups wide).
//
// How the compute units get reserved without CU masks (which this stack ignores, tools/micro/cumask_probe.hip): place-holder
// work-groups that each fill a CU's LDS are launched first, the update kernel's grid is exactly what fits on the others,
// and its last arriving work-group releases the holders -- from then on only work-groups that need a whole CU find room
// there.
//
// Arithmetic: every tile receives the same products in the same order as in potrf.hip's schedule (PROG words order the
// updates of a tile), the window rows are swept by the same kernel: those entries of L are bit-identical.  Rows below a
// sub-panel's window are X = A W^T instead of a substitution: equal to rounding (<= 1e-12; cond(L_ss) <~ 2e3), tested.
#include "common.h"
#include "kernels.h"

#include <algorithm>
#include <vector>

namespace gpirt {

namespace {

constexpr int RT_W = 1024, RT_H = 512;     // outer panel / sub-panel width this schedule is built for (the defaults)
constexpr int RT_GB = RT_W / 64;           // row blocks per row group
constexpr int RT_WINDOW_GROUPS = 2;        // row groups a sub-panel kernel sweeps: its own outer panel and the next (with three, the third
                                           // group -- whose input comes two dependent tile products late -- kept every first sub-panel kernel
                                           // alive 0.6-1.3 ms after its diagonal rows were done, and the next launch waits for the kernel to END)
constexpr int RT_RESERVED = 64;            // compute units kept free for the sub-panel kernels (48 window + 8 identity row blocks): a multiple
                                           // of 32 -- work-groups are dealt round-robin to 8 XCCs x 4 shader engines, and only with 2 holders on
                                           // EVERY engine do 18 workers and 2 panel work-groups per engine always find their place (census: tools/rt_trace.py)

struct SubPanel { int p; int64_t k0, k1; int64_t near_end; bool valid; };

}  // namespace

struct RtState {
    int64_t n = 0, nr = 0, nid = 0;        // columns, real rows, first identity row (= nr rounded up to a row group)
    int P = 0, NS = 0, NG = 0, nbr = 0;    // outer panels, sub-panel slots (2 P), row groups incl. the identity group, real row blocks
    std::vector<SubPanel> sub;
    std::vector<unsigned int> in_need;     // [NS * NG]
    RtTask* d_tasks = nullptr; int ntasks[2] = { 0, 0 };
    int* d_state = nullptr;
    unsigned long long* d_words = nullptr; // cnt [2 NS NG] | prog [ntiles + 1] | strip [2 * 4096] | head [2] | ctl [4]
    size_t words = 0, off_prog = 0, off_strip = 0, off_head = 0, off_ctl = 0;
    unsigned int* d_in_need = nullptr;
    double* d_stage = nullptr;
    unsigned long long epoch = 0;
    int nworkers = 0, reserved = 0;
    double flops = 0.0;
    int nseg = 0;                          // segments of the bulk queue (one per step that needs the tiles next)
    std::vector<int> seg_begin;            // [nseg + 1]
    hipEvent_t ev_done = nullptr;
};

namespace {

__global__ void identity_rows_kernel(double* A, int64_t lda, int64_t row0, int64_t n)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) A[row0 + (c % RT_H) + c * lda] = 1.0;
}

__global__ void fill_words_kernel(unsigned long long* p, unsigned long long v) { *p = v; }

struct Unit { int kb, klen, cls; int sidK; };          // cls: 0 a, 1 b1, 2 b2, 3 c, 4 D
struct Keyed { long long k[7]; RtTask t; int queue; int prev; double est, dur; };

int build(gpirt_handle_t h, RtState* rt, int64_t n, int64_t nr)
{
    rt->n = n; rt->nr = nr;
    rt->nbr = (int)((nr + 63) / 64);
    const int G = (rt->nbr + RT_GB - 1) / RT_GB;
    rt->nid = (int64_t)G * RT_W;
    rt->NG = G + 1;
    rt->P = (int)((n + RT_W - 1) / RT_W);
    rt->NS = 2 * rt->P;
    const int P = rt->P, NS = rt->NS, NG = rt->NG, nbr = rt->nbr;
    const int ncb = (int)(n / 64);
    rt->sub.assign((size_t)NS, SubPanel{});
    for (int p = 0; p < P; ++p) {
        const int64_t c1 = (int64_t)p * RT_W, c2 = std::min<int64_t>(c1 + RT_W, n), cA = std::min<int64_t>(c1 + RT_H, c2);
        const int64_t ne = std::min<int64_t>(nr, (int64_t)(p + RT_WINDOW_GROUPS) * RT_W);
        rt->sub[2 * p] = SubPanel{ p, c1, cA, ne, true };
        rt->sub[2 * p + 1] = SubPanel{ p, cA, c2, ne, cA < c2 };
    }
    auto grp = [&](int rb) { return rb / RT_GB; };
    auto cnt_done = [&](int sid, int g) { return sid * NG + g; };
    auto cnt_in = [&](int sid, int g) { return NS * NG + sid * NG + g; };
    auto done_need = [&](int sid, int g) -> unsigned {
        const SubPanel& s = rt->sub[sid];
        if (g == G) return (unsigned)((s.k1 - s.k0) / 64);
        const int lo = std::max<int>((int)(s.k0 / 64), g * RT_GB), hi = std::min<int>((g + 1) * RT_GB, nbr);
        return hi > lo ? (unsigned)(hi - lo) : 0u;
    };
    rt->in_need.assign((size_t)NS * NG, 0u);
    const int ntiles = nbr * (nbr + 1) / 2;
    std::vector<Keyed> q[2], all;
    double flops = 0.0;
    // ---- the products: per tile the units of potrf.hip's schedule, in its order
    for (int bj = 0; bj < ncb; ++bj) {
        const int r = bj / RT_GB, hb = (bj % RT_GB) / (RT_H / 64);
        const int sid_c = 2 * r + hb;                              // the sub-panel that consumes this tile's column
        for (int bi = bj; bi < nbr; ++bi) {
            Unit u[40];
            int nu = 0;
            for (int qq = 0; qq + 2 <= r; ++qq) u[nu++] = Unit{ qq * RT_GB, RT_GB, 4, 2 * qq + 1 };
            if (r >= 1) {
                if (hb == 0) {
                    u[nu++] = Unit{ (r - 1) * RT_GB, RT_H / 64, 1, 2 * (r - 1) };
                    u[nu++] = Unit{ (r - 1) * RT_GB + RT_H / 64, RT_H / 64, 2, 2 * (r - 1) + 1 };
                } else {
                    u[nu++] = Unit{ (r - 1) * RT_GB, RT_GB, 3, 2 * (r - 1) + 1 };
                }
            }
            if (hb == 1) u[nu++] = Unit{ r * RT_GB, RT_H / 64, 0, 2 * r };
            if (nu == 0) continue;
            const int gi = grp(bi);
            rt->in_need[(size_t)sid_c * NG + gi] += 1;
            int prev_unit = -1;
            for (int k = 0; k < nu; ++k) {
                Keyed e{};
                e.prev = prev_unit; e.dur = 8.0 + 6.5 * u[k].klen;
                prev_unit = (int)all.size();
                RtTask& t = e.t;
                t.bi = (uint16_t)bi; t.bj = (uint16_t)bj; t.kb = (uint16_t)u[k].kb; t.klen = (uint16_t)u[k].klen;
                t.type = 0;
                const int sk = u[k].sidK;
                t.dep0 = cnt_done(sk, gi); t.need0 = done_need(sk, gi);
                if (grp(bj) != gi) { t.dep1 = cnt_done(sk, grp(bj)); t.need1 = done_need(sk, grp(bj)); } else { t.dep1 = -1; t.need1 = 0; }
                t.tile = bi * (bi + 1) / 2 + bj; t.prog_need = (uint16_t)k;
                t.last = (k == nu - 1) ? 1 : 0;
                t.out_idx = cnt_in(sid_c, gi);
                flops += 2.0 * 64 * 64 * 64.0 * u[k].klen;
                const int pk = sk / 2;
                // On the pivot chain's path: the tiles of the consuming sub-panel kernel's window (row groups r .. r + 2) AND of
                // the first row group below it (r + 3) -- those feed the X = A W^T tasks whose rows the NEXT outer panel's
                // window needs.  (With only the window urgent, the `a` tiles of group r + 3 sat in the bulk queue behind a
                // whole step of trailing updates and every first sub-panel kernel waited 2-10 ms for them.)
                const bool urgent = u[k].cls != 4 && gi <= r + RT_WINDOW_GROUPS;
                if (urgent) {
                    // queue 0, in the order the pivot chain releases and needs them: per releasing sub-panel first the tiles
                    // whose rows its own kernel swept, then (below) the X = A W^T tiles of the first row group under its
                    // window, then the tiles that read those rows; what gates the next sub-panel kernel (a, b2) before the rest
                    const bool far_rows = gi > pk + RT_WINDOW_GROUPS - 1;
                    e.k[0] = sk; e.k[1] = far_rows ? 2 : 0; e.k[2] = (u[k].cls == 0 || u[k].cls == 2) ? 0 : 1; e.k[3] = gi; e.k[4] = bi; e.k[5] = bj;
                    e.queue = 0; all.push_back(e);
                } else {
                    // The bulk queue, by DEADLINE.  What panel pk contributes to the rows of group g ("stage (pk, g)": X = A W^T
                    // for its first sub-panel, the a / b1 tiles, X = A W^T for the second, b2 / c, then its trailing updates of
                    // the later block columns) can only start when stage (pk - 1, g) is through, and stage (g - 2, g) is the
                    // urgent one the next chain launch waits for -- a pipeline along every row group.  Ordered by step first
                    // (every far row of panel p before anything of panel p + 1) the second far group of a panel came ~1 ms after
                    // its sub-panels, and with it the FIRST far group of the next panel, which the chain does wait for.  So:
                    // by the step that needs the tile next, then by row group (nearest first), then by panel, then in the
                    // order of a stage.
                    const int need_step = (u[k].cls == 4) ? r - 1 : (u[k].cls == 0 ? pk : pk + 1);
                    const int in_stage = (u[k].cls == 4) ? 4 : ((sk & 1) ? 3 : 1);
                    e.k[0] = need_step; e.k[1] = pk; e.k[2] = gi; e.k[3] = in_stage; e.k[4] = bi; e.k[5] = bj; e.k[6] = u[k].kb;
                    e.queue = 1; all.push_back(e);
                }
            }
        }
    }
    // ---- the far rows of every sub-panel: X = A W^T
    for (int sid = 0; sid < NS; ++sid) {
        const SubPanel& s = rt->sub[sid];
        if (!s.valid) continue;
        const int wb = (int)((s.k1 - s.k0) / 64);
        for (int bi = (int)(s.near_end / 64); bi < nbr; ++bi) {
            const int gi = grp(bi);
            for (int jb = 0; jb < wb; ++jb) {
                Keyed e{};
                RtTask& t = e.t;
                t.bi = (uint16_t)bi; t.bj = (uint16_t)jb; t.kb = (uint16_t)(s.k0 / 64); t.klen = (uint16_t)wb;
                t.type = 1;
                t.dep0 = cnt_done(sid, G); t.need0 = done_need(sid, G);
                const unsigned inn = rt->in_need[(size_t)sid * NG + gi];
                t.dep1 = inn ? cnt_in(sid, gi) : -1; t.need1 = inn;
                t.tile = ntiles; t.prog_need = 0;
                t.last = (uint8_t)wb; t.out_idx = cnt_done(sid, gi);
                flops += 2.0 * 64 * 64 * 64.0 * (jb + 1);
                e.prev = -1; e.dur = 14.0 + 6.5 * (jb + 1);
                if (gi == s.p + RT_WINDOW_GROUPS) {
                    e.k[0] = sid; e.k[1] = 1; e.k[2] = 0; e.k[3] = bi; e.k[4] = jb;
                    e.queue = 0; all.push_back(e);
                } else {
                    e.k[0] = s.p; e.k[1] = s.p; e.k[2] = gi; e.k[3] = (sid & 1) ? 2 : 0; e.k[4] = bi; e.k[5] = jb; e.k[6] = 0;
                    e.queue = 1; all.push_back(e);
                }
            }
        }
    }
    // ---- order of the queues.  Urgent: the order the pivot chain releases and needs its tiles.  Bulk: by the step that
    // needs the tile next, and within a step by operand panel -- so that in every step's SEGMENT of the queue the entries
    // whose operands exist come first and the ones that wait for sub-panels still to be factored sit together at its end.
    // A worker looks at the front of each segment in turn (gemm_f64.hip): scanning one queue in deadline order from its head,
    // a dequeue walked over thousands of blocked entries (~85 us on average, half of every worker's time).
    // (Measured and not kept: both queues sorted by the earliest start a host-side run of the dataflow gives each task --
    // short scans, but every trailing update that CAN run early then DOES, in front of what the chain needs: 9.7 -> 10.4 ms.)
    for (auto& e : all) q[e.queue].push_back(e);
    for (int z = 0; z < 2; ++z)
        std::stable_sort(q[z].begin(), q[z].end(), [](const Keyed& a, const Keyed& b) {
            for (int i = 0; i < 7; ++i) if (a.k[i] != b.k[i]) return a.k[i] < b.k[i];
            return false;
        });
    rt->nseg = P + 1;
    rt->seg_begin.assign((size_t)rt->nseg + 1, (int)q[1].size());
    for (int i = (int)q[1].size() - 1; i >= 0; --i) {
        const int d = (int)std::min<long long>(std::max<long long>(q[1][(size_t)i].k[0], 0), P);
        rt->seg_begin[(size_t)d] = i;
    }
    for (int d = rt->nseg - 1; d >= 0; --d) rt->seg_begin[(size_t)d] = std::min(rt->seg_begin[(size_t)d], rt->seg_begin[(size_t)d + 1]);
    rt->flops = flops;
    // ---- device copies
    rt->ntasks[0] = (int)q[0].size(); rt->ntasks[1] = (int)q[1].size();
    const size_t nt = q[0].size() + q[1].size();
    std::vector<RtTask> flat;
    flat.reserve(nt);
    for (int z = 0; z < 2; ++z) for (auto& e : q[z]) flat.push_back(e.t);
    GP_HIP(hipMalloc(&rt->d_tasks, nt * sizeof(RtTask)));
    GP_HIP(hipMalloc(&rt->d_state, nt * sizeof(int)));
    rt->off_prog = (size_t)2 * NS * NG;
    rt->off_strip = rt->off_prog + (size_t)ntiles + 1;
    rt->off_head = rt->off_strip + 2 * 4096;
    rt->off_ctl = rt->off_head + 64;                       // heads: [0] urgent queue, [1 + d] segment d of the bulk queue
    rt->words = rt->off_ctl + 8;
    GP_HIP(hipMalloc(&rt->d_words, rt->words * sizeof(unsigned long long)));
    GP_HIP(hipMalloc(&rt->d_in_need, (size_t)NS * NG * sizeof(unsigned int)));
    GP_HIP(hipMalloc(&rt->d_stage, (size_t)2 * rt->nid * RT_H * sizeof(double)));
    GP_HIP(hipMemcpy(rt->d_tasks, flat.data(), nt * sizeof(RtTask), hipMemcpyHostToDevice));
    GP_HIP(hipMemcpy(rt->d_in_need, rt->in_need.data(), (size_t)NS * NG * sizeof(unsigned int), hipMemcpyHostToDevice));
    GP_HIP(hipMemset(rt->d_words, 0, rt->words * sizeof(unsigned long long)));
    GP_HIP(hipMemset(rt->d_words + rt->off_prog + ntiles, 0xff, sizeof(unsigned long long)));     // the always-ready progress word
    GP_HIP(hipDeviceSynchronize());
    int per_cu = 0;
    GP_TRY(update_workers_per_cu(&per_cu));
    if (per_cu < 1) { set_error("update worker kernel does not fit on a compute unit"); return GPIRT_E_HIP; }
    if (per_cu > 3) per_cu = 3;
    rt->reserved = h->cfg.rt_reserved > 0 ? h->cfg.rt_reserved : RT_RESERVED;
    rt->nworkers = h->cfg.rt_workers > 0 ? h->cfg.rt_workers : per_cu * (h->n_cu - rt->reserved);
    GP_HIP(hipEventCreateWithFlags(&rt->ev_done, hipEventDisableTiming));
    rt->epoch = 0;
    return 0;
}

void destroy(RtState* rt)
{
    if (!rt) return;
    if (rt->d_tasks) hipFree(rt->d_tasks);
    if (rt->d_state) hipFree(rt->d_state);
    if (rt->d_words) hipFree(rt->d_words);
    if (rt->d_in_need) hipFree(rt->d_in_need);
    if (rt->d_stage) hipFree(rt->d_stage);
    if (rt->ev_done) hipEventDestroy(rt->ev_done);
    delete rt;
}

}  // namespace

int64_t potrf_runtime_scratch_row0(int64_t nr) { return (nr + RT_W - 1) / RT_W * RT_W; }
int64_t potrf_runtime_scratch_rows() { return RT_H; }

bool potrf_runtime_usable(gpirt_handle_t h, int64_t n, int64_t nr)
{
    return h->cfg.runtime == 2 && h->cfg.panel != 2 && h->cfg.lookahead == 1 && h->cfg.nbo == RT_W && h->cfg.nbp == RT_H &&
           (n % 64) == 0 && n > 2 * RT_W && nr >= n && h->n_cu > RT_RESERVED + 64 && nr / 64 < 4000;
}

// debug: the task lists as the update workers see them (urgent queue first); returns the number copied
int potrf_runtime_tasks(gpirt_handle_t h, RtTask* host_out, int max_tasks, int* n0, int* n1)
{
    if (!h->rt) { *n0 = *n1 = 0; return 0; }
    *n0 = h->rt->ntasks[0]; *n1 = h->rt->ntasks[1];
    const int nt = std::min(max_tasks, *n0 + *n1);
    GP_HIP(hipMemcpy(host_out, h->rt->d_tasks, (size_t)nt * sizeof(RtTask), hipMemcpyDeviceToHost));
    return 0;
}

void potrf_runtime_destroy(gpirt_handle_t h)
{
    destroy(h->rt);
    h->rt = nullptr;
}

// after a hang-guard expiry: every counter is suspect -- start from a clean slate (the task lists are rebuilt lazily)
void potrf_runtime_reset(gpirt_handle_t h) { potrf_runtime_destroy(h); }

// A: (nid + 512) x n with leading dimension lda >= nid + 512; rows [nr, nid) are padding, rows [nid, nid + 512) scratch.
int potrf_runtime(gpirt_handle_t h, hipStream_t stream, double* A, int64_t n, int64_t lda, int64_t nr)
{
    if (h->rt && (h->rt->n != n || h->rt->nr != nr)) potrf_runtime_destroy(h);
    if (!h->rt) {
        GP_HIP(hipStreamSynchronize(stream));
        h->rt = new RtState();
        const int rc = build(h, h->rt, n, nr);
        if (rc) { potrf_runtime_destroy(h); return rc; }
    }
    RtState* rt = h->rt;
    if (lda < rt->nid + RT_H) { set_error("dependency-driven factorisation: the matrix buffer has no room for the identity rows"); return GPIRT_E_ARG; }
    if (!h->side) {
        int lo_pri = 0, hi_pri = 0;
        GP_HIP(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
        GP_HIP(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi_pri));
        GP_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_mid, hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&h->ev_half, hipEventDisableTiming));
    }
    hipStream_t chain = h->side;
    const unsigned long long E = ++rt->epoch;
    unsigned long long* W = rt->d_words;
    const size_t nt = (size_t)rt->ntasks[0] + (size_t)rt->ntasks[1];
    // per-factorisation words: queue heads and claim flags start at zero (the counters run on, scaled by the epoch)
    GP_HIP(hipMemsetAsync(W + rt->off_head, 0, 64 * sizeof(unsigned long long), stream));
    GP_HIP(hipMemsetAsync(W + rt->off_ctl, 0, sizeof(unsigned long long), stream));               // workers arrived
    GP_HIP(hipMemsetAsync(W + rt->off_ctl + 3, 0, sizeof(unsigned long long), stream));           // holders arrived
    GP_HIP(hipMemsetAsync(rt->d_state, 0, nt * sizeof(int), stream));
    // the identity rows: zero, then the ones of every sub-panel's block
    GP_HIP(hipMemset2DAsync(A + rt->nid, (size_t)lda * 8, 0, (size_t)RT_H * 8, (size_t)n, stream));
    hipLaunchKernelGGL(identity_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, A, lda, rt->nid, n);
    GP_HIP(hipEventRecord(h->ev_fork, stream));
    GP_HIP(hipStreamWaitEvent(chain, h->ev_fork, 0));
    // compute units for the sub-panel kernels: holders first, then the workers on what is left
    GP_TRY(launch_cu_holders(chain, W + rt->off_ctl, E, rt->reserved, h->d_info, h->rt_census));
    GP_TRY(launch_wait_counter(stream, W + rt->off_ctl + 3, (unsigned long long)rt->reserved, h->d_info));
    RtArgs a{};
    a.A = A; a.lda = lda; a.M = (int)nr; a.N = (int)n;
    a.tasks[0] = rt->d_tasks; a.tasks[1] = rt->d_tasks + rt->ntasks[0];
    a.ntasks[0] = rt->ntasks[0]; a.ntasks[1] = rt->ntasks[1];
    a.state[0] = rt->d_state; a.state[1] = rt->d_state + rt->ntasks[0];
    a.head = W + rt->off_head;
    a.nseg = rt->nseg;
    for (int d = 0; d <= rt->nseg && d < 40; ++d) a.seg_begin[d] = rt->seg_begin[(size_t)d];
    a.cnt = W; a.prog = W + rt->off_prog; a.strip = W + rt->off_strip;
    a.stage = rt->d_stage; a.ld_stage = rt->nid; a.wt_row0 = rt->nid;
    a.epoch = E; a.prog_base = E * 64ull;
    a.census = h->rt_census;
    a.ctl = W + rt->off_ctl; a.nworkers = rt->nworkers; a.quorum = rt->nworkers - rt->nworkers / 8; a.info = h->d_info;
    ProfPair pp{nullptr, nullptr, 0.0, 0, 0.0};
    if (h->prof.enabled) {
        if (!h->prof.free_pairs.empty()) { pp = h->prof.free_pairs.back(); h->prof.free_pairs.pop_back(); }
        else { GP_HIP(hipEventCreate(&pp.e0)); GP_HIP(hipEventCreate(&pp.e1)); }
        GP_HIP(hipEventRecord(pp.e0, stream));
    }
    GP_TRY(launch_update_workers(stream, a, rt->nworkers));
    if (pp.e0) {
        GP_HIP(hipEventRecord(pp.e1, stream));
        pp.flops = rt->flops; pp.bytes = 0.0; pp.cls = 1;
        h->prof.pending.push_back(pp);
    }
    // the sub-panel kernels, back to back: each is resident as soon as its predecessor leaves and starts a row block when
    // that row block's tiles are in
    h->prelast_cols = 0;
    const int64_t ntot = rt->nid + RT_H;
    for (int sid = 0; sid < rt->NS; ++sid) {
        const SubPanel& s = rt->sub[(size_t)sid];
        if (!s.valid) continue;
        PanelLink link;
        link.in_cnt = W + (size_t)rt->NS * rt->NG + (size_t)sid * rt->NG;
        link.in_need = rt->d_in_need + (size_t)sid * rt->NG;
        link.epoch = E;
        link.done_cnt = W + (size_t)sid * rt->NG;
        link.rows_per_group = RT_W;
        link.extra_row0 = rt->nid;
        link.ev_cnt = W + rt->off_ctl + 4;
        // (A launch's last row group -- two dependent tile products behind the previous sub-panel -- ends after its diagonal
        // rows, and the next launch cannot become resident before that.  Two ways around it were built and removed, both for
        // the same reason -- a work-group that spins while it HOLDS a compute unit must never be resident before everything it
        // waits for is: (i) that row group as a launch of its own on a second stream: work-groups are dealt to shader engines
        // statically and only two compute units per engine are reserved, so one of its work-groups waited for chain
        // work-groups that waited for it (guard expiry; 9.0 ms instead of 12.8 when it did not); (ii) first and second
        // sub-panels on two alternating streams: the second sub-panel's work-groups took compute units the first one's last
        // work-groups still needed and spun on them (guard expiry at n = 4096).)
        GP_TRY(launch_panel_ll(h, chain, A, ntot, lda, s.k0, s.k1, s.near_end, nullptr, &link));
        if ((sid & 1) && s.p == rt->P - 2) {
            if (!h->ev_prelast) GP_HIP(hipEventCreateWithFlags(&h->ev_prelast, hipEventDisableTiming));
            GP_HIP(hipEventRecord(h->ev_prelast, chain));
            h->prelast_cols = s.k1;
        }
    }
    GP_HIP(hipEventRecord(h->ev_join, chain));
    GP_HIP(hipStreamWaitEvent(stream, h->ev_join, 0));
    GP_HIP(hipGetLastError());
    return 0;
}

}  // namespace gpirt
