// cmf_api.hip -- C ABI (include/cmfhip.h) over the gfx950 kernels.
// Host-side orchestration of the alternating U/V/Z factor updates
// (reference: pycmf/cmf_solvers.py:248-263 MU, :510-522 Newton, :36-42 error).
#include "../../include/cmfhip.h"
#include "cmf_kernels.hip.h"
#include "cmf_gemm_pair.hip.h"
#include "cmf_eigen.hip.h"
#include "cmf_chol_mfma.hip.h"
#include "cmf_sparse.hip.h"
#include "cmf_rowhess.hip.h"
#include "cmf_bf16x6.hip.h"
#include "cmf_rowhess6.hip.h"
#include "cmf_shared64.hip.h"
#include "cmf_refine64.hip.h"
#include "cmf_eigclamp.hip.h"
#include "cmf_rank1clamp.hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <set>
#include <string>
#include <thread>
#include <vector>

using namespace cmfk;

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? CMF_ENOMEM : CMF_EHIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                 \
    } while (0)
#define CHK(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != CMF_OK) return rc_; \
    } while (0)

static inline int64_t rup(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct CsrDev {
    int64_t *indptr = nullptr;
    int32_t *idx = nullptr;
    float *val = nullptr;
    int64_t rows = 0, cols = 0, nnz = 0;
    // column-blocked regrouping for the output-stationary SpMM (cmf_sparse.hip.h), built when the gathered operand
    // exceeds the L2: entries sorted by (row group, column block, owner wave, row)
    cmfk::BcsrEntry *b_ent = nullptr;
    int64_t *b_seg = nullptr;
    int32_t *b_grow = nullptr;
    int b_ngroups = 0, b_nblocks = 0, b_rows_per_group = 0, b_nsync = 1;
    // r06: a row with far more than a wave's share of non-zeros (the hot words of a bag-of-words matrix in X^T U) is cut into pieces,
    // each an accumulator row of its own; b_vmap: accumulator row -> output row (>= 0) or -(1 + slot) of the partial buffer, whose
    // slots b_pfirst[i] .. + b_pcnt[i] are summed into output row b_prow[i] behind the launch
    int32_t *b_vmap = nullptr, *b_prow = nullptr, *b_pfirst = nullptr, *b_pcnt = nullptr;
    int b_nsplit = 0, b_nslots = 0;
};

// A captured update step: replayed with hipGraphLaunch while the key (hyper-parameters baked into
// kernel arguments) and every device pointer it references stay unchanged.
struct StepGraph {
    hipGraphExec_t exec = nullptr;
    double key[8] = {0};
    int warm = 0;       // eager runs with this key (the first one sizes every workspace)
    bool failed = false;
};

struct EvPair {
    hipEvent_t a, b;
    int cls;
    double flops;
};

struct CmfComm;
struct cmf_ctx {
    int device = 0;
    CmfComm *comm = nullptr;              // RCCL communicator of a sharded run (cmf_comm.hip.h), or none
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;

    int64_t m = 0, d = 0, p = 0;
    int k = 0;
    int64_t mp = 0, dp = 0, pp = 0;
    int kp = 0;
    bool have_problem = false;
    unsigned long long *dbg_stamps = nullptr; // set by cmf_debug_clock for its own launches only
    bool diag_ok = false;  // CMF_DIAG=1 in the environment: the timing-only knobs (wrong results) may be set
    int opt_graph = -1;    // replay MU / linear-Newton steps from a captured hipGraph: 1 always | 0 never | -1 (default) inside cmf_run
                           // only (single steps measured neutral: the ~4 us per dependent kernel boundary is device-side)
    StepGraph mu_graph, newton_graph;
    int opt_pipe_small = 4; // staging schedule of the factor-side products (0 or 4; 4 measured +5..15 %, tools/ab_small.py)
    int opt_pipe = 4;      // GEMM staging schedule (see gemm_kernel PIPE); 4 measured best (tools/ab_gemm.py)
    int opt_split = -1;    // force split-K factor (<=0: heuristic)
    int opt_class_depth = 4; // class blocks in flight per thread of class_sum_blocks_kernel (4 | 8 | 16)
    int opt_pair = 1;      // k_pad = 128, dense X and Y: the two data passes of an MU half-iteration as one balanced launch (cmf_gemm_pair.hip.h) | 0: two split launches
    int opt_tile512 = 0;   // A/B: k_pad = 128 data passes on a 512 x 128 x 16 tile (GemmCfg TILE 1) instead of 256 x 128 x 32
    int opt_rounds = 1;    // split-K heuristic of the data passes: aim at this many workgroups per CU (A/B option: 2 measured within noise of 1 at C2)
    int opt_arith_min_tiles = 8;   // ... only for operands of at least this many 256-row tiles
    int opt_arith = 0;     // data passes at k_pad = 256: 0 fp32 MFMA | 1 bf16x6 (three bf16 planes per operand, fp32-equivalent)
    int opt_ns = 1;        // flagged per-row Hessians at k_pad = 256: Newton-Schulz spectral clamp (0: Jacobi)
    bool hess_psd = true;  // the Hessians of the current step are positive semi-definite by construction (0 <= alpha <= 1)
    int opt_pipe_nt = 4;   // staging schedule of the NT (residual / error) GEMMs: 0 | 4
    int opt_refine_map = 1;     // batched float64 clamp: spectral map (lambda - pert) / (lambda + pert) in front of the sign iteration (refine_rows64_batched)
    int opt_gemm64_tile128 = 1; // batched float64 256^3 products of the refinement on 128 x 128 tiles (gemm64_tile128_kernel); 0: 32 x 32 tiles
    int opt_nt_debug = 0;  // measurement only (tools/r05_nt_probe.py): bit 0 = the error pass without its targets
    int opt_nt_bn256 = 0;  // NT passes on 256 x 256 tiles where the column extent allows (A/B option: 39.9 ms at C4 against 39.7 for the 256 x 128 x 16 tile)
    int opt_nt_raster = 0; // NT passes: XCD-aware tile order (blocks of 4 x 8 tiles per XCD; gemm_kernel) -- measured no gain at C4 (40.9 against 40.6 ms): the operands beyond L2 are not the bound
    int opt_nt_tile16 = 1; // 256 x 128 NT passes on the 16-deep K-step (two workgroups per CU) instead of 32-deep (one): 39.6 against 40.7 ms at C4
    int opt_choldiag = 0;  // timing diagnostics of chol_solve_kernel (wrong results)
    int opt_chol = 1;      // Cholesky fast path of the safe inverse (0: always Jacobi)
    int opt_chol_mfma = 1; // k_pad = 256 per-row solves: blocked Cholesky on the matrix pipe (0: the rank-1 register kernel chol_solve_kernel<16>)
    int opt_zlogit_l2 = 1;  // 0: Cython-twin numerics (Z's logit Hessian without l2 I, pyx:287-290)
    int opt_rowdiag = 0;    // diagnostic builds of row_hess_kernel<256> (CMF_DIAG_BUILD; 1: no staging, 2: stage only, 3: gather only, 4-11: cmf_rowhess.hip.h / tools/r06_rowdiag.sh)
    int opt_rowsym = 4;     // row_hess_kernel<256>: 0 full blocks | 1 upper block triangle, raw + weighted images | 3 ... one sqrt-weighted image | 4 ... and 16-wide diagonal sub-blocks
    int opt_rowstagger = 1; // row_hess_kernel: waves 4-7 stage half a K-step after waves 0-3
    int opt_rowkernel = 1; // per-row Newton sweeps: fused gather kernel (1) or masked-dense GEMMs (0)
    double flop_scale = 1.0;   // algorithmic/executed flop ratio of the launches being issued (sampled sweeps run masked-dense)
    bool dev_sampling = false; // armed for one cmf_newton_step by cmf_newton_step_device_sampled
    uint64_t dev_seed = 0;
    int64_t sample_off[3] = {0, 0, 0}; // global index of local row 0 of U / V / Z for the device sampler's keys

    float *X = nullptr, *Y = nullptr; // dense, row-major, ld = dp / pp (null while a sparse input stays native)
    CsrDev sp[2][2];                  // [X|Y][A | A^T] native CSR images
    bool sparse[2] = {false, false};
    double sp_sq[2] = {0.0, 0.0};     // sum of squares of the stored values
    int opt_sparse = 0;               // 0 auto, 1 always expand to dense, 2 always native CSR
    int opt_spmm_blocked = 1;         // column-blocked output-stationary SpMM: 0 never | 1 when the gathered operand exceeds L2 | 2 always
    int64_t opt_spmm_block_cols = 0;  // gathered rows per column block (0: 3 MB worth)
    int64_t opt_spmm_stretch = 0;     // entries of a group between two re-alignments of its XCD class (0: 8192)
    DevBuf spmm_bar;                  // rendezvous counters of the blocked SpMM (8 x 16 bytes)
    DevBuf spmm_part;                 // partial output rows of split rows (spmm_blocked_kernel)
    int opt_spmm_split = 1;           // cut rows with far more than a wave's share of non-zeros into pieces, waves by longest-piece-first (0: round 5's layout)
    float *F[3] = {nullptr, nullptr, nullptr};
    int64_t frows[3] = {0, 0, 0}, frows_pad[3] = {0, 0, 0};
    int64_t v_rows_alloc = 0;         // rows allocated behind F[CMF_V] (>= dp: the row-blocked MU driver all-gathers world equal blocks in place)

    // workspaces
    float *num = nullptr, *den = nullptr; // max(mp,dp,pp) x kp
    float *G = nullptr, *G2 = nullptr, *Hm = nullptr, *Hinv = nullptr, *Eye = nullptr; // kp x kp
    float *vbuf = nullptr;                // dp*kp + kp*kp
    DevBuf gslab32;                       // partial tiles of the small-Gram kernels
    int opt_gram32_shares = 32;           // row shares (= partial slabs) of a small Gram (C2: 8 -> 1127, 16 -> 1163, 24 -> 1166, 32 -> 1170-1177, 64 -> 1160, 128 -> 1099 it/s)
    int opt_gram32 = 1;                   // k_pad 64 / 128 Grams on gram32_partial_kernel (0: one-tile TN GEMM, split + slab sum)
    DevBuf slabs, slabs_b;                // split-K partial tiles (grow-only); second set for a product whose slabs must outlive the next one
    int slab_sel = 0;                     // which set gemm() writes
    DevBuf tickets;                       // one arrival counter per output tile of a split-K GEMM (zero between launches)
    int timed_open = 0;                   // open Timed scopes that record (see Timed)
    DevBuf narrow_tmp;                    // output image of an in-place fused update cut into column tiles (gemm(): the siblings of a row tile read all of F)
    int opt_inred = 0;                    // 0: split-K partials summed by a chip-wide kernel | 1: by the last-arriving workgroup of each tile
                                          // inside the GEMM kernel (A/B option; measured slower)
    int opt_side_gram = 0;                // A/B option: small Grams (k_pad 64 / 128) on a side stream beside the data pass that follows them
                                          // (cmf_mu_step).  Measured at C2, interleaved: 1204-1205 it/s with, 1215-1216 without -- the Gram's
                                          // workgroups take LDS bandwidth and issue slots from the data pass for longer than they save; default off
    hipStream_t side = nullptr;           // ... that stream, with the two events of a fork / join
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_pending = false;
    int opt_narrow_update = 1;            // fused factor updates with few row tiles run on 256 x 128 / 256 x 64 tiles (gemm())
    int opt_fused_mu = 1;                 // 1: F <- F num / reg(F G) in the epilogue of the F G product | 0: separate kernel
    int opt_small_tile = 1;               // 1: k_pad 64 / 128 factor updates on 64-row tiles (factor_update_kernel) | 0: gemm_kernel's 256-row tile
    DevBuf resid;                         // Newton residual / weights scratch (grow-only)
    DevBuf resid2, resid3;                // sigma' / sample weights (X side, Y side)
    DevBuf kr1, kr2;                      // Khatri-Rao squares of factors
    DevBuf hrows;                         // chunk of per-row Hessians / inverses
    DevBuf mask1, mask2;                  // stochastic sample masks (bytes)
    DevBuf lists1, lists2;                // device copies of the per-row sample index lists
    DevBuf hpart;                         // partial Hessians / gradients of the split row launches (few rows, long lists)
    int opt_ft_tile = 256;                // tile of factor_times64_kernel: 256 = 128 rows x 64 columns, eight waves (default; C5: 2.2 ms per 1e6 x 256 x 256 product) | 64 = 64 x 64 (2.4 ms) | 128 = 64 x 128 (2.8 ms)
    int opt_rowsplit = 1;                 // split the samples of a row over several workgroups when a sweep has fewer rows than CUs
    DevBuf lr_small, lr_rows;             // low-rank per-row side (sweep_v_lowrank): B / Z^T / K images; per-row p x p systems
    int opt_lowrank = 1;                  // Woodbury form of the V sweep when the per-row side has fewer samples than components
    DevBuf lists1s, lists2s, zerobuf;     // ascending copies of host-drawn lists (sparse target term); a few zero floats (zero targets)
    DevBuf cls_idx[2], cls_off[2], cls_cnt[2], cls_pat[2], hclass; // shared partial sums of linear sampled sides: class lists, pattern bytes, class images
    DevBuf certimg, certflag;             // per half group: the part of the Hessians common to its rows, and whether it alone passes the threshold test
    int opt_direct_step = 1;              // linear shared-Hessian sweeps with l1 = 0 and an unclamped inverse: F <- clamp(s (T O) H^-1) in one product
    int opt_rowcert = 1;                  // use those certificates (0: every row runs its own threshold test)
    int64_t opt_rowchunk = 0;             // > 0: cap on the rows per Hessian chunk of the per-row sweeps (tests exercise chunk boundaries)
    int opt_rowclasses = -1;              // rows per group of the shared-partial-sum form: -1 automatic, 0 / 1 row by row, 2..6 forced
    DevBuf idxbuf;                        // uploaded sample index lists
    DevBuf eigws;                         // Jacobi workspace when k_pad > 128
    DevBuf eigflag, eigcopy;              // Cholesky fast path: per-matrix fallback flags, input copy
    DevBuf clampstat;                     // [count (u64), max ||H||_F / pert (float bits)] of the float32 spectral clamp (cmf_newton_clamp_stats)
    DevBuf badbuf, rw64, rh64;            // float64 refinement of ill-conditioned rows: [count, list], sample weights, Hessians
    DevBuf ref_w, ref_w2, ref_g, ref_i, ref_ns; // its batched form (cmf_refine64.hip.h): factor images, gradients, index lists, scratch rows, clamp images
    int opt_refine_batched = 1;           // 1: listed rows redone together (64 < n_components <= 256) | 0: one at a time (round-3 path)
    std::vector<int> bad_host;            // rows of the current chunk to redo in float64 (relative to the chunk)
    int opt_refine = 1;                   // redo clamped rows with ||H||_F / pert > refine_ratio in float64 (0: float32 only, recorded)
    double opt_refine_ratio = 3.0e3;       // clamped rows: ||H||_F / pert above this (campaign: 2.7e3 -> 3e-4 off the float64 reference, 9.8e3 -> 1.9e-3)
    double opt_refine_cond = 1.0e3;        // plain Cholesky solves: max H_ii / min L_ii^2 (a LOWER bound of cond H) above this
    double opt_refine_tol = 2.0e-5;        // ... and only if eps32 * ratio * ||step_i|| > tol * ||F_i||: the float32 error bound of the row's update against the tolerance on the factor row (0: the ratio alone decides, round 5)
    int64_t opt_refine_max = (int64_t)1 << 40; // cap on the rows redone per sweep (no cap since the refinement is batched; the option remains); beyond: float32, recorded
    int64_t refined_sweep = 0, refined_total = 0;
    DevBuf bfp[2][2], bff;                // gemm_arith = 1: bf16 planes of X / Y (normal, transposed) and of the factor operand
    bool bfp_valid[2][2] = {{false, false}, {false, false}};
    DevBuf nsidx, nsws;                   // Newton-Schulz clamp: flagged-row list + counters, matrix workspaces
    DevBuf eigcl_snap;                    // snapshot of a chunk's flags (which matrices the clamp acted on) for the refinement's error-bound test
    DevBuf eigcl_ws, eigcl_log, eigcl_fail; // tridiagonal eigen-clamp (cmf_eigclamp.hip.h): d / e / Q^T g / tau, rotation logs, per-matrix give-up flags
    int opt_eig_clamp = 1;                // flagged per-row Hessians at k_pad 128 / 256: Householder + QL solve on the vector units (0: Newton-Schulz polynomials / Jacobi; 3: without the early exit of the QL iteration)
    int64_t eig_clamp_rows = 0;           // matrices served by it since the context was created (tests / bench)
    DevBuf r1_ws;                         // rank-one clamp (cmf_rank1clamp.hip.h): top eigenvectors, eigenvalues, convergence / certificate flags
    int opt_rank1_clamp = 1;              // flagged PSD Hessians with ONE eigenvalue above the threshold: power iteration + Cholesky certificate (0: off)
    int64_t rank1_rows = 0;               // matrices served by it since the context was created
    DevBuf g64a, g64b, gmix64, h64;       // float64 Grams / shared Hessian of the linear-link Newton sweeps (cmf_shared64.hip.h)
    DevBuf gslab64, w64, ns64;            // their split slabs, Cholesky workspaces + L^-1 image, Newton-Schulz images
    bool gmix64_valid = false;            // gmix64 = alpha U^T U + (1 - alpha) Z^T Z of the partials just formed (single-GPU step)
    int opt_shared64 = 1;                 // 1: shared Hessian in float64 (default) | 0: float32 Grams + float32 inverse (round-1 path)
    DevBuf hinv64, opr;                   // float64 image of the shared safe inverse; pre-conditioned operands O Hinv of the re-associated sweeps
    int opt_reassoc = 1;                  // 1: shared sweeps as F E + T (O Hinv) (cmf_newton.hip.h) | 0: gradient form F - grad Hinv
    bool v_plain = false;                 // cmf_newton_v_products -> cmf_newton_v_finish: the inverse was not clamped (E = 0)
    DevBuf dpart;                         // double partial sums
    int opt_trace_error_off = 0;          // 1: cmf_mu_step_error always takes the NT error pass (A/B, tests)
    bool mu_dots = false;                 // this MU step also leaves <U, X V> / <Z, Y^T V> in dscalar[2] / [3] (cmf_mu_step_error)
    DevBuf trace64;                       // float64 Grams of U, V, Z for the trace form of the error metric
    double dsq_cache[2] = {0.0, 0.0};     // ||X||^2, ||Y||^2 of the dense images (cmf_mu_step_error), valid while dsq_valid
    bool dsq_valid[2] = {false, false};
    double *dscalar = nullptr;            // 8 slots of 8 bytes; slot 7: sample rows gathered by class launches (unsigned long long)
    double rh_credited = 0.0, rh_gathered = 0.0; // per-row Newton accounting while timing is on: sample rows of the algorithm / gathered by row launches
    std::vector<void *> owned;            // problem-scoped allocations (released by the next cmf_set_problem)
    std::vector<void *> scratch;          // cmf_scratch_alloc buffers: live until cmf_scratch_free / cmf_ctx_destroy
    std::set<const void *> lds_opt_in;    // kernels whose >64 KB dynamic-LDS attribute is set on THIS device

    // timing
    int timing = 0;                       // 0 off | 1 events around every launch | 2 around the data-pass classes only
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> evpool;
    double ms[CMF_K_COUNT] = {0};
    int64_t launches[CMF_K_COUNT] = {0};
    double flops[CMF_K_COUNT] = {0};
    std::vector<hipEvent_t> markers;      // cmf_marker: per-iteration time series of a bench run
    // host <-> device staging: two pinned 64 MB buffers, filled / drained by several host threads while the other one is
    // on the wire (upload_strided / download_strided)
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool pin_busy[2] = {false, false};
    double sp_sum[2] = {0.0, 0.0};        // sum of the stored values of a native sparse input
};

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        (void)hipGetDevice(&prev);
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// ------------------------------------------------------------------ timing helpers
static int ev_get(cmf_ctx *c, hipEvent_t *e) {
    if (!c->evpool.empty()) {
        *e = c->evpool.back();
        c->evpool.pop_back();
        return CMF_OK;
    }
    HIPCHK(hipEventCreate(e));
    return CMF_OK;
}
struct Timed {
    cmf_ctx *c;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    bool on = false;
    double flops = 0.0;
    Timed(cmf_ctx *c_, int cls_, double flops_ = 0.0) : c(c_), cls(cls_), flops(flops_) {
        // scopes nest (a solve that falls through to the spectral clamp, the float64 refinement around its Grams and products): only
        // the OUTERMOST open scope records, so a stretch of the stream is attributed to exactly one class and the classes of an
        // iteration can never add up to more than the iteration (r04's bench lines double-counted the clamp path)
        const bool want = c->timed_open == 0 &&
                          (c->timing == 1 || (c->timing == 2 && (cls == CMF_K_GEMM_NN || cls == CMF_K_GEMM_TN || cls == CMF_K_GEMM_PAIR || cls == CMF_K_SPMM || cls == CMF_K_ROWHESS)));
        if (want && ev_get(c, &a) == CMF_OK && ev_get(c, &b) == CMF_OK) {
            on = true;
            ++c->timed_open;
            (void)hipEventRecord(a, c->stream);
        }
    }
    ~Timed() {
        if (on) {
            --c->timed_open;
            (void)hipEventRecord(b, c->stream);
            c->pending.push_back({a, b, cls, flops});
        }
    }
};
static int flush_timing(cmf_ctx *c) {
    if (c->pending.empty()) return CMF_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &e : c->pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
        c->ms[e.cls] += ms;
        c->launches[e.cls] += 1;
        c->flops[e.cls] += e.flops;
        c->evpool.push_back(e.a);
        c->evpool.push_back(e.b);
    }
    c->pending.clear();
    return CMF_OK;
}

// ------------------------------------------------------------------ memory helpers
static int dev_alloc(cmf_ctx *c, void **p, size_t bytes, bool zero = true) {
    HIPCHK(hipMalloc(p, bytes ? bytes : 16));
    c->owned.push_back(*p);
    if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes ? bytes : 16, c->stream));
    return CMF_OK;
}
static void dev_free(cmf_ctx *c, void *p) {
    if (!p) return;
    auto it = std::find(c->owned.begin(), c->owned.end(), p);
    if (it != c->owned.end()) c->owned.erase(it);
    (void)hipFree(p);
}
static void drop_graph(StepGraph &g) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    g = StepGraph();
}
static void invalidate_graphs(cmf_ctx *c) {
    drop_graph(c->mu_graph);
    drop_graph(c->newton_graph);
}

static int ensure(cmf_ctx *c, DevBuf &b, size_t bytes) {
    if (b.bytes >= bytes) return CMF_OK;
    invalidate_graphs(c); // a captured step may hold the old pointer
    if (b.p) {
        HIPCHK(hipStreamSynchronize(c->stream));
        dev_free(c, b.p);
        b.p = nullptr;
        b.bytes = 0;
    }
    CHK(dev_alloc(c, &b.p, bytes, false));
    b.bytes = bytes;
    return CMF_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember it per context, not per process
static int allow_big_lds(cmf_ctx *c, const void *fn, int bytes) {
    if (c->lds_opt_in.count(fn)) return CMF_OK;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    c->lds_opt_in.insert(fn);
    return CMF_OK;
}

// ------------------------------------------------------------------ GEMM launcher
struct GemmPlan {
    int bn;
    int ntiles_n;
    int64_t tiles_m;
    int nsplit;
    int64_t klen;
    int tile = 0; // 1: the 512 x 128 x 16 tile (GemmCfg TILE 1)
};

static GemmPlan plan_gemm(const cmf_ctx *c, int64_t mout, int64_t n, int64_t kred, bool allow_split, bool data_pass = false) {
    GemmPlan pl;
    pl.bn = n >= 256 ? 256 : (int)n; // n in {32,64,128} or a multiple of 256
    pl.ntiles_n = (int)(n / pl.bn);
    pl.tile = (c->opt_tile512 && data_pass && allow_split && pl.bn == 128 && mout % 512 == 0 && c->opt_pipe == 4) ? 1 : 0;
    pl.tiles_m = pl.tile ? mout / 512 : (mout + 255) / 256;
    const int64_t tiles = pl.tiles_m * pl.ntiles_n;
    int64_t s = 1;
    if (allow_split && c->opt_split > 0) {
        s = std::min<int64_t>(c->opt_split, std::max<int64_t>(1, kred / 32));
    } else if (allow_split && tiles < (int64_t)(c->num_cu * 3) / 4) {
        // a GEMM with too few output tiles splits its reduction so that every CU gets a workgroup (`gemm_rounds` = r: r
        // per CU for the data passes -- at C2 two rounds measured within run-to-run noise of one, three and four slower)
        const int64_t want = (int64_t)c->num_cu * (data_pass ? std::max(1, c->opt_rounds) : 1);
        s = (want + tiles / 2) / tiles;
        const int64_t maxs = std::max<int64_t>(1, kred / 128);
        s = std::max<int64_t>(1, std::min(s, maxs));
    }
    pl.klen = rup((kred + s - 1) / s, 32);
    pl.nsplit = (int)((kred + pl.klen - 1) / pl.klen);
    return pl;
}

template <int MODE, int ROLE, int PIPE>
static int launch_gemm_pipe(cmf_ctx *c, const GemmArgs &a, const GemmPlan &pl) {
    dim3 grid((unsigned)pl.tiles_m, (unsigned)pl.ntiles_n, (unsigned)pl.nsplit);
    dim3 block(512);
#define CMF_LAUNCH(BN_)                                                                          \
    do {                                                                                         \
        using Cfg = GemmCfg<MODE, BN_>;                                                          \
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&gemm_kernel<MODE, BN_, ROLE, PIPE>), (int)Cfg::LDS_BYTES)); \
        hipLaunchKernelGGL((gemm_kernel<MODE, BN_, ROLE, PIPE>), grid, block, Cfg::LDS_BYTES, c->stream, a); \
    } while (0)
    if constexpr (MODE == MODE_NT) {
        if (pl.bn == 256 && pl.tile == 0) { // 256 x 256 x 32: the data passes' tile (twice the MFMA work per barrier and per LDS fill)
            CMF_LAUNCH(256);
            HIPCHK(hipGetLastError());
            return CMF_OK;
        }
        if (pl.bn != 128) return fail(CMF_EINVAL, "NT tile width must be 128 or 256");
        if (pl.tile == 2) { // 256 x 128 x 16: two workgroups per CU (GemmCfg TILE 2)
            using Cfg = GemmCfg<MODE, 128, 2>;
            CHK(allow_big_lds(c, reinterpret_cast<const void *>(&gemm_kernel<MODE, 128, ROLE, PIPE, 2>), (int)Cfg::LDS_BYTES));
            hipLaunchKernelGGL((gemm_kernel<MODE, 128, ROLE, PIPE, 2>), grid, block, Cfg::LDS_BYTES, c->stream, a);
        } else CMF_LAUNCH(128);
    } else {
        if constexpr (ROLE == 0 && PIPE == 4) {
            if (pl.tile == 1 && pl.bn == 128) { // A/B: 512 x 128 x 16 tile
                using Cfg = GemmCfg<MODE, 128, 1>;
                CHK(allow_big_lds(c, reinterpret_cast<const void *>(&gemm_kernel<MODE, 128, ROLE, PIPE, 1>), (int)Cfg::LDS_BYTES));
                hipLaunchKernelGGL((gemm_kernel<MODE, 128, ROLE, PIPE, 1>), grid, block, Cfg::LDS_BYTES, c->stream, a);
                HIPCHK(hipGetLastError());
                return CMF_OK;
            }
        }
        switch (pl.bn) {
        case 256: CMF_LAUNCH(256); break;
        case 128: CMF_LAUNCH(128); break;
        case 64: CMF_LAUNCH(64); break;
        case 32: CMF_LAUNCH(32); break;
        default: return fail(CMF_EINVAL, "unsupported tile width %d", pl.bn);
        }
    }
#undef CMF_LAUNCH
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

template <int MODE, int ROLE>
static int launch_gemm_mode(cmf_ctx *c, const GemmArgs &a, const GemmPlan &pl) {
    if (ROLE == 0 && MODE != MODE_NT) {
        if (c->opt_pipe == 1) return launch_gemm_pipe<MODE, ROLE, (ROLE == 0 && MODE != MODE_NT) ? 1 : 0>(c, a, pl);
        if (c->opt_pipe == 2) return launch_gemm_pipe<MODE, ROLE, (ROLE == 0 && MODE != MODE_NT) ? 2 : 0>(c, a, pl);
        if (c->opt_pipe == 3) return launch_gemm_pipe<MODE, ROLE, (ROLE == 0 && MODE != MODE_NT) ? 3 : 0>(c, a, pl);
        if (c->opt_pipe == 4) return launch_gemm_pipe<MODE, ROLE, (ROLE == 0 && MODE != MODE_NT) ? 4 : 0>(c, a, pl);
        if (c->opt_pipe == 5) return launch_gemm_pipe<MODE, ROLE, (ROLE == 0 && MODE != MODE_NT) ? 5 : 0>(c, a, pl);
        if (c->opt_pipe == 10) return launch_gemm_pipe<MODE, ROLE, (ROLE == 0 && MODE != MODE_NT) ? 10 : 0>(c, a, pl);
    }
    if (ROLE == 1 && MODE != MODE_NT && c->opt_pipe_small == 4)
        return launch_gemm_pipe<MODE, ROLE, (ROLE == 1 && MODE != MODE_NT) ? 4 : 0>(c, a, pl);
    if (MODE == MODE_NT && c->opt_pipe_nt == 4) return launch_gemm_pipe<MODE, ROLE, (MODE == MODE_NT) ? 4 : 0>(c, a, pl);
    return launch_gemm_pipe<MODE, ROLE, 0>(c, a, pl);
}

static int sum_slabs(cmf_ctx *c, float *dst, const float *src, int64_t n, int nslab, int64_t stride,
                     bool accumulate) {
    Timed tm(c, CMF_K_ELEMWISE);
    const int64_t n4 = n / 4;
    const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks), dim3(256), 0, c->stream, dst, src, n4, nslab, stride,
                       accumulate ? 1 : 0);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

static int mu_apply(cmf_ctx *c, float *F, const float *num, const float *den, int64_t n, double l1, double l2);

struct SlabRef { // a split-K result left unreduced for a consumer that sums the slabs itself (factor_update_kernel)
    const float *base = nullptr;
    int nslab = 0;
    int64_t stride = 0;
    int64_t quota = 0, unit0 = 0; // slabs of a paired launch (quota > 0): a 256-row tile has pair_slots(quota, unit0, ksteps, tile) of them
    int ksteps = 0;
};

struct Epilogue { // fused factor update in the epilogue of a factor-side product (gemm_kernel, epi != 0)
    int kind = 0;           // EPI_MU | EPI_GRAD | EPI_APPLY
    const float *F = nullptr, *P = nullptr;
    const SlabRef *Pslabs = nullptr, *Pslabs2 = nullptr; // P (additionally) as unreduced slabs (small-tile kernel only)
    float *out = nullptr;
    double a = 0.0, b = 0.0, c = 0.0;
    int64_t rows = 0;
    int kvalid = 0, nn = 0;
};

// C[mout x n] (+)= op(A) * B.  mode NN: A is [mout_pad x kred]; TN: A is [kred x >=mout].
// Result lands in `out` (ld = n); split-K partials go through the slab workspace and are summed in slab order -- by the
// last-arriving workgroup of each output tile inside the GEMM kernel (default) or by a separate kernel.
// `A` is a factor-sized operand when lda == k_pad (Grams, F*G, step products): those launches
// use the ROLE=1 symbol and the CMF_K_GEMM_SMALL timing class.
static int gemm(cmf_ctx *c, int mode, const float *A, int64_t lda, const float *B, int64_t ldb, float *out,
                int64_t mout, int64_t n, int64_t kred, bool accumulate = false, const Epilogue *mu = nullptr, SlabRef *defer = nullptr) {
    const bool data_pass = (lda != c->kp);
    DevBuf &slabbuf = c->slab_sel ? c->slabs_b : c->slabs;
    if (kred % 32 || n % 32) return fail(CMF_EINVAL, "gemm: unpadded extent (k=%lld n=%lld)", (long long)kred, (long long)n);
    GemmPlan pl = plan_gemm(c, mout, n, kred, mu == nullptr, data_pass);
    if (mu && c->opt_narrow_update && pl.ntiles_n == 1 && pl.tile == 0) {
        // a factor-side product with a fused update cannot split its reduction (K = k_pad), and one 256 x 256 x 256 tile is 55 us of
        // one CU whatever the row count: with few row tiles (a rank's 8192-row shard of U at C4 / 8 GPUs: 32) cut the tile's
        // columns instead -- same arithmetic per element, bit-identical results, 4 x the workgroups
        while (pl.bn > 64 && pl.tiles_m * (n / pl.bn) * 2 <= (int64_t)c->num_cu) pl.bn /= 2;
        pl.ntiles_n = (int)(n / pl.bn);
    }
    // An in-place update (EPI_MU / EPI_COMBINE: A == out == F) on column tiles would let a late workgroup stream columns of F its
    // siblings of the same row tile have already overwritten: the column tiles write a scratch image that is copied
    // over F behind the launch (stream order).  Products whose output does not alias A (EPI_APPLY / DIRECT / GRAD) write directly.
    float *alias_out = nullptr;
    if (mu && pl.ntiles_n > 1 && mu->out == A) {
        CHK(ensure(c, c->narrow_tmp, (size_t)rup(mout, 256) * n * sizeof(float)));
        alias_out = mu->out;
    }
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb;
    a.ldc = n;
    a.Mout = (mode == MODE_TN) ? mout : rup(mout, 256);
    a.Kred = kred; a.klen = pl.klen;
    a.dbg = data_pass ? c->dbg_stamps : nullptr;
    const int64_t rows_store = (mode == MODE_TN) ? mout : rup(mout, 256);
    // in-kernel forms: fused update epilogue; out += acc of an unsplit accumulate; (option) last-arriver reduction of splits
    const bool in_kernel = mu != nullptr || pl.nsplit == 1 || (c->opt_inred != 0 && (int64_t)pl.tiles_m * pl.ntiles_n >= 16 && pl.nsplit <= 16);
    const bool direct = (pl.nsplit == 1 && !accumulate);
    if (mu) {
        if (mode != MODE_NN || pl.nsplit != 1) return fail(CMF_EINVAL, "fused update needs an unsplit NN product");
        a.epi = mu->kind; a.epi_F = mu->F; a.epi_P = mu->P; a.epi_out = alias_out ? (float *)c->narrow_tmp.p : mu->out;
        a.epi_a = (float)mu->a; a.epi_b = (float)mu->b; a.epi_c = (float)mu->c;
        a.epi_rows = mu->rows; a.epi_kvalid = mu->kvalid; a.epi_nn = mu->nn;
        a.C = out; // unused
    } else if (direct) {
        a.C = out;
        a.slab_stride = 0;
    } else if (in_kernel && pl.nsplit == 1) {
        a.red_out = out; a.red_acc = 1; // out += acc in the epilogue
        a.C = out;
    } else {
        const size_t need = (size_t)pl.nsplit * rows_store * n * sizeof(float);
        CHK(ensure(c, slabbuf, need));
        a.C = (float *)slabbuf.p;
        a.slab_stride = rows_store * n;
        if (in_kernel) {
            const size_t ntile = (size_t)pl.tiles_m * pl.ntiles_n;
            if (c->tickets.bytes < ntile * sizeof(unsigned)) {
                CHK(ensure(c, c->tickets, std::max<size_t>(4096, 2 * ntile) * sizeof(unsigned)));
                HIPCHK(hipMemsetAsync(c->tickets.p, 0, c->tickets.bytes, c->stream));
            }
            a.red_out = out; a.red_acc = accumulate ? 1 : 0; a.ticket = (unsigned *)c->tickets.p;
        }
    }
    {
        Timed tm(c, !data_pass ? CMF_K_GEMM_SMALL : (mode == MODE_NN ? CMF_K_GEMM_NN : CMF_K_GEMM_TN),
                 2.0 * (double)mout * (double)n * (double)kred * c->flop_scale);
        if (mode == MODE_NN) {
            if (data_pass) CHK((launch_gemm_mode<MODE_NN, 0>(c, a, pl)));
            else CHK((launch_gemm_mode<MODE_NN, 1>(c, a, pl)));
        } else {
            if (data_pass) CHK((launch_gemm_mode<MODE_TN, 0>(c, a, pl)));
            else CHK((launch_gemm_mode<MODE_TN, 1>(c, a, pl)));
        }
    }
    if (alias_out) HIPCHK(hipMemcpyAsync(alias_out, c->narrow_tmp.p, (size_t)rup(mout, 256) * n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    if (!direct && !in_kernel) {
        if (defer && !accumulate) { // the consumer sums the slabs (valid until the next GEMM reuses the slab workspace)
            defer->base = (const float *)slabbuf.p; defer->nslab = pl.nsplit; defer->stride = a.slab_stride;
            return CMF_OK;
        }
        CHK(sum_slabs(c, out, (const float *)slabbuf.p, rows_store * n, pl.nsplit, a.slab_stride, accumulate));
    }
    if (defer) *defer = SlabRef();
    return CMF_OK;
}

// G (k_pad x k_pad) = F^T F over rows_pad rows of a factor-sized operand: the dedicated small-Gram kernels at k_pad 64 / 128
// (cmf_kernels.hip.h), else the TN GEMM
// Fork / join of the side stream: a small Gram whose consumer sits BEHIND the next data pass (U^T U + Z^T Z before the V update,
// V^T V before the U / Z updates) is launched on a second stream that starts where the main stream stands and is joined in front
// of the consumer -- its 96 small workgroups (24 KB of LDS, 256 threads) run in the slots the data pass leaves free on every CU
// (one 512-thread workgroup with 98 KB of LDS at k_pad = 128) instead of in front of it.  Captured into the step graph as a fork.
static bool side_gram_ok(const cmf_ctx *c, int64_t rows_pad) {
    return c->opt_side_gram && c->timing != 1 && c->opt_gram32 && (c->kp == 64 || c->kp == 128) && rows_pad >= 1024 && c->opt_arith == 0;
}
static int side_fork(cmf_ctx *c) {
    if (!c->side) {
        HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    return CMF_OK;
}
static int side_join(cmf_ctx *c) {
    if (!c->side_pending) return CMF_OK;
    HIPCHK(hipEventRecord(c->ev_join, c->side));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    c->side_pending = false;
    return CMF_OK;
}

static int gram32(cmf_ctx *c, const float *F, int64_t rows_pad, float *G, bool on_side = false) {
    // (k_pad = 256, C4: measured equal to the TN GEMM + slab sum, 0.70 against 0.53 + 0.13 ms per iteration -- not used there)
    if (!(c->opt_gram32 && (c->kp == 64 || c->kp == 128) && rows_pad >= 1024))
        return gemm(c, MODE_TN, F, c->kp, F, c->kp, G, c->kp, c->kp, rows_pad);
    CHK(side_join(c)); // (the partial slabs are shared: one small Gram in flight at a time)
    const int T = c->kp / 64, ntile = T * (T + 1) / 2;
    // row shares: 32 for a few thousand rows (C2); long factors (C4: 131072 stacked rows) take one share per 1024 rows
    int64_t nsplit = std::min<int64_t>(std::max<int64_t>(c->opt_gram32_shares, std::min<int64_t>(256, rows_pad / 1024)), rows_pad / 32);
    const int64_t chunk = rup((rows_pad + nsplit - 1) / nsplit, 32);
    nsplit = (rows_pad + chunk - 1) / chunk;
    CHK(ensure(c, c->gslab32, (size_t)nsplit * ntile * 64 * 64 * sizeof(float)));
    hipStream_t st = c->stream;
    if (on_side && side_gram_ok(c, rows_pad)) {
        CHK(side_fork(c));
        st = c->side;
        c->side_pending = true;
    }
    Timed tm(c, CMF_K_GEMM_SMALL, 2.0 * (double)rows_pad * c->kp * c->kp);
    hipLaunchKernelGGL(gram32_partial_kernel, dim3((unsigned)ntile, (unsigned)nsplit), dim3(256), 0, st, F, c->kp, rows_pad, chunk, (float *)c->gslab32.p);
    hipLaunchKernelGGL(gram32_reduce_kernel, dim3((unsigned)std::min(64, (c->kp * c->kp + 255) / 256)), dim3(256), 0, st, (const float *)c->gslab32.p,
                       c->kp, (int)nsplit, G);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// rows x k_pad x k_pad product with a fused factor update on 64-row tiles (k_pad 64 / 128): see factor_update_kernel
static bool small_tile_ok(const cmf_ctx *c, int64_t rows_pad) {
    return c->opt_fused_mu && c->opt_small_tile && (c->kp == 64 || c->kp == 128) && rows_pad * c->kp <= ((int64_t)1 << 27);
}
static void factor_update_args(FactorUpdArgs &g, const float *A, const float *B, const Epilogue &e) {
    memset(&g, 0, sizeof g);
    g.A = A; g.B = B; g.epi = e.kind; g.F = e.F; g.P = e.P; g.out = e.out;
    if (e.Pslabs && e.Pslabs->nslab > 0) {
        g.S1 = e.Pslabs->base; g.n1 = e.Pslabs->nslab; g.stride1 = e.Pslabs->stride;
        g.q1 = e.Pslabs->quota; g.u1 = e.Pslabs->unit0; g.ks1 = e.Pslabs->ksteps;
    }
    if (e.Pslabs2 && e.Pslabs2->nslab > 0) {
        g.S2 = e.Pslabs2->base; g.n2 = e.Pslabs2->nslab; g.stride2 = e.Pslabs2->stride;
        g.q2 = e.Pslabs2->quota; g.u2 = e.Pslabs2->unit0; g.ks2 = e.Pslabs2->ksteps;
    }
    g.a = (float)e.a; g.b = (float)e.b; g.c = (float)e.c; g.rows_valid = e.rows; g.kvalid = e.kvalid; g.nn = e.nn;
}
// two MU updates with the same Gram in ONE launch (k_pad = 128): cmf_solvers.py:233-234 and :239-240 behind the paired data passes
static int factor_update2(cmf_ctx *c, const float *A0, const Epilogue &e0, int64_t rows0, const float *A1, const Epilogue &e1, int64_t rows1, const float *B) {
    FactorUpdArgs g0, g1;
    factor_update_args(g0, A0, B, e0);
    factor_update_args(g1, A1, B, e1);
    Timed tm(c, CMF_K_GEMM_SMALL, 2.0 * (double)(rows0 + rows1) * c->kp * c->kp);
    const size_t lds = (size_t)(128 * 128 + 64 * 132) * sizeof(float);
    CHK(allow_big_lds(c, reinterpret_cast<const void *>(&factor_update2_kernel<128>), (int)lds));
    hipLaunchKernelGGL((factor_update2_kernel<128>), dim3((unsigned)((rows0 + rows1) / 64)), dim3(256), lds, c->stream, g0, g1, (int)(rows0 / 64));
    HIPCHK(hipGetLastError());
    return CMF_OK;
}
static int factor_update(cmf_ctx *c, const float *A, const float *B, const Epilogue &e, int64_t rows_pad) {
    FactorUpdArgs g;
    factor_update_args(g, A, B, e);
    Timed tm(c, CMF_K_GEMM_SMALL, 2.0 * (double)rows_pad * c->kp * c->kp);
    const dim3 grid((unsigned)(rows_pad / 64));
    if (c->kp == 128) {
        const size_t lds = (size_t)(128 * 128 + 64 * 132) * sizeof(float);
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&factor_update_kernel<128>), (int)lds));
        hipLaunchKernelGGL((factor_update_kernel<128>), grid, dim3(256), lds, c->stream, g);
    } else {
        const size_t lds = (size_t)(64 * 64 + 64 * 68) * sizeof(float);
        hipLaunchKernelGGL((factor_update_kernel<64>), grid, dim3(256), lds, c->stream, g);
    }
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// F <- F * num / reg(F G)   (MUSolver._regularized_delta, cmf_solvers.py:212-228): one launch when the factor-side
// product has a single N tile (k_pad <= 256), else product + elementwise kernel
static int mu_update(cmf_ctx *c, float *F, const float *G, const float *num, int64_t rows_pad, double l1, double l2,
                     const SlabRef *num_slabs = nullptr, const SlabRef *num_slabs2 = nullptr) {
    if (small_tile_ok(c, rows_pad)) {
        Epilogue e;
        e.kind = EPI_MU; e.F = F; e.out = F; e.a = l1; e.b = l2; e.c = 1.1920928955078125e-07;
        // a deferred product left `num` unwritten: its value is the sum of its slabs
        e.P = (num_slabs && num_slabs->nslab > 0) ? nullptr : num;
        e.Pslabs = num_slabs; e.Pslabs2 = num_slabs2;
        return factor_update(c, F, G, e, rows_pad);
    }
    if (c->opt_fused_mu && c->kp <= 256) {
        Epilogue e;
        e.kind = EPI_MU; e.F = F; e.P = num; e.out = F; e.a = l1; e.b = l2; e.c = 1.1920928955078125e-07;
        return gemm(c, MODE_NN, F, c->kp, G, c->kp, c->den, rows_pad, c->kp, c->kp, false, &e);
    }
    CHK(gemm(c, MODE_NN, F, c->kp, G, c->kp, c->den, rows_pad, c->kp, c->kp));
    return mu_apply(c, F, num, c->den, rows_pad * c->kp, l1, l2);
}

struct NtOut {
    const float *T = nullptr; int64_t ldt = 0;
    float *R = nullptr; float *W = nullptr; int64_t ldr = 0;
    const uint8_t *mask = nullptr; int64_t ldm = 0;
    double *sq = nullptr; // device scalar receiving the total
    float scale_r = 1.f, scale_w = 1.f;
    int link = 0, w_is_slope = 0;
};

// S = L[rows x kp] * Rt[cols x kp]^T with fused epilogue
static int gemm_nt(cmf_ctx *c, const float *L, int64_t rows_pad, int64_t rows_valid, const float *Rt,
                   int64_t cols_pad, int64_t cols_valid, const NtOut &o) {
    GemmPlan pl;
    pl.bn = 128; pl.ntiles_n = (int)(cols_pad / 128); pl.tiles_m = rows_pad / 256; pl.nsplit = 1; pl.klen = c->kp;
    pl.tile = c->opt_nt_tile16 ? 2 : 0;
    if (c->opt_nt_bn256 && cols_pad % 256 == 0 && (rows_pad / 256) * (cols_pad / 256) >= 2 * (int64_t)c->num_cu) {
        pl.bn = 256; pl.ntiles_n = (int)(cols_pad / 256); pl.tile = 0;
    }
    int ras_rb = 0, ras_cb = 0;
    if (c->opt_nt_raster && pl.ntiles_n % 8 == 0 && pl.tiles_m * pl.ntiles_n >= 8 * 64) {
        const int per_x = pl.ntiles_n / 8;
        ras_cb = per_x % 8 == 0 ? 8 : (per_x % 4 == 0 ? 4 : (per_x % 2 == 0 ? 2 : 1));
        ras_rb = 32 / ras_cb;
        while (ras_rb > 1 && pl.tiles_m % ras_rb) ras_rb /= 2;
    }
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.A = L; a.lda = c->kp; a.B = Rt; a.ldb = c->kp;
    a.Kred = c->kp; a.klen = c->kp; a.Mout = rows_pad;
    a.T = o.T; a.ldt = o.ldt; a.R = o.R; a.W = o.W; a.ldr = o.ldr; a.mask = o.mask; a.ldm = o.ldm;
    a.Mvalid = rows_valid; a.Nvalid = cols_valid;
    a.scale_r = o.scale_r; a.scale_w = o.scale_w; a.link = o.link; a.w_is_slope = o.w_is_slope;
    a.ras_rb = ras_rb; a.ras_cb = ras_cb;
    const int64_t nwg = pl.tiles_m * pl.ntiles_n;
    if (o.sq) {
        CHK(ensure(c, c->dpart, (size_t)nwg * sizeof(double)));
        a.sq_out = (double *)c->dpart.p;
    }
    {
        Timed tm(c, CMF_K_GEMM_NT, 2.0 * (double)rows_valid * (double)cols_valid * (double)c->kp * c->flop_scale);
        CHK((launch_gemm_mode<MODE_NT, 0>(c, a, pl)));
    }
    if (o.sq) {
        Timed tm(c, CMF_K_ELEMWISE);
        hipLaunchKernelGGL(sum_doubles_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)c->dpart.p, nwg, o.sq);
        HIPCHK(hipGetLastError());
    }
    return CMF_OK;
}

static int mu_apply(cmf_ctx *c, float *F, const float *num, const float *den, int64_t n, double l1, double l2) {
    Timed tm(c, CMF_K_ELEMWISE);
    const int64_t n4 = n / 4;
    const int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(mu_apply_kernel, dim3(blocks), dim3(256), 0, c->stream, F, num, den, n4, (float)l1, (float)l2,
                       1.1920928955078125e-07f);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

#include "cmf_sparse_host.hip.h"

// ------------------------------------------------------------------ C ABI: basics
extern "C" const char *cmf_last_error(void) { return g_err.c_str(); }

// sha256 of the sources this library was compiled from (pycmf_amd/build.py passes it; _lib.load() compares it with the tree)
#ifndef CMF_SOURCE_HASH
#define CMF_SOURCE_HASH "unstamped"
#endif
// (the stamp is also findable in the file's bytes behind the marker, so that the build script never has to dlopen the library)
static const char g_source_stamp[] = "cmfhip-source-sha256:" CMF_SOURCE_HASH;
extern "C" const char *cmf_source_hash(void) { return g_source_stamp + sizeof("cmfhip-source-sha256:") - 1; }

extern "C" int cmf_device_count(int *count) {
    if (!count) return fail(CMF_EINVAL, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return CMF_OK;
}

extern "C" int cmf_ctx_create(cmf_ctx **out, int device, void *stream) {
    if (!out) return fail(CMF_EINVAL, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(CMF_ENODEV, "no HIP device visible: libcmfhip needs an MI355X (gfx950); there is no CPU fallback");
    if (device < 0 || device >= n) return fail(CMF_EINVAL, "device %d out of range (have %d)", device, n);
    DeviceGuard dg(device); // leaves the calling thread's current device as it was
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CMF_ENODEV, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    cmf_ctx *c = new cmf_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return fail(CMF_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        c->own_stream = true;
    }
    c->diag_ok = getenv("CMF_DIAG") != nullptr && atoi(getenv("CMF_DIAG")) != 0;
    if (const char *e = getenv("CMF_GEMM_PIPE")) { // A/B hook for the test-suite: staging schedule of the data-pass GEMMs
        const int v = atoi(e);
        if ((v >= 0 && v <= 5) || v == 10) c->opt_pipe = v;
    }
    void *ds = nullptr;
    int rc = dev_alloc(c, &ds, 8 * sizeof(double));
    if (rc != CMF_OK) {
        delete c;
        return rc;
    }
    c->dscalar = (double *)ds;
    *out = c;
    return CMF_OK;
}

static void release_problem(cmf_ctx *c) {
    (void)hipStreamSynchronize(c->stream);
    invalidate_graphs(c);
    for (void *p : c->owned)
        if (p != (void *)c->dscalar) (void)hipFree(p);
    c->owned.clear();
    c->owned.push_back(c->dscalar);
    c->X = c->Y = nullptr;
    c->F[0] = c->F[1] = c->F[2] = nullptr;
    c->num = c->den = c->G = c->G2 = c->Hm = c->Hinv = c->Eye = c->vbuf = nullptr;
    c->slabs = DevBuf(); c->slabs_b = DevBuf(); c->gslab32 = DevBuf(); c->slab_sel = 0; c->tickets = DevBuf(); c->narrow_tmp = DevBuf(); c->resid = DevBuf(); c->resid2 = DevBuf(); c->resid3 = DevBuf(); c->dpart = DevBuf();
    c->kr1 = DevBuf(); c->kr2 = DevBuf(); c->hrows = DevBuf(); c->mask1 = DevBuf(); c->mask2 = DevBuf();
    c->lists1 = DevBuf(); c->lists2 = DevBuf(); c->lists1s = DevBuf(); c->lists2s = DevBuf(); c->zerobuf = DevBuf(); c->lr_small = DevBuf(); c->lr_rows = DevBuf(); c->hpart = DevBuf();
    for (int q = 0; q < 2; ++q) { c->cls_idx[q] = DevBuf(); c->cls_off[q] = DevBuf(); c->cls_cnt[q] = DevBuf(); c->cls_pat[q] = DevBuf(); }
    c->hclass = DevBuf(); c->certimg = DevBuf(); c->certflag = DevBuf();
    c->idxbuf = DevBuf(); c->eigws = DevBuf(); c->eigflag = DevBuf(); c->eigcopy = DevBuf(); c->clampstat = DevBuf(); c->badbuf = DevBuf(); c->rw64 = DevBuf(); c->rh64 = DevBuf(); c->bad_host.clear();
    c->ref_w = DevBuf(); c->ref_w2 = DevBuf(); c->ref_g = DevBuf(); c->ref_i = DevBuf(); c->ref_ns = DevBuf();
    c->trace64 = DevBuf(); c->dsq_valid[0] = c->dsq_valid[1] = false;
    c->nsidx = DevBuf(); c->nsws = DevBuf(); c->eigcl_ws = DevBuf(); c->eigcl_log = DevBuf(); c->eigcl_fail = DevBuf(); c->eigcl_snap = DevBuf(); c->r1_ws = DevBuf();
    c->spmm_bar = DevBuf(); c->spmm_part = DevBuf();
    c->g64a = DevBuf(); c->g64b = DevBuf(); c->gmix64 = DevBuf(); c->h64 = DevBuf();
    c->gslab64 = DevBuf(); c->w64 = DevBuf(); c->ns64 = DevBuf();
    c->hinv64 = DevBuf(); c->opr = DevBuf(); c->v_plain = false;
    c->gmix64_valid = false;
    for (int w = 0; w < 2; ++w)
        for (int o = 0; o < 2; ++o) { c->bfp[w][o] = DevBuf(); c->bfp_valid[w][o] = false; }
    c->bff = DevBuf();
    c->have_problem = false;
    for (int w = 0; w < 2; ++w) {
        c->sparse[w] = false;
        c->sp_sq[w] = 0.0;
        for (int t = 0; t < 2; ++t) c->sp[w][t] = CsrDev();
    }
}

static void pin_give(void *p);
extern "C" int cmf_comm_destroy(cmf_ctx *c);
extern "C" int cmf_ctx_destroy(cmf_ctx *c) {
    if (!c) return CMF_OK;
    DeviceGuard dg(c->device);
    (void)cmf_comm_destroy(c);
    release_problem(c);
    for (void *p : c->scratch) (void)hipFree(p);
    c->scratch.clear();
    (void)hipFree(c->dscalar);
    for (auto &e : c->pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto e : c->evpool) (void)hipEventDestroy(e);
    for (auto e : c->markers) (void)hipEventDestroy(e);
    for (int b = 0; b < 2; ++b) {
        pin_give(c->pin[b]);
        if (c->pin_ev[b]) (void)hipEventDestroy(c->pin_ev[b]);
    }
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return CMF_OK;
}

extern "C" int cmf_set_option(cmf_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return fail(CMF_EINVAL, "null argument");
    invalidate_graphs(c);
    if (!strcmp(name, "graph")) {
        c->opt_graph = value < 0 ? -1 : (value != 0);
        return CMF_OK;
    }
    if (!strcmp(name, "gemm_pipe")) {
        if (value < 0 || (value > 5 && value != 10)) return fail(CMF_EINVAL, "gemm_pipe must be 0..5 or 10");
        c->opt_pipe = (int)value;
    } else if (!strcmp(name, "gemm_pipe_nt")) {
        c->opt_pipe_nt = (int)value;
    } else if (!strcmp(name, "gemm_pipe_small")) {
        c->opt_pipe_small = (int)value;
    } else if (!strcmp(name, "gemm_split")) {
        c->opt_split = (int)value;
    } else if (!strcmp(name, "small_gram_shares")) {
        c->opt_gram32_shares = (int)std::max<int64_t>(1, std::min<int64_t>(value, 1024));
    } else if (!strcmp(name, "small_gram")) {
        c->opt_gram32 = value != 0;
    } else if (!strcmp(name, "chol_mfma")) {
        c->opt_chol_mfma = value != 0;
    } else if (!strcmp(name, "side_gram")) {
        c->opt_side_gram = value != 0;
    } else if (!strcmp(name, "refine_spectral_map")) {
        c->opt_refine_map = value != 0;
    } else if (!strcmp(name, "gemm64_tile128")) {
        c->opt_gemm64_tile128 = value != 0;
    } else if (!strcmp(name, "nt_debug")) {
        c->opt_nt_debug = (int)value;
    } else if (!strcmp(name, "nt_bn256")) {
        c->opt_nt_bn256 = value != 0;
    } else if (!strcmp(name, "nt_raster")) {
        c->opt_nt_raster = value != 0;
    } else if (!strcmp(name, "nt_tile16")) {
        c->opt_nt_tile16 = value != 0;
    } else if (!strcmp(name, "narrow_update")) {
        c->opt_narrow_update = value != 0;
    } else if (!strcmp(name, "class_sum_depth")) {
        c->opt_class_depth = (int)value;
    } else if (!strcmp(name, "pair_passes")) {
        c->opt_pair = (int)value; // 0: split launches | 1: paired data passes, U and Z updates in one launch | 2: paired data passes, separate updates
    } else if (!strcmp(name, "gemm_tile512")) {
        c->opt_tile512 = value != 0;
    } else if (!strcmp(name, "gemm_rounds")) {
        c->opt_rounds = (int)std::max<int64_t>(1, value);
    } else if (!strcmp(name, "z_logit_hessian_l2")) {
        c->opt_zlogit_l2 = value != 0;
    } else if (!strcmp(name, "row_diag")) {
#ifndef CMF_DIAG_BUILD
        if (value != 0) return fail(CMF_EINVAL, "row_diag is a timing-only diagnostic (wrong results) that the default build does not carry: "
                                                "build with `python -m pycmf_amd.build --diag` and set CMF_DIAG=1");
#endif
        if (value != 0 && !c->diag_ok) return fail(CMF_EINVAL, "row_diag is a timing-only diagnostic (wrong results): set CMF_DIAG=1 in the environment to allow it");
        c->opt_rowdiag = (int)value;
    } else if (!strcmp(name, "row_stagger")) {
        c->opt_rowstagger = value != 0;
    } else if (!strcmp(name, "row_symmetric")) {
        c->opt_rowsym = (int)value;
    } else if (!strcmp(name, "row_kernel")) {
        c->opt_rowkernel = value != 0;
    } else if (!strcmp(name, "direct_newton_step")) {
        c->opt_direct_step = value != 0;
    } else if (!strcmp(name, "row_certificates")) {
        c->opt_rowcert = value != 0;
    } else if (!strcmp(name, "row_classes")) {
        if (value < -1 || value > 6) return fail(CMF_EINVAL, "row_classes: -1 (automatic), 0 (off) or 2..6 rows per group");
        c->opt_rowclasses = (int)value;
    } else if (!strcmp(name, "row_chunk")) {
        c->opt_rowchunk = std::max<int64_t>(0, value);
    } else if (!strcmp(name, "sample_row_offset_u")) {
        c->sample_off[CMF_U] = value;
    } else if (!strcmp(name, "sample_row_offset_v")) {
        c->sample_off[CMF_V] = value;
    } else if (!strcmp(name, "sample_row_offset_z")) {
        c->sample_off[CMF_Z] = value;
    } else if (!strcmp(name, "gemm_arith")) {
        if (value != 0 && value != 1) return fail(CMF_EINVAL, "gemm_arith must be 0 (fp32 MFMA) or 1 (bf16x6)");
        c->opt_arith = (int)value;
    } else if (!strcmp(name, "gemm_arith_min_tiles")) {
        c->opt_arith_min_tiles = (int)std::max<int64_t>(1, value);
    } else if (!strcmp(name, "trace_error")) {
        c->opt_trace_error_off = value == 0;
    } else if (!strcmp(name, "eig_clamp")) {
        c->opt_eig_clamp = (int)value;
    } else if (!strcmp(name, "rank1_clamp")) {
        c->opt_rank1_clamp = value != 0;
    } else if (!strcmp(name, "newton_schulz")) {
        c->opt_ns = value != 0;
    } else if (!strcmp(name, "chol_diag")) {
#ifndef CMF_DIAG_BUILD
        if (value != 0) return fail(CMF_EINVAL, "chol_diag is a timing-only diagnostic (wrong results) that the default build does not carry: "
                                                "build with `python -m pycmf_amd.build --diag` and set CMF_DIAG=1");
#endif
        if (value != 0 && !c->diag_ok) return fail(CMF_EINVAL, "chol_diag is a timing-only diagnostic (wrong results): set CMF_DIAG=1 in the environment to allow it");
        c->opt_choldiag = (int)value;
    } else if (!strcmp(name, "safe_inverse_cholesky")) {
        c->opt_chol = value != 0;
    } else if (!strcmp(name, "split_reduce_in_kernel")) {
        c->opt_inred = value != 0;
    } else if (!strcmp(name, "small_tile_update")) {
        c->opt_small_tile = value != 0;
    } else if (!strcmp(name, "fused_mu_update")) {
        c->opt_fused_mu = value != 0;
    } else if (!strcmp(name, "shared_hessian_f64")) {
        c->opt_shared64 = value != 0;
    } else if (!strcmp(name, "factor_times_tile")) {
        c->opt_ft_tile = (value == 64 || value == 256) ? (int)value : 128;   // 64: 64 x 64 tile | 128: 64 x 128 | 256: 128 rows x 64 columns, eight waves
    } else if (!strcmp(name, "refine_rows")) {
        c->opt_refine = value != 0;
    } else if (!strcmp(name, "refine_rows_batched")) {
        c->opt_refine_batched = value != 0;
    } else if (!strcmp(name, "refine_rows_ratio")) {
        c->opt_refine_ratio = (double)std::max<int64_t>(1, value);
    } else if (!strcmp(name, "refine_rows_cond")) {
        c->opt_refine_cond = (double)std::max<int64_t>(1, value);
    } else if (!strcmp(name, "refine_rows_tol_ppm")) {
        c->opt_refine_tol = 1e-6 * (double)std::max<int64_t>(0, value);
    } else if (!strcmp(name, "refine_rows_max")) {
        c->opt_refine_max = std::max<int64_t>(0, value);
    } else if (!strcmp(name, "row_split")) {
        c->opt_rowsplit = value != 0;
    } else if (!strcmp(name, "lowrank_rows")) {
        if (value < 0 || value > 2) return fail(CMF_EINVAL, "lowrank_rows: 0 off, 1 on (p x p systems through chol_solve_kernel), 2 on (one wave per system in registers: A/B, slower)");
        c->opt_lowrank = (int)value;
    } else if (!strcmp(name, "newton_reassoc")) {
        c->opt_reassoc = value != 0;
    } else if (!strcmp(name, "spmm_blocked")) {
        if (value < 0 || value > 2) return fail(CMF_EINVAL, "spmm_blocked must be 0 (never), 1 (auto) or 2 (always); set before cmf_set_data_csr");
        c->opt_spmm_blocked = (int)value;
    } else if (!strcmp(name, "spmm_split")) {
        c->opt_spmm_split = value != 0;
    } else if (!strcmp(name, "spmm_stretch")) {
        c->opt_spmm_stretch = std::max<int64_t>(0, value);
    } else if (!strcmp(name, "spmm_block_cols")) {
        c->opt_spmm_block_cols = std::max<int64_t>(0, value);
    } else if (!strcmp(name, "sparse_mode")) {
        if (value < 0 || value > 2) return fail(CMF_EINVAL, "sparse_mode must be 0 (auto), 1 (dense) or 2 (native CSR)");
        c->opt_sparse = (int)value;
    } else {
        return fail(CMF_EINVAL, "unknown option %s", name);
    }
    return CMF_OK;
}

extern "C" int cmf_sync(cmf_ctx *c) {
    if (!c) return fail(CMF_EINVAL, "null context");
    DeviceGuard dg(c->device);
    HIPCHK(hipStreamSynchronize(c->stream));
    return CMF_OK;
}

static int pad_k(int k) {
    if (k <= 32) return 32;
    if (k <= 64) return 64;
    if (k <= 128) return 128;
    return (int)rup(k, 256);
}

extern "C" int cmf_set_problem(cmf_ctx *c, int64_t m, int64_t d, int64_t p, int k) {
    if (!c) return fail(CMF_EINVAL, "null context");
    if (m < 0 || d < 0 || p < 0 || k <= 0) return fail(CMF_EINVAL, "bad problem size m=%lld d=%lld p=%lld k=%d", (long long)m, (long long)d, (long long)p, k);
    DeviceGuard dg(c->device);
    release_problem(c);
    c->m = m; c->d = d; c->p = p; c->k = k;
    c->mp = rup(std::max<int64_t>(m, 1), 256);
    c->dp = rup(std::max<int64_t>(d, 1), 256);
    c->pp = rup(std::max<int64_t>(p, 1), 256);
    c->kp = pad_k(k);
    const int64_t rows[3] = {m, d, p}, rowsp[3] = {c->mp, c->dp, c->pp};
    for (int f = 0; f < 3; ++f) {
        c->frows[f] = rows[f];
        c->frows_pad[f] = rowsp[f];
    }
    // U and Z share one allocation, Z right behind U: U^T U + Z^T Z (cmf_solvers.py:245) is then the Gram of the stacked
    // (m_pad + p_pad) x k_pad matrix -- one launch (padding rows are zero)
    CHK(dev_alloc(c, (void **)&c->F[CMF_U], (size_t)(c->mp + c->pp) * c->kp * sizeof(float)));
    c->F[CMF_Z] = c->F[CMF_U] + c->mp * c->kp;
    CHK(dev_alloc(c, (void **)&c->F[CMF_V], (size_t)c->dp * c->kp * sizeof(float)));
    c->v_rows_alloc = c->dp;
    const int64_t rmax = std::max(c->mp, std::max(c->dp, c->pp));
    CHK(dev_alloc(c, (void **)&c->num, (size_t)rmax * c->kp * sizeof(float)));
    CHK(dev_alloc(c, (void **)&c->den, (size_t)rmax * c->kp * sizeof(float)));
    const size_t kk = (size_t)c->kp * c->kp * sizeof(float);
    CHK(dev_alloc(c, (void **)&c->G, kk));
    CHK(dev_alloc(c, (void **)&c->G2, kk));
    CHK(dev_alloc(c, (void **)&c->Hm, kk));
    CHK(dev_alloc(c, (void **)&c->Hinv, kk));
    CHK(dev_alloc(c, (void **)&c->Eye, kk));
    CHK(dev_alloc(c, (void **)&c->vbuf, (size_t)c->dp * c->kp * sizeof(float) + kk));
    c->have_problem = true;
    return CMF_OK;
}

#define NEED_PROBLEM(c)                                                       \
    do {                                                                      \
        if (!(c)) return fail(CMF_EINVAL, "null context");                    \
        if (!(c)->have_problem) return fail(CMF_EINVAL, "cmf_set_problem has not been called"); \
    } while (0)

static int data_dims(cmf_ctx *c, int which, int64_t *rows, int64_t *cols, int64_t *rowsp, int64_t *colsp, float ***slot) {
    if (which == 0) { *rows = c->m; *cols = c->d; *rowsp = c->mp; *colsp = c->dp; *slot = &c->X; }
    else if (which == 1) { *rows = c->d; *cols = c->p; *rowsp = c->dp; *colsp = c->pp; *slot = &c->Y; }
    else return fail(CMF_EINVAL, "which must be 0 (X) or 1 (Y)");
    return CMF_OK;
}

static int ensure_dense(cmf_ctx *c, int which) {
    invalidate_graphs(c);
    c->dsq_valid[which] = false;
    c->bfp_valid[which][0] = c->bfp_valid[which][1] = false; // the dense image is about to be (re)written
    int64_t r, cc, rp, cp; float **slot;
    CHK(data_dims(c, which, &r, &cc, &rp, &cp, &slot));
    if (!*slot) CHK(dev_alloc(c, (void **)slot, (size_t)rp * cp * sizeof(float)));
    return CMF_OK;
}

static constexpr size_t PIN_BYTES = (size_t)64 << 20;
// The two pinned 64 MB staging buffers of a context come from a small process-wide pool: allocating and freeing 128 MB of pinned
// memory costs ~5 ms each way, which was two thirds of a whole `CMF.fit` at the reference's own benchmark shape (2000 x 150:
// 17 ms, of which 1 ms solver loop).  A context borrows a pair on its first transfer and hands it back when it is destroyed.
static std::mutex g_pin_mu;
static std::vector<void *> g_pin_pool;
static int pin_take(void **p) {
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (!g_pin_pool.empty()) { *p = g_pin_pool.back(); g_pin_pool.pop_back(); return CMF_OK; }
    }
    HIPCHK(hipHostMalloc(p, PIN_BYTES, hipHostMallocPortable));
    return CMF_OK;
}
static void pin_give(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (g_pin_pool.size() < 4) { g_pin_pool.push_back(p); return; }
    }
    (void)hipHostFree(p);
}
static int pin_buffers(cmf_ctx *c) {
    for (int b = 0; b < 2; ++b) {
        if (!c->pin[b]) CHK(pin_take(&c->pin[b]));
        if (!c->pin_ev[b]) HIPCHK(hipEventCreateWithFlags(&c->pin_ev[b], hipEventDisableTiming));
    }
    return CMF_OK;
}
// run fn(i0, i1) over [0, n) on up to `want` host threads (inline when the work is small)
template <typename F>
static void parallel_rows(int64_t n, int64_t work_per_row, int want, F &&fn) {
    int nt = (int)std::min<int64_t>(want, std::max<int64_t>(1, (n * work_per_row) >> 19)); // >= 512k elements per thread
    nt = (int)std::min<int64_t>(nt, n);
    if (nt <= 1) { fn((int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { fn(n * t / nt, n * (t + 1) / nt); });
    for (auto &x : th) x.join();
}
static int host_threads() {
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(16, std::max<unsigned>(1, hc));
}

// Host matrix (any element strides, float64 or float32) -> padded float32 device matrix.  Rows are converted and packed by
// several host threads into one of two pinned staging buffers while the previous chunk is on the wire (float32 bytes:
// half the PCIe traffic of the float64 source).
template <typename T>
static int upload_strided(cmf_ctx *c, float *dst, int64_t ld, int64_t rows, int64_t cols, const T *src, int64_t rs, int64_t cs) {
    if (rows == 0 || cols == 0) return CMF_OK;
    CHK(pin_buffers(c));
    const int64_t chunk_rows = std::max<int64_t>(1, std::min<int64_t>(rows, (int64_t)PIN_BYTES / (cols * (int64_t)sizeof(float))));
    if (cols * (int64_t)sizeof(float) > (int64_t)PIN_BYTES) return fail(CMF_EUNSUPPORTED, "row of %lld columns exceeds the staging buffer", (long long)cols);
    const int nthr = host_threads();
    int k = 0;
    for (int64_t r0 = 0; r0 < rows; r0 += chunk_rows, ++k) {
        const int b = k & 1;
        if (c->pin_busy[b]) { HIPCHK(hipEventSynchronize(c->pin_ev[b])); c->pin_busy[b] = false; }
        const int64_t nr = std::min(chunk_rows, rows - r0);
        float *stage = (float *)c->pin[b];
        parallel_rows(nr, cols, nthr, [&](int64_t i0, int64_t i1) {
            for (int64_t i = i0; i < i1; ++i) {
                const T *srow = src + (r0 + i) * rs;
                float *drow = stage + i * cols;
                if (cs == 1) for (int64_t j = 0; j < cols; ++j) drow[j] = (float)srow[j];
                else for (int64_t j = 0; j < cols; ++j) drow[j] = (float)srow[j * cs];
            }
        });
        HIPCHK(hipMemcpy2DAsync(dst + r0 * ld, ld * sizeof(float), stage, cols * sizeof(float), cols * sizeof(float), nr,
                                hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipEventRecord(c->pin_ev[b], c->stream));
        c->pin_busy[b] = true;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->pin_busy[0] = c->pin_busy[1] = false;
    return CMF_OK;
}

// padded float32 device matrix -> host matrix (any strides), through the same pinned buffers
template <typename T>
static int download_strided(cmf_ctx *c, const float *srcd, int64_t ld, int64_t rows, int64_t cols, T *dst, int64_t rs, int64_t cs) {
    if (rows == 0 || cols == 0) return CMF_OK;
    CHK(pin_buffers(c));
    if (cols * (int64_t)sizeof(float) > (int64_t)PIN_BYTES) return fail(CMF_EUNSUPPORTED, "row of %lld columns exceeds the staging buffer", (long long)cols);
    const int64_t chunk_rows = std::max<int64_t>(1, std::min<int64_t>(rows, (int64_t)PIN_BYTES / (cols * (int64_t)sizeof(float))));
    const int nthr = host_threads();
    HIPCHK(hipStreamSynchronize(c->stream));
    c->pin_busy[0] = c->pin_busy[1] = false;
    auto issue = [&](int64_t r0, int b) -> int {
        const int64_t nr = std::min(chunk_rows, rows - r0);
        HIPCHK(hipMemcpy2DAsync(c->pin[b], cols * sizeof(float), srcd + r0 * ld, ld * sizeof(float), cols * sizeof(float), nr,
                                hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(c->pin_ev[b], c->stream));
        return CMF_OK;
    };
    CHK(issue(0, 0));
    int k = 0;
    for (int64_t r0 = 0; r0 < rows; r0 += chunk_rows, ++k) {
        const int b = k & 1;
        if (r0 + chunk_rows < rows) CHK(issue(r0 + chunk_rows, b ^ 1)); // next chunk travels while this one is unpacked
        HIPCHK(hipEventSynchronize(c->pin_ev[b]));
        const int64_t nr = std::min(chunk_rows, rows - r0);
        const float *stage = (const float *)c->pin[b];
        parallel_rows(nr, cols, nthr, [&](int64_t i0, int64_t i1) {
            for (int64_t i = i0; i < i1; ++i) {
                T *drow = dst + (r0 + i) * rs;
                const float *srow = stage + i * cols;
                if (cs == 1) for (int64_t j = 0; j < cols; ++j) drow[j] = (T)srow[j];
                else for (int64_t j = 0; j < cols; ++j) drow[j * cs] = (T)srow[j];
            }
        });
    }
    return CMF_OK;
}

template <typename T>
static int set_data(cmf_ctx *c, int which, const T *ptr, int64_t rs, int64_t cs) {
    NEED_PROBLEM(c);
    if (!ptr) return fail(CMF_EINVAL, "null data pointer");
    DeviceGuard dg(c->device);
    int64_t r, cc, rp, cp; float **slot;
    CHK(data_dims(c, which, &r, &cc, &rp, &cp, &slot));
    CHK(ensure_dense(c, which));
    c->sparse[which] = false;
    HIPCHK(hipMemsetAsync(*slot, 0, (size_t)rp * cp * sizeof(float), c->stream));
    return upload_strided<T>(c, *slot, cp, r, cc, ptr, rs, cs);
}

extern "C" int cmf_set_data_f64(cmf_ctx *c, int which, const double *ptr, int64_t rs, int64_t cs) { return set_data<double>(c, which, ptr, rs, cs); }
extern "C" int cmf_set_data_f32(cmf_ctx *c, int which, const float *ptr, int64_t rs, int64_t cs) { return set_data<float>(c, which, ptr, rs, cs); }

extern "C" int cmf_set_data_csr(cmf_ctx *c, int which, const int64_t *indptr, const int32_t *indices, const double *data, int64_t nnz) {
    NEED_PROBLEM(c);
    if (!indptr || (nnz > 0 && (!indices || !data))) return fail(CMF_EINVAL, "null CSR pointer");
    DeviceGuard dg(c->device);
    int64_t r, cc, rp, cp; float **slot;
    CHK(data_dims(c, which, &r, &cc, &rp, &cp, &slot));
    // native CSR (SpMM / SDDMM kernels) when the matrix is genuinely sparse and large, or on
    // request; otherwise expand into the dense layout and use the MFMA path
    const double dense_bytes = (double)rp * (double)cp * 4.0;
    const double density = (r > 0 && cc > 0) ? (double)nnz / ((double)r * (double)cc) : 1.0;
    const bool native = c->opt_sparse == 2 || (c->opt_sparse == 0 && density < 0.02 && dense_bytes > 1e9);
    // validate the row pointer before anything reads through it
    if (indptr[0] != 0) return fail(CMF_EINVAL, "CSR indptr[0] must be 0");
    for (int64_t i = 0; i < r; ++i)
        if (indptr[i + 1] < indptr[i]) return fail(CMF_EINVAL, "CSR indptr is not monotonic at row %lld", (long long)i);
    if (indptr[r] != nnz) return fail(CMF_EINVAL, "CSR indptr[rows] = %lld does not match nnz = %lld", (long long)indptr[r], (long long)nnz);
    invalidate_graphs(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    if (*slot) { dev_free(c, *slot); *slot = nullptr; }
    c->bfp_valid[which][0] = c->bfp_valid[which][1] = false; // bf16 planes of a previous matrix are stale
    for (int t = 0; t < 2; ++t) {                              // so are its native CSR images
        CsrDev &old = c->sp[which][t];
        dev_free(c, old.indptr); dev_free(c, old.idx); dev_free(c, old.val);
        dev_free(c, old.b_ent); dev_free(c, old.b_seg); dev_free(c, old.b_grow);
        dev_free(c, old.b_vmap); dev_free(c, old.b_prow); dev_free(c, old.b_pfirst); dev_free(c, old.b_pcnt);
        old = CsrDev();
    }
    c->sparse[which] = false;
    c->sp_sq[which] = 0.0;
    if (native) return set_data_csr_native(c, which, indptr, indices, data, nnz, r, cc);
    if (dense_bytes > 200e9) return fail(CMF_EUNSUPPORTED, "CSR input too large to expand densely (%lld x %lld)", (long long)r, (long long)cc);
    CHK(ensure_dense(c, which));
    HIPCHK(hipMemsetAsync(*slot, 0, (size_t)rp * cp * sizeof(float), c->stream));
    const int64_t chunk_rows = std::max<int64_t>(1, std::min<int64_t>(r, (int64_t)(64 << 20) / (std::max<int64_t>(cc, 1) * 4)));
    std::vector<float> stage((size_t)chunk_rows * cc);
    for (int64_t r0 = 0; r0 < r; r0 += chunk_rows) {
        const int64_t nr = std::min(chunk_rows, r - r0);
        std::fill(stage.begin(), stage.begin() + nr * cc, 0.f);
        for (int64_t i = 0; i < nr; ++i)
            for (int64_t q = indptr[r0 + i]; q < indptr[r0 + i + 1]; ++q) {
                if (indices[q] < 0 || indices[q] >= cc) return fail(CMF_EINVAL, "CSR column index out of range");
                stage[i * cc + indices[q]] += (float)data[q]; // duplicates sum, as scipy does
            }
        HIPCHK(hipMemcpy2DAsync(*slot + r0 * cp, cp * sizeof(float), stage.data(), cc * sizeof(float), cc * sizeof(float), nr,
                                hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return CMF_OK;
}

extern "C" int cmf_get_data_f32(cmf_ctx *c, int which, float *ptr, int64_t rs, int64_t cs) {
    NEED_PROBLEM(c);
    DeviceGuard dg(c->device);
    int64_t r, cc, rp, cp; float **slot;
    CHK(data_dims(c, which, &r, &cc, &rp, &cp, &slot));
    if (!*slot && c->sparse[which]) CHK(need_dense(c, which));
    if (!*slot) return fail(CMF_EINVAL, "data %d not set", which);
    return download_strided<float>(c, *slot, cp, r, cc, ptr, rs, cs);
}

// which image of X / Y lives on the device: *dense = 1 when a dense float32 image exists (dense input, small sparse input, or a
// sparse input that some sweep had to expand), *native = 1 when the CSR pair (A, A^T) is resident
extern "C" int cmf_data_layout(cmf_ctx *c, int which, int *dense, int *native) {
    NEED_PROBLEM(c);
    if (which != 0 && which != 1) return fail(CMF_EINVAL, "which must be 0 (X) or 1 (Y)");
    if (dense) *dense = (which == 0 ? c->X : c->Y) != nullptr ? 1 : 0;
    if (native) *native = c->sparse[which] ? 1 : 0;
    return CMF_OK;
}

// layout of the column-blocked SpMM images of a native sparse X / Y: out[0..3] = (row groups, split rows, pieces, accumulator rows per
// group) of A, out[4..7] the same of A^T; zeros where an orientation has no blocked image
extern "C" int cmf_sparse_layout(cmf_ctx *c, int which, int64_t *out) {
    NEED_PROBLEM(c);
    if ((which != 0 && which != 1) || !out) return fail(CMF_EINVAL, "bad argument");
    for (int t = 0; t < 2; ++t) {
        const CsrDev &A = c->sp[which][t];
        out[4 * t] = A.b_ent ? A.b_ngroups : 0;
        out[4 * t + 1] = A.b_ent ? A.b_nsplit : 0;
        out[4 * t + 2] = A.b_ent ? A.b_nslots : 0;
        out[4 * t + 3] = A.b_ent ? A.b_rows_per_group : 0;
    }
    return CMF_OK;
}

// a rows x cols block of the DENSE device image of X / Y into a packed host array (parity tests at sizes whose whole matrix
// does not fit the host)
extern "C" int cmf_get_data_block_f32(cmf_ctx *c, int which, int64_t row0, int64_t nrows, int64_t col0, int64_t ncols, float *dst) {
    NEED_PROBLEM(c);
    DeviceGuard dg(c->device);
    int64_t r, cc, rp, cp; float **slot;
    CHK(data_dims(c, which, &r, &cc, &rp, &cp, &slot));
    if (!dst || row0 < 0 || col0 < 0 || nrows < 0 || ncols < 0 || row0 + nrows > r || col0 + ncols > cc) return fail(CMF_EINVAL, "block out of range");
    if (!*slot) return fail(CMF_EINVAL, "%s has no dense device image (native sparse input or not set)", which == 0 ? "X" : "Y");
    if (nrows == 0 || ncols == 0) return CMF_OK;
    HIPCHK(hipMemcpy2DAsync(dst, (size_t)ncols * sizeof(float), *slot + row0 * cp + col0, (size_t)cp * sizeof(float), (size_t)ncols * sizeof(float),
                            (size_t)nrows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return CMF_OK;
}

static int launch_fill(cmf_ctx *c, float *A, int64_t ld, int64_t rows, int64_t cols, uint64_t seed, int64_t row0, int64_t col0, float scale, int kind = 0) {
    Timed tm(c, CMF_K_ELEMWISE);
    const int64_t total = rows * ((cols + 3) / 4);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
    if (total == 0) return CMF_OK;
    hipLaunchKernelGGL(fill_absnormal_kernel, dim3(blocks), dim3(256), 0, c->stream, A, ld, rows, cols, seed, row0, col0, scale, kind);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

extern "C" int cmf_fill_data_synthetic(cmf_ctx *c, int which, uint64_t seed, int64_t row0, int64_t col0) {
    NEED_PROBLEM(c);
    DeviceGuard dg(c->device);
    int64_t r, cc, rp, cp; float **slot;
    CHK(data_dims(c, which, &r, &cc, &rp, &cp, &slot));
    CHK(ensure_dense(c, which));
    HIPCHK(hipMemsetAsync(*slot, 0, (size_t)rp * cp * sizeof(float), c->stream));
    return launch_fill(c, *slot, cp, r, cc, seed, row0, col0, 1.0f);
}

extern "C" int cmf_fill_data_synthetic_kind(cmf_ctx *c, int which, uint64_t seed, int64_t row0, int64_t col0, int kind, double param) {
    NEED_PROBLEM(c);
    if (kind < 0 || kind > 2) return fail(CMF_EINVAL, "synthetic data kind: 0 |N(0,1)|, 1 sigmoid(N(0,1)), 2 Bernoulli(param)");
    DeviceGuard dg(c->device);
    int64_t r, cc, rp, cp; float **slot;
    CHK(data_dims(c, which, &r, &cc, &rp, &cp, &slot));
    CHK(ensure_dense(c, which));
    HIPCHK(hipMemsetAsync(*slot, 0, (size_t)rp * cp * sizeof(float), c->stream));
    return launch_fill(c, *slot, cp, r, cc, seed, row0, col0, kind == 0 ? 1.0f : (float)param, kind);
}

extern "C" int cmf_fill_factor_synthetic(cmf_ctx *c, int which, uint64_t seed, int64_t row0, double scale) {
    NEED_PROBLEM(c);
    if (which < 0 || which > 2) return fail(CMF_EINVAL, "bad factor id");
    DeviceGuard dg(c->device);
    HIPCHK(hipMemsetAsync(c->F[which], 0, (size_t)c->frows_pad[which] * c->kp * sizeof(float), c->stream));
    return launch_fill(c, c->F[which], c->kp, c->frows[which], c->k, seed, row0, 0, (float)scale);
}

extern "C" int cmf_set_factor_f64(cmf_ctx *c, int which, const double *ptr, int64_t rs, int64_t cs) {
    NEED_PROBLEM(c);
    if (which < 0 || which > 2 || !ptr) return fail(CMF_EINVAL, "bad factor argument");
    DeviceGuard dg(c->device);
    HIPCHK(hipMemsetAsync(c->F[which], 0, (size_t)c->frows_pad[which] * c->kp * sizeof(float), c->stream));
    return upload_strided<double>(c, c->F[which], c->kp, c->frows[which], c->k, ptr, rs, cs);
}

extern "C" int cmf_get_factor_f64(cmf_ctx *c, int which, double *ptr, int64_t rs, int64_t cs) {
    NEED_PROBLEM(c);
    if (which < 0 || which > 2 || !ptr) return fail(CMF_EINVAL, "bad factor argument");
    DeviceGuard dg(c->device);
    return download_strided<double>(c, c->F[which], c->kp, c->frows[which], c->k, ptr, rs, cs);
}

extern "C" int cmf_get_geometry(cmf_ctx *c, int64_t *mp, int64_t *dp, int64_t *pp, int *kp) {
    NEED_PROBLEM(c);
    if (mp) *mp = c->mp;
    if (dp) *dp = c->dp;
    if (pp) *pp = c->pp;
    if (kp) *kp = c->kp;
    return CMF_OK;
}

extern "C" int cmf_factor_dev_ptr(cmf_ctx *c, int which, float **ptr) {
    NEED_PROBLEM(c);
    if (which < 0 || which > 2 || !ptr) return fail(CMF_EINVAL, "bad factor argument");
    *ptr = c->F[which];
    return CMF_OK;
}

// zero-filled device scratch owned by the context (the partial / staging buffers of the sharded drivers when the caller
// has no allocator of its own at hand); released by cmf_scratch_free or cmf_ctx_destroy -- it SURVIVES cmf_set_problem
extern "C" int cmf_scratch_alloc(cmf_ctx *c, int64_t bytes, void **dev_ptr) {
    if (!c || !dev_ptr || bytes < 0) return fail(CMF_EINVAL, "bad scratch request");
    DeviceGuard dg(c->device);
    const size_t n = bytes ? (size_t)bytes : 16;
    HIPCHK(hipMalloc(dev_ptr, n));
    c->scratch.push_back(*dev_ptr);
    HIPCHK(hipMemsetAsync(*dev_ptr, 0, n, c->stream));
    return CMF_OK;
}

extern "C" int cmf_scratch_free(cmf_ctx *c, void *dev_ptr) {
    if (!c) return fail(CMF_EINVAL, "null context");
    if (!dev_ptr) return CMF_OK;
    DeviceGuard dg(c->device);
    auto it = std::find(c->scratch.begin(), c->scratch.end(), dev_ptr);
    if (it == c->scratch.end()) return fail(CMF_EINVAL, "not a scratch buffer of this context");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->scratch.erase(it);
    (void)hipFree(dev_ptr);
    return CMF_OK;
}

// device-to-device exchange of factor rows (fp32, k_pad floats per row): what a sharded driver all-gathers
extern "C" int cmf_export_factor_rows(cmf_ctx *c, int which, float *dev_dst) {
    NEED_PROBLEM(c);
    if (which < 0 || which > 2 || !dev_dst) return fail(CMF_EINVAL, "bad factor argument");
    DeviceGuard dg(c->device);
    HIPCHK(hipMemcpyAsync(dev_dst, c->F[which], (size_t)c->frows[which] * c->kp * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return CMF_OK;
}

extern "C" int cmf_import_factor_rows(cmf_ctx *c, int which, const float *dev_src) {
    NEED_PROBLEM(c);
    if (which < 0 || which > 2 || !dev_src) return fail(CMF_EINVAL, "bad factor argument");
    DeviceGuard dg(c->device);
    HIPCHK(hipMemcpyAsync(c->F[which], dev_src, (size_t)c->frows[which] * c->kp * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return CMF_OK;
}

extern "C" int cmf_v_buf_elems(cmf_ctx *c, int64_t *n) {
    NEED_PROBLEM(c);
    if (!n) return fail(CMF_EINVAL, "null argument");
    *n = c->dp * c->kp + (int64_t)c->kp * c->kp;
    return CMF_OK;
}

// ------------------------------------------------------------------ MU solver
// V numerator/Gram partials of this shard: cmf_solvers.py:244-245
extern "C" int cmf_mu_v_partials(cmf_ctx *c, float *buf) {
    NEED_PROBLEM(c);
    if (!buf) return fail(CMF_EINVAL, "null buffer");
    if (!have_data(c, 0) || !have_data(c, 1)) return fail(CMF_EINVAL, "X and Y must be set before a V update");
    DeviceGuard dg(c->device);
    float *P = buf, *Gs = buf + c->dp * c->kp;
    // P = X^T U + Y Z
    CHK(data_times(c, 0, true, c->F[CMF_U], P));
    CHK(data_times(c, 1, false, c->F[CMF_Z], P, true));
    // G = U^T U + Z^T Z: Gram of the stacked [U; Z]
    CHK(gram32(c, c->F[CMF_U], c->mp + c->pp, Gs));
    return CMF_OK;
}

// The same partial for a block of V rows only: P[row0 .. row0 + nrows) = X[:, rows]^T U + Y[rows, :] Z (dense X and Y), plus the
// Gram part when with_gram.  A sharded driver computes block c + 1 while block c is being all-reduced on the communicator's side
// stream (cmf_comm_allreduce_f32_bg): the only way to overlap the one collective of an MU iteration with compute, since everything
// after it needs the new V (pycmf/cmf_solvers.py:248-263).
extern "C" int cmf_mu_v_partials_rows(cmf_ctx *c, float *buf, int64_t row0, int64_t nrows, int with_gram) {
    NEED_PROBLEM(c);
    if (!buf) return fail(CMF_EINVAL, "null buffer");
    if (!c->X || !c->Y) return fail(CMF_EUNSUPPORTED, "cmf_mu_v_partials_rows needs dense images of X and Y");
    if (c->opt_arith != 0) return fail(CMF_EUNSUPPORTED, "cmf_mu_v_partials_rows: fp32 MFMA arithmetic only");
    if (row0 < 0 || nrows <= 0 || row0 % 256 || nrows % 256 || row0 + nrows > c->dp) return fail(CMF_EINVAL, "row block must be 256-aligned inside [0, d_pad)");
    DeviceGuard dg(c->device);
    float *P = buf + row0 * c->kp;
    CHK(gemm(c, MODE_TN, c->X + row0, c->dp, c->F[CMF_U], c->kp, P, nrows, c->kp, c->mp));                       // X[:, rows]^T U
    CHK(gemm(c, MODE_NN, c->Y + row0 * c->pp, c->pp, c->F[CMF_Z], c->kp, P, nrows, c->kp, c->pp, true));        // + Y[rows, :] Z
    if (with_gram)
        CHK(gram32(c, c->F[CMF_U], c->mp + c->pp, buf + c->dp * c->kp));
    return CMF_OK;
}

// V *= P / reg(V G): cmf_solvers.py:245, :212-228, :253-255
extern "C" int cmf_mu_v_apply(cmf_ctx *c, const float *buf, double l1, double l2) {
    NEED_PROBLEM(c);
    if (!buf) return fail(CMF_EINVAL, "null buffer");
    DeviceGuard dg(c->device);
    const float *P = buf, *Gs = buf + c->dp * c->kp;
    return mu_update(c, c->F[CMF_V], Gs, P, c->dp, l1, l2);
}

// ---- paired data passes (cmf_gemm_pair.hip.h): product i = X or Y (which), transposed or not, times the factor B ----
struct PairSide { int which; bool trans; const float *B; };
static void pair_shape(const cmf_ctx *c, const PairSide &sd, int64_t *mout, int64_t *kred) {
    const int64_t rp = sd.which == 0 ? c->mp : c->dp, cp = sd.which == 0 ? c->dp : c->pp;
    *mout = sd.trans ? cp : rp; *kred = sd.trans ? rp : cp;
}
// worth it when BOTH products would otherwise split their reduction (few output tiles against the CUs), and only on the plain
// fp32 path the kernel restates (k_pad = 128, staging schedule 4, dense inputs, slabs summed by factor_update_kernel)
static bool pair_ok(const cmf_ctx *c, const PairSide &a0, const PairSide &a1) {
    if (!c->opt_pair || c->kp != 128 || !c->X || !c->Y || c->opt_arith || c->opt_pipe != 4 || c->opt_tile512 || c->opt_split > 0 ||
        c->opt_inred || !c->opt_fused_mu || !c->opt_small_tile)
        return false;
    for (const PairSide *sd : {&a0, &a1}) {
        int64_t mout, kred;
        pair_shape(c, *sd, &mout, &kred);
        if (mout / 256 >= (int64_t)(c->num_cu * 3) / 4 || kred < 256 || !small_tile_ok(c, mout)) return false;
    }
    return true;
}
static int data_times_pair(cmf_ctx *c, const PairSide &a0, const PairSide &a1, SlabRef *s0, SlabRef *s1) {
    PairArgs g;
    memset(&g, 0, sizeof g);
    const PairSide *sides[2] = {&a0, &a1};
    SlabRef *refs[2] = {s0, s1};
    DevBuf *bufs[2] = {&c->slabs, &c->slabs_b};
    int64_t unit = 0, mout[2];
    double flops = 0.0;
    for (int i = 0; i < 2; ++i) {
        int64_t kred;
        pair_shape(c, *sides[i], &mout[i], &kred);
        PairProb &p = g.p[i];
        p.A = sides[i]->which == 0 ? c->X : c->Y; p.lda = sides[i]->which == 0 ? c->dp : c->pp;
        p.B = sides[i]->B; p.ldb = c->kp;
        p.mode = sides[i]->trans ? MODE_TN : MODE_NN;
        p.tiles = (int)(mout[i] / 256); p.ksteps = (int)(kred / 32);
        p.unit0 = unit; p.slab_stride = mout[i] * c->kp;
        unit += (int64_t)p.tiles * p.ksteps;
        flops += 2.0 * (double)mout[i] * (double)c->kp * (double)kred;
    }
    g.total = unit;
    g.quota = (unit + c->num_cu - 1) / c->num_cu;
    const int64_t nwg = (unit + g.quota - 1) / g.quota;
    for (int i = 0; i < 2; ++i) {
        PairProb &p = g.p[i];
        const int64_t maxslots = (p.ksteps + g.quota - 1) / g.quota + 1;
        CHK(ensure(c, *bufs[i], (size_t)maxslots * p.slab_stride * sizeof(float)));
        p.C = (float *)bufs[i]->p;
        refs[i]->base = p.C; refs[i]->nslab = (int)maxslots; refs[i]->stride = p.slab_stride;
        refs[i]->quota = g.quota; refs[i]->unit0 = p.unit0; refs[i]->ksteps = p.ksteps;
    }
    constexpr size_t lds = GemmCfg<MODE_NN, 128>::LDS_BYTES > GemmCfg<MODE_TN, 128>::LDS_BYTES ? GemmCfg<MODE_NN, 128>::LDS_BYTES : GemmCfg<MODE_TN, 128>::LDS_BYTES;
    CHK(allow_big_lds(c, reinterpret_cast<const void *>(&gemm_pair_kernel), (int)lds));
    Timed tm(c, CMF_K_GEMM_PAIR, flops * c->flop_scale);
    hipLaunchKernelGGL(gemm_pair_kernel, dim3((unsigned)nwg), dim3(512), lds, c->stream, g);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

static int mu_uz_update_with(cmf_ctx *c, const float *G2, double l1, double l2, int mask);
// U *= X V / reg(U V^T V), Z *= Y^T V / reg(Z V^T V): cmf_solvers.py:230-240, :257-263.
// (U V^T) V is evaluated as U (V^T V): same value, 2mk^2 instead of 4mdk flops.
extern "C" int cmf_mu_uz_update(cmf_ctx *c, double l1, double l2, int mask) {
    NEED_PROBLEM(c);
    DeviceGuard dg(c->device);
    if (!(mask & (CMF_UPD_U | CMF_UPD_Z))) return CMF_OK;
    CHK(gram32(c, c->F[CMF_V], c->dp, c->G2, true));   // V^T V beside the first data pass of the U / Z updates
    return mu_uz_update_with(c, c->G2, l1, l2, mask);
}

// ---- row-blocked V update: the ONE all-reduce of the MU iteration cut in two around the epilogue ("Partitioning") ----
// With N ranks, block r of V = rows [r B, (r + 1) B), B = 256 ceil(d_pad / 256 / N).  Every rank forms the whole partial
// P = X_g^T U_g + Y_g Z_g (cmf_mu_v_partials_split), a reduce-scatter leaves rank r with the SUM of block r, rank r alone applies
// V_r *= P_r / reg(V_r G) to its B rows (cmf_mu_v_apply_rows) and forms its share of V^T V (cmf_mu_gram_v_rows), an all-gather
// reassembles V in place.  Against the replicated epilogue this removes (N - 1) / N of the 2 d k^2 product V G and of the d-row
// Gram V^T V from every rank -- 0.3 of the 0.5 ms a C4 rank spends outside its four data passes at N = 8.  The two k x k Grams
// (U^T U + Z^T Z before the epilogue, V^T V after it) are summed by two latency-bound all-reduces of k_pad^2 floats.
// cmf_solvers.py:242-246 (V), :230-240 (U, Z): sums over rows are associative, so the iterates equal the unsharded ones up to the
// order of the float32 additions across ranks.
static int grow_v(cmf_ctx *c, int64_t rows) {
    if (rows <= c->v_rows_alloc) return CMF_OK;
    invalidate_graphs(c);
    float *nv = nullptr;
    CHK(dev_alloc(c, (void **)&nv, (size_t)rows * c->kp * sizeof(float))); // zero-filled (dev_alloc): rows >= d_pad stay zero
    HIPCHK(hipMemcpyAsync(nv, c->F[CMF_V], (size_t)c->dp * c->kp * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    dev_free(c, c->F[CMF_V]);
    c->F[CMF_V] = nv;
    c->v_rows_alloc = rows;
    return CMF_OK;
}
extern "C" int cmf_mu_blocked_layout(cmf_ctx *c, int world, int64_t *block_rows, int64_t *pbuf_elems) {
    NEED_PROBLEM(c);
    if (world < 1) return fail(CMF_EINVAL, "world must be >= 1");
    DeviceGuard dg(c->device);
    const int64_t tiles = c->dp / 256;
    const int64_t B = 256 * ((tiles + world - 1) / world);
    CHK(grow_v(c, B * world)); // the in-place all-gather writes world equal blocks behind F[V]; rows >= d_pad stay zero
    if (block_rows) *block_rows = B;
    if (pbuf_elems) *pbuf_elems = B * world * c->kp;
    return CMF_OK;
}
// P (d_pad x k_pad, written; the caller's buffer may be longer: rows >= d_pad are not touched) and G (k_pad x k_pad) of this shard
extern "C" int cmf_mu_v_partials_split(cmf_ctx *c, float *P, float *G) {
    NEED_PROBLEM(c);
    if (!P || !G) return fail(CMF_EINVAL, "null buffer");
    if (!have_data(c, 0) || !have_data(c, 1)) return fail(CMF_EINVAL, "X and Y must be set before a V update");
    DeviceGuard dg(c->device);
    CHK(gram32(c, c->F[CMF_U], c->mp + c->pp, G));
    CHK(data_times(c, 0, true, c->F[CMF_U], P));
    CHK(data_times(c, 1, false, c->F[CMF_Z], P, true));
    return CMF_OK;
}
// V[row0 .. row0 + nrows) *= P_rows / reg(V_rows G): P_rows points at the (summed) rows of the partial that belong to these V rows
extern "C" int cmf_mu_v_apply_rows(cmf_ctx *c, const float *P_rows, const float *G, int64_t row0, int64_t nrows, double l1, double l2) {
    NEED_PROBLEM(c);
    if (!P_rows || !G) return fail(CMF_EINVAL, "null buffer");
    if (row0 < 0 || nrows < 0 || row0 % 256 || nrows % 256 || row0 + nrows > c->dp) return fail(CMF_EINVAL, "row block must be 256-aligned inside [0, d_pad)");
    if (nrows == 0) return CMF_OK;
    DeviceGuard dg(c->device);
    return mu_update(c, c->F[CMF_V] + row0 * c->kp, G, P_rows, nrows, l1, l2);
}
// G2 = V_rows^T V_rows of a 256-aligned block of V rows (zero for an empty block): this rank's share of V^T V
extern "C" int cmf_mu_gram_v_rows(cmf_ctx *c, int64_t row0, int64_t nrows, float *G2) {
    NEED_PROBLEM(c);
    if (!G2) return fail(CMF_EINVAL, "null buffer");
    if (row0 < 0 || nrows < 0 || row0 % 256 || nrows % 256 || row0 + nrows > c->dp) return fail(CMF_EINVAL, "row block must be 256-aligned inside [0, d_pad)");
    DeviceGuard dg(c->device);
    if (nrows == 0) {
        HIPCHK(hipMemsetAsync(G2, 0, (size_t)c->kp * c->kp * sizeof(float), c->stream));
        return CMF_OK;
    }
    return gram32(c, c->F[CMF_V] + row0 * c->kp, nrows, G2);
}
static int mu_uz_update_with(cmf_ctx *c, const float *G2, double l1, double l2, int mask) {
    const PairSide xv{0, false, c->F[CMF_V]}, ytv{1, true, c->F[CMF_V]};
    // c->mu_dots (cmf_mu_step_error): the numerators X V and Y^T V are wanted whole in c->num behind the updates -- no pair launch,
    // no partial tiles left to the update kernel
    auto factor_dot = [&](const float *F, int64_t n, double *out) -> int {
        const int blocks = 1024;
        CHK(ensure(c, c->dpart, blocks * sizeof(double)));
        Timed tm(c, CMF_K_ELEMWISE);
        hipLaunchKernelGGL(dot_kernel, dim3(blocks), dim3(256), 0, c->stream, F, (const float *)c->num, n / 4, (double *)c->dpart.p);
        hipLaunchKernelGGL(sum_doubles_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)c->dpart.p, (int64_t)blocks, out);
        HIPCHK(hipGetLastError());
        return CMF_OK;
    };
    if (!c->mu_dots && (mask & CMF_UPD_U) && (mask & CMF_UPD_Z) && pair_ok(c, xv, ytv)) {
        // X V and Y^T V (cmf_solvers.py:232, :238) read the same V: one balanced launch, then the two updates
        SlabRef su, sz;
        CHK(data_times_pair(c, xv, ytv, &su, &sz));
        CHK(side_join(c));
        if (c->opt_pair == 2) { // A/B: the two updates as two launches
            CHK(mu_update(c, c->F[CMF_U], G2, c->num, c->mp, l1, l2, &su));
            return mu_update(c, c->F[CMF_Z], G2, c->num, c->pp, l1, l2, &sz);
        }
        Epilogue eu, ez;
        eu.kind = EPI_MU; eu.F = c->F[CMF_U]; eu.out = c->F[CMF_U]; eu.a = l1; eu.b = l2; eu.c = 1.1920928955078125e-07; eu.Pslabs = &su;
        ez = eu; ez.F = c->F[CMF_Z]; ez.out = c->F[CMF_Z]; ez.Pslabs = &sz;
        return factor_update2(c, c->F[CMF_U], eu, c->mp, c->F[CMF_Z], ez, c->pp, G2);
    }
    if (mask & CMF_UPD_U) {
        if (!have_data(c, 0)) { (void)side_join(c); return fail(CMF_EINVAL, "X must be set before a U update"); }
        SlabRef sl;
        CHK(data_times(c, 0, false, c->F[CMF_V], c->num, false, (small_tile_ok(c, c->mp) && !c->mu_dots) ? &sl : nullptr));
        CHK(side_join(c)); // G2 (when it was formed on the side stream)
        CHK(mu_update(c, c->F[CMF_U], G2, c->num, c->mp, l1, l2, &sl));
        if (c->mu_dots) CHK(factor_dot(c->F[CMF_U], c->mp * c->kp, c->dscalar + 2));
    }
    if (mask & CMF_UPD_Z) {
        if (!have_data(c, 1)) { (void)side_join(c); return fail(CMF_EINVAL, "Y must be set before a Z update"); }
        SlabRef sl;
        CHK(data_times(c, 1, true, c->F[CMF_V], c->num, false, (small_tile_ok(c, c->pp) && !c->mu_dots) ? &sl : nullptr));
        CHK(side_join(c));
        CHK(mu_update(c, c->F[CMF_Z], G2, c->num, c->pp, l1, l2, &sl));
        if (c->mu_dots) CHK(factor_dot(c->F[CMF_Z], c->pp * c->kp, c->dscalar + 3));
    }
    return side_join(c);
}
// cmf_mu_uz_update with V^T V supplied by the caller (the all-reduced sum of the ranks' cmf_mu_gram_v_rows)
extern "C" int cmf_mu_uz_update_gram(cmf_ctx *c, const float *G2, double l1, double l2, int mask) {
    NEED_PROBLEM(c);
    if (!G2) return fail(CMF_EINVAL, "null buffer");
    DeviceGuard dg(c->device);
    if (!(mask & (CMF_UPD_U | CMF_UPD_Z))) return CMF_OK;
    return mu_uz_update_with(c, G2, l1, l2, mask);
}

// Run `eager` (a fixed sequence of launches on c->stream) directly, or capture it into a hipGraph
// on its second call with an unchanged key and replay the graph from then on.  Launch-bound small
// problems (the reference's own test sizes run ~25 launches of a few microseconds per iteration)
// are then paced by one graph launch instead.
template <typename F>
static int run_graphed(cmf_ctx *c, StepGraph &g, const double *key, int nkey, F &&eager) {
    if (c->opt_graph <= 0 || c->timing) return eager();
    bool same = true;
    for (int i = 0; i < nkey; ++i) same = same && (g.key[i] == key[i]);
    if (!same) {
        drop_graph(g);
        for (int i = 0; i < nkey; ++i) g.key[i] = key[i];
    }
    if (g.exec) {
        HIPCHK(hipGraphLaunch(g.exec, c->stream));
        return CMF_OK;
    }
    if (g.failed || g.warm < 1) {
        g.warm += 1;
        return eager();
    }
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        g.failed = true;
        (void)hipGetLastError();
        return eager();
    }
    const int rc = eager();
    const hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc != CMF_OK || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        g.failed = true;
        // capture swallowed the launches (or hit a host synchronisation): run the step for real
        return eager();
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess || !exec) {
        (void)hipGetLastError();
        g.failed = true;
        return eager();
    }
    // ensure() may have dropped the graph during capture (it must not allocate then); guard anyway
    g.exec = exec;
    HIPCHK(hipGraphLaunch(g.exec, c->stream));
    return CMF_OK;
}

// V update of the single-GPU step with both numerator products left as slabs for the update kernel (no slab-sum launches,
// no round trip of P through HBM): cmf_solvers.py:242-246, :253-255
static int mu_v_fused(cmf_ctx *c, double l1, double l2) {
    float *P = c->vbuf, *Gs = c->vbuf + c->dp * c->kp;
    SlabRef s1, s2;
    // the Gram first: its own split goes through slab set 0, which the deferred products below must own until the update
    CHK(gram32(c, c->F[CMF_U], c->mp + c->pp, Gs, true));             // beside the data passes below; joined in front of the update
    const PairSide xtu{0, true, c->F[CMF_U]}, yz{1, false, c->F[CMF_Z]};
    if (pair_ok(c, xtu, yz)) { // P = X^T U + Y Z (cmf_solvers.py:244) as one balanced launch, its partial tiles summed by the update
        CHK(data_times_pair(c, xtu, yz, &s1, &s2));
        CHK(side_join(c));
        Epilogue e;
        e.kind = EPI_MU; e.F = c->F[CMF_V]; e.out = c->F[CMF_V]; e.a = l1; e.b = l2; e.c = 1.1920928955078125e-07;
        e.P = nullptr; e.Pslabs = &s1; e.Pslabs2 = &s2;
        return factor_update(c, c->F[CMF_V], Gs, e, c->dp);
    }
    CHK(data_times(c, 0, true, c->F[CMF_U], P, false, &s1));           // X^T U: slabs (set 0) or, unsplit, P itself
    const float *direct = s1.nslab > 0 ? nullptr : P;
    if (s1.nslab > 0) {
        c->slab_sel = 1;                                                // Y Z must not overwrite the slabs of X^T U
        const int rc = data_times(c, 1, false, c->F[CMF_Z], c->num, false, &s2);
        c->slab_sel = 0;
        CHK(rc);
        if (s2.nslab == 0) direct = c->num;                            // unsplit: its value sits in c->num
    } else {
        CHK(data_times(c, 1, false, c->F[CMF_Z], P, true));            // accumulate onto P as usual
    }
    CHK(side_join(c));
    Epilogue e;
    e.kind = EPI_MU; e.F = c->F[CMF_V]; e.out = c->F[CMF_V]; e.a = l1; e.b = l2; e.c = 1.1920928955078125e-07;
    e.P = direct; e.Pslabs = &s1; e.Pslabs2 = &s2;
    return factor_update(c, c->F[CMF_V], Gs, e, c->dp);
}

static int mu_step_eager(cmf_ctx *c, double l1, double l2, int mask) {
    if ((mask & CMF_UPD_V) && small_tile_ok(c, c->dp) && c->X && c->Y && !c->opt_arith) {
        CHK(mu_v_fused(c, l1, l2));
        return cmf_mu_uz_update(c, l1, l2, mask);
    }
    if (mask & CMF_UPD_V) {
        CHK(cmf_mu_v_partials(c, c->vbuf));
        CHK(cmf_mu_v_apply(c, c->vbuf, l1, l2));
    }
    return cmf_mu_uz_update(c, l1, l2, mask);
}

extern "C" int cmf_mu_step(cmf_ctx *c, double l1, double l2, int mask) {
    NEED_PROBLEM(c);
    DeviceGuard dg(c->device);
    const double key[3] = {l1, l2, (double)mask};
    return run_graphed(c, c->mu_graph, key, 3, [&]() { return mu_step_eager(c, l1, l2, mask); });
}

// ------------------------------------------------------------------ error metric
extern "C" int cmf_residual_sq(cmf_ctx *c, int x_link, int y_link, double *ex2, double *ey2) {
    NEED_PROBLEM(c);
    DeviceGuard dg(c->device);
    double host[2] = {0, 0};
    HIPCHK(hipMemsetAsync(c->dscalar, 0, 2 * sizeof(double), c->stream));
    if (ex2 && have_data(c, 0)) {
        if (!c->X && c->sparse[0]) {
            CHK(sparse_residual_sq(c, 0, c->dscalar, x_link));
        } else {
            CHK(need_dense(c, 0));
            NtOut o; o.T = c->opt_nt_debug & 1 ? nullptr : c->X; o.ldt = c->dp; o.sq = c->dscalar; o.link = x_link;
            CHK(gemm_nt(c, c->F[CMF_U], c->mp, c->m, c->F[CMF_V], c->dp, c->d, o));
        }
    }
    if (ey2 && have_data(c, 1)) {
        if (!c->Y && c->sparse[1]) {
            CHK(sparse_residual_sq(c, 1, c->dscalar + 1, y_link));
        } else {
            CHK(need_dense(c, 1));
            NtOut o; o.T = c->opt_nt_debug & 1 ? nullptr : c->Y; o.ldt = c->pp; o.sq = c->dscalar + 1; o.link = y_link;
            CHK(gemm_nt(c, c->F[CMF_V], c->dp, c->d, c->F[CMF_Z], c->pp, c->p, o));
        }
    }
    HIPCHK(hipMemcpyAsync(host, c->dscalar, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (ex2) *ex2 = host[0];
    if (ey2) *ey2 = host[1];
    return CMF_OK;
}

extern "C" int cmf_data_sq(cmf_ctx *c, double *x2, double *y2) {
    NEED_PROBLEM(c);
    DeviceGuard dg(c->device);
    double host[2] = {0, 0};
    HIPCHK(hipMemsetAsync(c->dscalar, 0, 2 * sizeof(double), c->stream));
    const float *src[2] = {c->X, c->Y};
    const int64_t n[2] = {c->mp * c->dp, c->dp * c->pp};
    double sparse_sq[2] = {-1.0, -1.0};
    for (int w = 0; w < 2; ++w) {
        if (!src[w]) {
            if (c->sparse[w]) sparse_sq[w] = c->sp_sq[w];
            continue;
        }
        const int blocks = 1024;
        CHK(ensure(c, c->dpart, blocks * sizeof(double)));
        Timed tm(c, CMF_K_ELEMWISE);
        hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, c->stream, src[w], n[w] / 4, (double *)c->dpart.p);
        hipLaunchKernelGGL(sum_doubles_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)c->dpart.p, (int64_t)blocks, c->dscalar + w);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(host, c->dscalar, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (sparse_sq[0] >= 0) host[0] = sparse_sq[0];
    if (sparse_sq[1] >= 0) host[1] = sparse_sq[1];
    if (x2) *x2 = host[0];
    if (y2) *y2 = host[1];
    return CMF_OK;
}

// ------------------------------------------------------------------ timing API
// Diagnostic: launch the NN data pass X*V once with clock stamps; returns the median over workgroups of
// (shader cycles / elapsed time) in GHz and of the main-loop duration in microseconds.
extern "C" int cmf_debug_clock(cmf_ctx *c, double *ghz, double *loop_us) {
#ifndef CMF_DIAG_BUILD
    (void)c; (void)ghz; (void)loop_us;
    return fail(CMF_EUNSUPPORTED, "cmf_debug_clock: diagnostic builds only (python -m pycmf_amd.build --diag)");
#else
    NEED_PROBLEM(c);
    if (!c->X) return fail(CMF_EINVAL, "dense X required");
    DeviceGuard dg(c->device);
    const int64_t nwg = (c->mp / 256) * std::max(1, c->kp / 256) * 64;
    unsigned long long *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, (size_t)nwg * 4 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(d, 0, (size_t)nwg * 4 * sizeof(unsigned long long), c->stream));
    c->dbg_stamps = d;
    int rc = gemm(c, MODE_NN, c->X, c->dp, c->F[CMF_V], c->kp, c->num, c->mp, c->kp, c->dp);
    c->dbg_stamps = nullptr;
    (void)hipStreamSynchronize(c->stream);
    std::vector<unsigned long long> h((size_t)nwg * 4);
    if (rc == CMF_OK && hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(CMF_EHIP, "D2H failed");
    (void)hipFree(d);
    if (rc != CMF_OK) return rc;
    std::vector<double> f, us;
    for (int64_t i = 0; i < nwg; ++i) {
        const double cyc = (double)(h[4 * i + 2] - h[4 * i + 0]), rt = (double)(h[4 * i + 3] - h[4 * i + 1]);
        if (h[4 * i + 3] == 0 || rt <= 0) continue;
        f.push_back(cyc / (rt * 10.0) ); // memrealtime ticks at 100 MHz: cycles / (ticks * 10 ns) = GHz
        us.push_back(rt / 100.0);
    }
    if (f.empty()) return fail(CMF_EHIP, "no stamps recorded");
    std::sort(f.begin(), f.end()); std::sort(us.begin(), us.end());
    if (ghz) *ghz = f[f.size() / 2];
    if (loop_us) *loop_us = us[us.size() / 2];
    return CMF_OK;
#endif
}

extern "C" int cmf_marker(cmf_ctx *c) {
    if (!c) return fail(CMF_EINVAL, "null context");
    DeviceGuard dg(c->device);
    hipEvent_t e;
    CHK(ev_get(c, &e));
    HIPCHK(hipEventRecord(e, c->stream));
    c->markers.push_back(e);
    return CMF_OK;
}
extern "C" int cmf_marker_times(cmf_ctx *c, double *ms, int64_t cap, int64_t *n) {
    if (!c || !n || (cap > 0 && !ms)) return fail(CMF_EINVAL, "bad argument");
    DeviceGuard dg(c->device);
    HIPCHK(hipStreamSynchronize(c->stream));
    *n = (int64_t)c->markers.size();
    for (int64_t i = 0; i < *n && i < cap; ++i) {
        float t = 0.f;
        if (i > 0) HIPCHK(hipEventElapsedTime(&t, c->markers[0], c->markers[i]));
        ms[i] = (double)t;
    }
    for (auto e : c->markers) c->evpool.push_back(e);
    c->markers.clear();
    return CMF_OK;
}
extern "C" int cmf_get_stream(cmf_ctx *c, void **stream) {
    if (!c || !stream) return fail(CMF_EINVAL, "null argument");
    *stream = (void *)c->stream;
    return CMF_OK;
}

extern "C" int cmf_kernel_timing(cmf_ctx *c, int enable) {
    if (!c) return fail(CMF_EINVAL, "null context");
    DeviceGuard dg(c->device);
    CHK(flush_timing(c));
    c->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return CMF_OK;
}
extern "C" int cmf_kernel_time(cmf_ctx *c, int cls, double *ms, int64_t *launches, double *flops) {
    if (!c || cls < 0 || cls >= CMF_K_COUNT) return fail(CMF_EINVAL, "bad kernel class");
    DeviceGuard dg(c->device);
    CHK(flush_timing(c));
    if (ms) *ms = c->ms[cls];
    if (launches) *launches = c->launches[cls];
    if (flops) *flops = c->flops[cls];
    return CMF_OK;
}
extern "C" int cmf_kernel_timing_reset(cmf_ctx *c) {
    if (!c) return fail(CMF_EINVAL, "null context");
    DeviceGuard dg(c->device);
    CHK(flush_timing(c));
    for (int i = 0; i < CMF_K_COUNT; ++i) { c->ms[i] = 0; c->launches[i] = 0; c->flops[i] = 0; }
    c->rh_credited = c->rh_gathered = 0.0;
    HIPCHK(hipMemsetAsync(c->dscalar + 7, 0, 8, c->stream));
    return CMF_OK;
}
extern "C" int cmf_rowhess_samples(cmf_ctx *c, double *credited, double *gathered) {
    if (!c) return fail(CMF_EINVAL, "null context");
    DeviceGuard dg(c->device);
    unsigned long long dev = 0;
    HIPCHK(hipMemcpyAsync(&dev, c->dscalar + 7, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (credited) *credited = c->rh_credited;
    if (gathered) *gathered = c->rh_gathered + (double)dev;
    return CMF_OK;
}

#include "cmf_newton.hip.h"
#include "cmf_init.hip.h"
#include "cmf_comm.hip.h"
