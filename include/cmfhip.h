/*
 * cmfhip.h -- C ABI of libcmfhip.so: the MI355X (gfx950) factor-update engine
 * for collective matrix factorisation  X ~ f(U V^T),  Y ~ f(V Z^T).
 *
 * The reference (smn-ailab/PyCMF) has no FFI on this path: its seam is the
 * Python solver object built at pycmf/cmf.py:437-451 and driven through
 * `fit_iterative_update` (pycmf/cmf.py:454 -> pycmf/cmf_solvers.py:132-195),
 * whose body is `update_step` (:172).  The closest thing to a native ABI is the
 * (dead) Cython module pycmf/cmf_newton_solver.pyx (`_newton_update_left`
 * :240-292, `_newton_update_V` :295-362).  The entry points below are what a
 * binding for that seam needs; each cites the reference code it replaces.
 *
 * Conventions
 *   - every function returns 0 on success or a CMF_E* code; the message is
 *     available from cmf_last_error() (thread-local).
 *   - host matrices are caller-owned, any element strides (in elements), read
 *     only unless stated.  Device state is owned by the context.
 *   - arithmetic on the device is float32 (north_star); host I/O is float64
 *     or float32.
 *   - one context drives one GPU from one host thread; contexts are not
 *     thread-safe.  With several GPUs use one process + one context per GPU
 *     and the *_partials / *_apply pair around an all-reduce (section "sharded").
 */
#ifndef CMFHIP_H
#define CMFHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmf_ctx cmf_ctx;

enum {
    CMF_OK = 0,
    CMF_EINVAL = 1,   /* bad argument / wrong call order            */
    CMF_EHIP = 2,     /* a HIP runtime call failed                  */
    CMF_ENOMEM = 3,   /* device or host allocation failed           */
    CMF_ENODEV = 4,   /* no usable gfx950 device                    */
    CMF_EUNSUPPORTED = 5,
    CMF_ERCCL = 6     /* RCCL missing or a collective failed         */
};

enum { CMF_U = 0, CMF_V = 1, CMF_Z = 2 };           /* factor selector          */
enum { CMF_LINK_LINEAR = 0, CMF_LINK_LOGIT = 1 };   /* cmf_solvers.py:27-33     */
enum { CMF_UPD_U = 1, CMF_UPD_V = 2, CMF_UPD_Z = 4 }; /* update_U/V/Z, :252-261 */
enum { CMF_NN_U = 1, CMF_NN_V = 2, CMF_NN_Z = 4 };  /* *_non_negative, :321-326 */

/* kernel classes for cmf_kernel_time() */
enum {
    CMF_K_GEMM_NN = 0,   /* C = A   B over a data-sized A: X V, Y Z, R V, W KR   */
    CMF_K_GEMM_TN = 1,   /* C = A^T B over a data-sized A: X^T U, Y^T V, R^T U   */
    CMF_K_GEMM_NT = 2,   /* f(L R^T) - T : residual / error    */
    CMF_K_ELEMWISE = 3,  /* slab sums, MU ratio, row updates   */
    CMF_K_EIGEN = 4,     /* batched symmetric Jacobi           */
    CMF_K_GEMM_SMALL = 5,/* factor-side products: Grams, F G, grad H^-1 */
    CMF_K_SPMM = 6,      /* native CSR: A F, A^T F, sum_nnz a_ij (l_i . r_j)  */
    CMF_K_ROWHESS = 7,   /* fused per-row gradient + Hessian over the sampled rows */
    CMF_K_GEMM_PAIR = 8, /* k_pad = 128: the two data passes of an MU half-iteration as one balanced launch (X^T U with Y Z, X V with Y^T V) */
    CMF_K_COUNT = 9
};

const char *cmf_last_error(void);
/* sha256 of the sources the library was built from (everything under csrc/ and this header): pycmf_amd/_lib.py refuses a library whose stamp differs
 * from the sources next to it                                                                                                   */
const char *cmf_source_hash(void);
int cmf_device_count(int *count);

/* ---- context ---------------------------------------------------------- */
/* `stream` is a hipStream_t the caller already owns (e.g. torch's current
 * stream) or NULL for a context-private stream.                            */
int cmf_ctx_create(cmf_ctx **out, int device, void *stream);
int cmf_ctx_destroy(cmf_ctx *ctx);
int cmf_sync(cmf_ctx *ctx);
/* tuning knobs (A/B measurements in one process): "gemm_pipe" 0..5 | 10 = staging schedule of the
 * data-pass GEMM kernels, "gemm_split" n = force the split-K factor (<= 0: heuristic),
 * "sparse_mode" 0 auto | 1 dense | 2 native CSR (set before cmf_set_data_csr),
 * "row_kernel" 1 fused gather kernel | 0 masked-dense GEMMs for per-row Newton sweeps,
 * "z_logit_hessian_l2" 1 (live Python path, cmf_solvers.py:505-506) | 0 (Cython twin,
 * cmf_newton_solver.pyx:287-290: Z's logit Hessian without l2 I),
 * "safe_inverse_cholesky" 1 | 0, "graph" 1 | 0 (replay MU / linear-Newton steps from a
 * captured hipGraph; automatically off while cmf_kernel_timing is enabled),
 * "gemm_arith" 0 (fp32 MFMA, default) | 1 (k_pad = 256 data passes on the bf16 matrix pipe: every fp32 operand
 * split exactly into three bf16 planes, six cross products, fp32 accumulation; planes stay resident),
 * "row_symmetric" 4 (default) | 3 | 1 | 0 (k_pad = 256 row kernel: 0 every 32 x 32 block of H_i | 1 the blocks on or above the block
 * diagonal, raw and weighted sample images | 3 ... from one sqrt-weighted image | 4 ... with the 8 diagonal blocks as three 16 x 16
 * sub-blocks each),
 * "eig_clamp" 1 (default) | 0 | 3, "rank1_clamp" 1 (default) | 0, "refine_rows_tol_ppm" 20 (default): how the rows flagged by the
 * threshold test of _safe_invert (cmf_solvers.py:346-356) are served (cmf_newton_clamp_routes below) and which of them are redone
 * in float64 (cmf_newton_clamp_stats below),
 * "shared_hessian_f64" 1 (default) | 0: the single Hessian of a linear-link sweep is accumulated (float64 Grams on
 * the float64 matrix pipe) and inverted in float64 | float32 Grams and float32 inverse,
 * "newton_schulz" 1 | 0 (k_pad = 256: rows whose eigenvalue clamp acts go through the GEMM-only
 * spectral clamp | through the Jacobi eigen-solver),
 * "sample_row_offset_u|v|z" n = global index of this context's first U / V / Z row in the keys of
 * the device sampler (a row shard then draws what the unsharded problem draws for its rows),
 * "row_classes" -1 (automatic, default) | 0 | 2..6: linear-link sides with sg_sample_ratio < 1 -- groups of that many
 * consecutive rows share the outer-product sums of the samples they have in common (same H_i as row by row, sums in
 * another order; cmf_solvers.py:414-428 with the identity link) | every row gathers its own list,
 * "direct_newton_step" 1 (default) | 0: linear shared-Hessian sweeps with l1 = 0 whose inverse is the plain one (k > 64)
 * update F <- clamp(s (T O) H^-1) in one product | form the gradient and subtract the step,
 * "row_certificates" 1 (default) | 0: half such a group shares one threshold test of _safe_invert (cmf_solvers.py:346-356)
 * through the positive semi-definite part of their Hessians the rows have in common | every row runs its own         */
int cmf_set_option(cmf_ctx *ctx, const char *name, int64_t value);

/* ---- problem ---------------------------------------------------------- */
/* Local shard sizes: X is m x d, Y is d x p, factors have k columns.
 * (pycmf/cmf_solvers.py:132-160 argument shapes).  Allocates device state. */
int cmf_set_problem(cmf_ctx *ctx, int64_t m, int64_t d, int64_t p, int k);

/* Dense uploads; element (i,j) is at ptr[i*rs + j*cs].  which: 0 = X, 1 = Y.
 * Replaces the implicit "X, Y are ndarrays" of cmf_solvers.py:132.          */
int cmf_set_data_f64(cmf_ctx *ctx, int which, const double *ptr, int64_t rs, int64_t cs);
int cmf_set_data_f32(cmf_ctx *ctx, int which, const float *ptr, int64_t rs, int64_t cs);
/* CSR upload (scipy layout), accept_sparse=('csr','csc') of pycmf/cmf.py:679;
 * CSC is converted by the caller.  Large, genuinely sparse inputs (density < 2 %,
 * dense image > 1 GB) stay CSR on the device (A and A^T images, SpMM kernels);
 * others are expanded to the dense layout.  cmf_set_option("sparse_mode", 1|2)
 * forces dense | native.                                                     */
int cmf_set_data_csr(cmf_ctx *ctx, int which, const int64_t *indptr, const int32_t *indices,
                     const double *data, int64_t nnz);
/* Synthetic |N(0,1)| fill on the device (counter-based; value of element
 * (gi,gj) depends only on seed and its GLOBAL coordinates, so a shard can be
 * generated in place): rows [row0,row0+rows) x cols [col0,col0+cols) of the
 * global matrix land in the local matrix.  Used by bench.py.                 */
/* representation of X / Y on the device: *dense = a dense float32 image exists, *native = the CSR pair (A, A^T) is resident */
int cmf_data_layout(cmf_ctx *ctx, int which, int *dense, int *native);
/* layout of the column-blocked SpMM images of a native sparse X / Y (the products safe_sparse_dot runs at cmf_solvers.py:232, :244):
 * out[0..3] = row groups, rows cut into pieces (more non-zeros than a wave's share of a group), pieces, accumulator rows per group --
 * of A; out[4..7] the same of A^T.  Zeros where an orientation has no blocked image.                                          */
int cmf_sparse_layout(cmf_ctx *ctx, int which, int64_t *out8);
/* rows x cols block of the dense device image into a packed host array (parity tests at full BASELINE sizes) */
int cmf_get_data_block_f32(cmf_ctx *ctx, int which, int64_t row0, int64_t nrows, int64_t col0, int64_t ncols, float *host_dst);
int cmf_fill_data_synthetic(cmf_ctx *ctx, int which, uint64_t seed, int64_t row0, int64_t col0);
int cmf_fill_factor_synthetic(cmf_ctx *ctx, int which, uint64_t seed, int64_t row0, double scale);
/* the same generator with the target distributions of the logit workloads: kind 0 = |N(0,1)|,
 * 1 = sigmoid(N(0,1)) (benchmarks/benchmark_cmf.py:78), 2 = Bernoulli(param) in {0, 1} (samples/toxic_comments.ipynb labels) */
int cmf_fill_data_synthetic_kind(cmf_ctx *ctx, int which, uint64_t seed, int64_t row0, int64_t col0, int kind, double param);
/* out (host, rows_out x ncols, row-major float64) = op(A) * B with A = X (which 0) or Y (which 1),
 * op = transpose when trans != 0, B host row-major float64 (b_rows x ncols).  The big products of the
 * initialisers' randomized SVD (sklearn randomized_svd called at pycmf/cmf.py:126,:149) run through
 * this on the data already resident on the device (dense MFMA GEMM or native CSR SpMM).            */
int cmf_data_matmul_f64(cmf_ctx *ctx, int which, int trans, const double *B, int64_t b_rows, int ncols, double *out);
/* Initialisers on the device copy (pycmf/cmf.py:41-202).
 * cmf_rsvd: randomized truncated SVD of X (which 0) or Y (which 1) as sklearn.utils.extmath.randomized_svd computes it for
 * _initialize_mf (cmf.py:126, :149): `omega` is the Gaussian test matrix (cols x size row-major float64 when transpose == 0,
 * rows x size when transpose != 0 -- sklearn's 'auto' transposition -- drawn by the caller from NumPy's RandomState),
 * n_iter normalised power iterations (CholeskyQR2 in float64 in place of sklearn's LU: the same subspace), then the SVD of
 * the small matrix.  Outputs, float64 row-major: U (rows x k), S (k), Vt (k x cols).  Sign convention left to the caller.
 * cmf_data_sum: sum of all entries of X and Y (M.mean() of the 'random' / 'nndsvda' / 'nndsvdar' rules, cmf.py:111, :186). */
int cmf_rsvd(cmf_ctx *ctx, int which, int transpose, int k, int size, int n_iter, const double *omega,
             double *U, double *S, double *Vt);
int cmf_data_sum(cmf_ctx *ctx, double *sum_x, double *sum_y);
/* float64 sums of the dense device image of X / Y over blocks of 256 rows (axis 0: out[rows_pad / 256][cols_pad]) or 256 columns
 * (axis 1: out[rows_pad][cols_pad / 256]): the inputs of the per-tile checksums of the full-size parity tests -- the column sums
 * of a 256-row output tile of X^T U, X V, Y Z, Y^T V (cmf_solvers.py:232, :238, :244) are (block sums)^T times the factor        */
int cmf_data_block_sums_f64(cmf_ctx *ctx, int which, int axis, double *out);
/* read back a block of X or Y (tests) */
int cmf_get_data_f32(cmf_ctx *ctx, int which, float *ptr, int64_t rs, int64_t cs);

/* Factors (in/out, pycmf/cmf_solvers.py:195 "return U, V, Z" after in-place
 * mutation :255,:259,:263,:324).                                             */
int cmf_set_factor_f64(cmf_ctx *ctx, int which, const double *ptr, int64_t rs, int64_t cs);
int cmf_get_factor_f64(cmf_ctx *ctx, int which, double *ptr, int64_t rs, int64_t cs);

/* ---- MU solver: MUSolver.update_step, pycmf/cmf_solvers.py:248-263 ----- */
int cmf_mu_step(cmf_ctx *ctx, double l1, double l2, int update_mask);
/* cmf_mu_step AND the error metric of the factors it leaves (compute_factorization_error, cmf_solvers.py:36-42, as the loop
 * evaluates it every 10th iteration, :175-187) from the products of the step itself: ||X||^2 - 2 <U, X V> + <U^T U, V^T V> (the
 * expansion of sklearn's sparse path, :40) -- no pass over X and Y.  Falls back to cmf_residual_sq per side where the expansion
 * would cancel (e^2 < 1e-3 ||.||^2), the side is not dense or its factor is not updated.  *ex2 / *ey2 (nullable): squared
 * Frobenius residuals of the local shard, as cmf_residual_sq.                                                              */
int cmf_mu_step_error(cmf_ctx *ctx, double l1, double l2, int update_mask, double *ex2, double *ey2);

/* sharded form: rank g holds rows of X/U and columns of
 * Y/Z, V replicated.  buf is a DEVICE buffer of cmf_v_buf_elems() floats:
 *   [ X_g^T U_g + Y_g Z_g  (d_pad x k_pad) | U_g^T U_g + Z_g^T Z_g (k_pad x k_pad) ]
 * The caller all-reduces it (RCCL) between the two calls.
 * cmf_solvers.py:242-246 (V numerator/denominator).                        */
int cmf_v_buf_elems(cmf_ctx *ctx, int64_t *n);
int cmf_mu_v_partials(cmf_ctx *ctx, float *dev_buf);
/* the partial of cmf_mu_v_partials for a 256-aligned block of V rows only (dense X, Y), the Gram part when with_gram: lets a
 * sharded driver compute block c + 1 while block c is all-reduced in the background (cmf_comm_allreduce_f32_bg)              */
int cmf_mu_v_partials_rows(cmf_ctx *ctx, float *dev_buf, int64_t row0, int64_t nrows, int with_gram);
int cmf_mu_v_apply(cmf_ctx *ctx, const float *dev_buf, double l1, double l2);
int cmf_mu_uz_update(cmf_ctx *ctx, double l1, double l2, int update_mask);
/* row-blocked V update: the same single sum over the ranks, cut in two around the epilogue (SURVEY.md 8(e) "Partitioning": reduce-
 * scatter + epilogue + all-gather).  Block r of V = rows [r B, (r + 1) B), B = block_rows (a multiple of 256, world * B >= d_pad;
 * cmf_mu_blocked_layout also grows the allocation behind V to world * B rows so that the all-gather runs in place on
 * cmf_factor_dev_ptr(V) -- query that pointer AFTER this call).  Per iteration:
 *   cmf_mu_v_partials_split(P, G)                     P: pbuf_elems floats, zero beyond d_pad rows; G: k_pad^2 floats
 *   all-reduce G (k_pad^2), reduce-scatter P (block_rows * k_pad per rank)
 *   cmf_mu_v_apply_rows(P + r B k_pad, G, r B, rows)  rows = the part of block r inside d_pad (may be 0)
 *   cmf_mu_gram_v_rows(r B, rows, G2); all-reduce G2; all-gather V (block_rows * k_pad per rank)
 *   cmf_mu_uz_update_gram(G2, ...)
 * cmf_solvers.py:242-246, :230-240; the sums over rows are the same sums, regrouped.                                          */
int cmf_mu_blocked_layout(cmf_ctx *ctx, int world, int64_t *block_rows, int64_t *pbuf_elems);
int cmf_mu_v_partials_split(cmf_ctx *ctx, float *dev_P, float *dev_G);
int cmf_mu_v_apply_rows(cmf_ctx *ctx, const float *dev_P_rows, const float *dev_G, int64_t row0, int64_t nrows, double l1, double l2);
int cmf_mu_gram_v_rows(cmf_ctx *ctx, int64_t row0, int64_t nrows, float *dev_G2);
int cmf_mu_uz_update_gram(cmf_ctx *ctx, const float *dev_G2, double l1, double l2, int update_mask);

/* ---- Newton solver: NewtonSolver.update_step, cmf_solvers.py:510-522 --- */
/* sample index lists (parity mode): for sg_ratio < 1 the caller passes the
 * indices the reference would have drawn (cmf_solvers.py:328-344), row after
 * row, as int32: u_idx[m][su] over d, z_idx[p][su] over d,
 * vx_idx[d][sm] over m, vy_idx[d][sp] over p.  NULL when sg_ratio == 1.
 * Every list must hold DISTINCT indices (the reference's permutation()[:s] never repeats one): the shared-partial-sum form of
 * linear sampled sides counts a repeated index once, the row-by-row form once per occurrence.                              */
int cmf_newton_step(cmf_ctx *ctx, double alpha, double l1, double l2,
                    int x_link, int y_link, int nn_mask, int update_mask,
                    double hessian_pertubation, double sg_ratio,
                    const int32_t *u_idx, const int32_t *z_idx,
                    const int32_t *vx_idx, const int32_t *vy_idx);

/* Throughput mode of sg_ratio < 1: the per-row samples (exactly int(n*ratio) distinct
 * candidates per row, uniform) are drawn on the device from a counter-based generator keyed by
 * (seed, sweep, row) -- same distribution as cmf_solvers.py:328-344, not NumPy's stream.       */
int cmf_newton_step_device_sampled(cmf_ctx *ctx, double alpha, double l1, double l2,
                                   int x_link, int y_link, int nn_mask, int update_mask,
                                   double hessian_pertubation, double sg_ratio, uint64_t seed);

/* the index lists that throughput mode draws for rows [row0, row0 + nrows) of one sweep (0: U, lists over d; 1: Z, over d;
 * 2: V / X side, over m; 3: V / Y side, over p): int(n * sg_ratio) ascending indices per row into host memory        */
int cmf_sample_lists(cmf_ctx *ctx, int sweep, uint64_t seed, double sg_ratio, int64_t row0, int64_t nrows, int32_t *host_out);

/* sharded Newton, linear links and sg_ratio == 1 only (same buffer shape as
 * the MU pair: gradient partial | Gram partial).                            */
int cmf_newton_uz_update(cmf_ctx *ctx, double alpha, double l1, double l2,
                         int nn_mask, int update_mask, double hessian_pertubation);
int cmf_newton_v_partials(cmf_ctx *ctx, double alpha, float *dev_buf);
int cmf_newton_v_apply(cmf_ctx *ctx, const float *dev_buf, double l1, double l2,
                       int nn_mask, double hessian_pertubation);

/* The same V sweep in its re-associated, cond(H)-independent form (default of cmf_newton_step; option
 * "newton_reassoc"): the reference's  V - grad Hinv  with  grad = (V G - P) + l1 sign V + l2 V,  H = G + l2 I
 * (cmf_solvers.py:436-450, :321-326) is evaluated as  V (I - H Hinv) + P' - l1 sign(V) Hinv  with
 * P' = X^T (alpha U Hinv) + Y ((1 - alpha) Z Hinv), Hinv applied to the factors in float64 BEFORE the float32 data pass.
 * A row-sharded run sums two buffers over the ranks: gbuf (k_pad^2 float64, device) between _gram and _products, pbuf
 * (d_pad * k_pad float32, device, clobbered by _finish) between _products and _finish.                                   */
int cmf_newton_v_gram(cmf_ctx *ctx, double alpha, double *dev_gbuf);
int cmf_newton_v_products(cmf_ctx *ctx, double alpha, double l2, double hessian_pertubation,
                          const double *dev_gbuf, float *dev_pbuf);
int cmf_newton_v_finish(cmf_ctx *ctx, float *dev_pbuf, double l1, int nn_mask);

/* ---- error metric: compute_factorization_error, cmf_solvers.py:36-42 --- */
/* squared Frobenius residuals of the local shard:
 *   *ex2 = ||X - f(U V^T)||^2, *ey2 = ||Y - f(V Z^T)||^2                    */
int cmf_residual_sq(cmf_ctx *ctx, int x_link, int y_link, double *ex2, double *ey2);
int cmf_data_sq(cmf_ctx *ctx, double *x2, double *y2);  /* ||X||^2, ||Y||^2 */

/* ---- batched safe inverse (exposed for tests): _safe_invert :346-356 --- */
/* H: n symmetric k x k float64 matrices (host), out: Q diag(1/max(|l|,pert)) Q^T */
int cmf_safe_invert_batch(cmf_ctx *ctx, const double *H, double *out, int n, int k, double pert);
/* The step of _row_newton_update (:321-326) for n independent rows: out_b = g_b * safe_inverse(H_b) (H: n symmetric k x k
 * float64 host matrices, g / out: n x k), through the float32 path the per-row sweeps take.  method 0: the sweeps' own dispatch
 * (Cholesky where lambda_min >= pert, the clamp path for the rest); 1: every matrix through the tridiagonal eigen-solve
 * (cmf_eigclamp.hip.h; 64 < k <= 256), lam (nullable, n x k) receives its eigenvalues in the order the QL iteration left them. */
int cmf_safe_solve_batch(cmf_ctx *ctx, const double *H, const double *g, double *out, double *lam, int n, int k,
                         double pert, int method);

/* Conditioning record of the per-row Newton sweeps.  The spectral clamp of _safe_invert (pycmf/cmf_solvers.py:346-356) acts on a
 * row's Hessian only when its smallest eigenvalue is below `pert`; on the device that Hessian is a float32 matrix, whose
 * eigenvalues are resolved to about eps32 * ||H||, so the clamped directions of the inverse carry a relative error of the order
 * eps32 * ||H|| / pert (the reference works in float64).  rows = matrices the float32 clamp acted on since the last reset,
 * max_ratio = the largest ||H||_F / pert among them (0 when there were none).  The stated tolerances hold
 * for max_ratio up to ~1e4.  Clamped rows above 3e3 (option "refine_rows_ratio"; plain solves above a condition estimate of 1e3,
 * "refine_rows_cond") are REDONE IN FLOAT64 (Hessian from float64 sums,
 * float64 clamp, float64 step; option "refine_rows", default on, at most "refine_rows_max" = 16384 rows per sweep, k <= 256) and
 * counted in `refined` instead; only rows left in float32 enter rows / max_ratio, and the estimator warns when max_ratio
 * exceeds 1e4.  The shared Hessians of the linear unsampled sweeps are formed and clamped in float64 and never appear here.
 * plain_cond: the largest condition estimate max H_ii / min L_ii^2 (<= cond(H)) over ALL rows solved by plain Cholesky -- rows above
 * the refinement ratio by that estimate are refined too; the ones left in float32 also enter max_ratio. */
int cmf_newton_clamp_stats(cmf_ctx *ctx, int64_t *rows, double *max_ratio, int64_t *refined, double *plain_cond, int reset);
/* Which route the rows flagged by the threshold test took through _safe_invert's clamp (pycmf/cmf_solvers.py:346-356) since the
 * context was created: eigen_rows through the tridiagonal eigen-solve (Householder + QL, any spectrum), rank1_rows through the
 * rank-one shortcut (one eigenvalue above pert, every other one certified below it by a Cholesky factorisation of
 * (pert - delta) I - (H - lambda_1 q q^T); positive semi-definite Hessians at k_pad = 256; option "rank1_clamp", default on). */
int cmf_newton_clamp_routes(cmf_ctx *ctx, int64_t *eigen_rows, int64_t *rank1_rows);

/* float64 path of the ONE shared Hessian of a linear-link sweep (cmf_solvers.py:407-410, :448-450): H is k x k
 * float64 on the host, k = the problem's n_components; out = Q diag(1/max(|l|,pert)) Q^T computed in float64 on
 * the device (positive semi-definite H), returned after its rounding to float32 (the form the step product uses) */
int cmf_safe_invert_f64(cmf_ctx *ctx, const double *H, double *out, int k, double pert);

/* ---- measurement ------------------------------------------------------ */
/* when enabled every kernel launch is bracketed by hipEvents on the context's
 * stream; cmf_kernel_time returns accumulated ms, launch count and algorithmic
 * flops (2*M*N*K of every GEMM launched, 0 for the other classes) per class.
 * enable = 2: only the data-pass classes (GEMM_NN, GEMM_TN, SPMM, ROWHESS) are
 * bracketed -- an event pair costs a few microseconds of stream serialisation per
 * launch, 7 % of a 0.9 ms iteration with 11 launches                          */
int cmf_kernel_timing(cmf_ctx *ctx, int enable);
int cmf_kernel_time(cmf_ctx *ctx, int kernel_class, double *ms, int64_t *launches, double *flops);
int cmf_kernel_timing_reset(cmf_ctx *ctx);
/* per-row Newton accounting since the last reset, counted while timing is enabled: `credited` = sample rows of the
 * algorithm (sum over updated rows of their list lengths, pycmf/cmf_solvers.py:414-428 runs one outer product per such
 * pair), `gathered` = sample rows that actually went through the outer-product kernel (fewer when linear sampled sides
 * share partial sums between rows: option "row_classes")                                                           */
int cmf_rowhess_samples(cmf_ctx *ctx, double *credited, double *gathered);
/* stream markers (bench.py's per-iteration time series, auditable beside the whole-region clock): cmf_marker records an
 * event on the launch stream; cmf_marker_times waits for the stream, writes the elapsed ms of every marker since the
 * first one (at most cap entries), stores the marker count in *n and clears the list                                */
int cmf_marker(cmf_ctx *ctx);
int cmf_marker_times(cmf_ctx *ctx, double *ms, int64_t cap, int64_t *n);
/* the hipStream_t the context launches on (its own, or the one given to cmf_ctx_create): a caller that enqueues
 * collectives between the *_partials / *_apply calls orders them on this stream                                      */
int cmf_get_stream(cmf_ctx *ctx, void **stream);
/* diagnostic: one X*V data pass with s_memtime / s_memrealtime stamps around the main loop;
 * median in-kernel shader clock (GHz) and main-loop duration (us) over the workgroups        */
int cmf_debug_clock(cmf_ctx *ctx, double *ghz, double *loop_us);
/* padded device geometry (m_pad, d_pad, p_pad, k_pad) */
int cmf_get_geometry(cmf_ctx *ctx, int64_t *m_pad, int64_t *d_pad, int64_t *p_pad, int *k_pad);
/* device pointers of the factor blocks (float32, row-major, ld = k_pad) */
int cmf_factor_dev_ptr(cmf_ctx *ctx, int which, float **ptr);
/* zero-filled device scratch owned by the context: the partial / staging buffer of the sharded entry points when the
 * caller brings no device allocator (bench.py at N = 1 runs without PyTorch); freed by cmf_scratch_free, by the next
 * cmf_set_problem or by cmf_ctx_destroy                                                                             */
int cmf_scratch_alloc(cmf_ctx *ctx, int64_t bytes, void **dev_ptr);
int cmf_scratch_free(cmf_ctx *ctx, void *dev_ptr);
/* device-to-device copies of all valid rows of a factor, k_pad floats per row, on the context's stream:
 * what a row-sharded Newton driver exchanges between its U/Z-sweep and V-sweep contexts
 * (the reference keeps U, V, Z in one address space: cmf_solvers.py:510-522)                      */
int cmf_export_factor_rows(cmf_ctx *ctx, int which, float *dev_dst);
int cmf_import_factor_rows(cmf_ctx *ctx, int which, const float *dev_src);


/* ---- the outer loop: _IterativeCMFSolver.fit_iterative_update, pycmf/cmf_solvers.py:132-195 --------------------------------
 * error at init (:170); for n_iter = 1 .. max_iter: one update_step (:172) -- cmf_mu_step, cmf_newton_step, or
 * cmf_newton_step_device_sampled(seed + n_iter) when sg_ratio < 1 (host-drawn index lists cannot enter here: that configuration
 * keeps its loop on the host) --; every check_every-th iteration (the reference: 10) when tol > 0 the error
 * alpha_err ||X - f(U V^T)||_F + (1 - alpha_err) ||Y - f(V Z^T)||_F (:128-130) and the stopping test
 * (previous - error) / error_at_init < tol (:183-186).  *n_iter = the iteration the loop stopped at (max_iter when it ran out,
 * exactly Python's loop variable).  err_trace / time_trace (nullable, trace_cap entries each): the error at init and at every
 * check, and the seconds since the call started at those points; *n_trace = how many there were.  One 16-byte read-back per
 * check is all that crosses the boundary; the step body is replayed from a hipGraph where it is a fixed launch sequence.      */
enum { CMF_SOLVER_MU = 0, CMF_SOLVER_NEWTON = 1 };
typedef struct {
    int solver;                 /* CMF_SOLVER_MU | CMF_SOLVER_NEWTON */
    double l1, l2;
    double alpha;               /* Newton: weight of the X side (cmf_solvers.py:510-522); MU: unused */
    double alpha_err;           /* weight of the X side in the error metric (MU: the constructor default 0.5, :99) */
    int x_link, y_link;         /* CMF_LINK_* (MU: both linear) */
    int nn_mask, update_mask;
    double hessian_pertubation, sg_ratio;
    uint64_t seed;              /* device sampler: iteration n_iter draws under seed + n_iter */
} cmf_run_params;
int cmf_run(cmf_ctx *ctx, const cmf_run_params *params, int max_iter, double tol, int check_every, int *n_iter,
            double *err_trace, double *time_trace, int trace_cap, int *n_trace);

/* ---- collectives of the sharded solvers: RCCL over xGMI, one communicator per context -------------
 * The reference is single-process; what is replaced here is the sum over row blocks hidden in its products
 * (cmf_solvers.py:242-246: X^T U + Y Z and U^T U + Z^T Z are sums over the rows each rank owns).  librccl.so.1 is loaded on the
 * first call (no link-time dependency).  Every collective is enqueued on the context's stream.  Bootstrap: one rank calls
 * cmf_comm_unique_id and hands the 128 bytes to the others out of band (pycmf_amd/comm.py: a file next to the job), then every
 * rank calls cmf_comm_init on its own context (one process per GPU).                                                        */
#define CMF_COMM_ID_BYTES 128
int cmf_comm_unique_id(char *id128);
int cmf_comm_init(cmf_ctx *ctx, int rank, int world, const char *id128);
int cmf_comm_destroy(cmf_ctx *ctx);                 /* also done by cmf_ctx_destroy */
int cmf_comm_info(cmf_ctx *ctx, int *rank, int *world);
/* in-place sum over the ranks (device memory): the (d + k) k partial buffer of cmf_mu_v_partials / cmf_newton_v_partials, the
 * d k buffer of cmf_newton_v_products (float32); the k^2 Gram of cmf_newton_v_gram (float64)                                */
int cmf_comm_allreduce_f32(cmf_ctx *ctx, float *dev_buf, int64_t n);
int cmf_comm_allreduce_f64(cmf_ctx *ctx, double *dev_buf, int64_t n);
/* the float32 all-reduce in the background: on a side stream, behind what the context's stream holds at the call; kernels
 * launched on the context's stream afterwards overlap with it.  cmf_comm_join: the context's stream waits for all of them.
 * cmf_comm_exposed_ms: time that stream spent waiting in joins (while cmf_comm_timing) -- the exposed part; the rest of the
 * collectives' duration (cmf_comm_stats) was hidden under compute                                                              */
int cmf_comm_allreduce_f32_bg(cmf_ctx *ctx, float *dev_buf, int64_t n);
int cmf_comm_join(cmf_ctx *ctx);
int cmf_comm_exposed_ms(cmf_ctx *ctx, double *ms, int reset);
/* in-place all-gather of equal chunks (factor rows of the row-sharded Newton): rank r's elems_per_rank floats already sit at
 * dev_full + r * elems_per_rank                                                                                             */
int cmf_comm_allgather_f32(cmf_ctx *ctx, float *dev_full, int64_t elems_per_rank);
/* in-place reduce-scatter of equal chunks: every rank holds world * elems_per_rank floats; afterwards chunk `rank` of the rank's
 * own buffer is the sum over the ranks of that chunk.  With cmf_comm_allgather_f32 on the updated rows it is the ONE all-reduce of
 * the MU V update (cmf_solvers.py:242-246) cut in two around the row-blocked epilogue (cmf_mu_v_apply_rows)                   */
int cmf_comm_reduce_scatter_f32(cmf_ctx *ctx, float *dev_full, int64_t elems_per_rank);
/* ncclGroupStart / ncclGroupEnd around the collectives enqueued in between: ONE launch point on the context's stream.  The
 * row-blocked MU iteration (cmf_solvers.py:242-246 cut around the epilogue) is two groups: {all-reduce of U^T U + Z^T Z,
 * reduce-scatter of X^T U + Y Z} and {all-reduce of the ranks' shares of V^T V, all-gather of V}.  cmf_comm_launch_points:
 * collectives outside groups + groups since the last cmf_comm_stats(reset)                                                    */
int cmf_comm_group_start(cmf_ctx *ctx);
int cmf_comm_group_end(cmf_ctx *ctx);
int cmf_comm_launch_points(cmf_ctx *ctx, int64_t *n);
/* what RCCL reports about the communicator (ncclCommCount, ncclCommUserRank) -- not what the launcher's environment says   */
int cmf_comm_count(cmf_ctx *ctx, int *ranks_seen, int *rank_seen);
/* at most 16 host scalars, op 0 = sum, 1 = max; waits for the result (convergence test on the global error, the slowest
 * rank's clock of bench.py); cmf_comm_barrier = one such reduction                                                          */
int cmf_comm_allreduce_host_f64(cmf_ctx *ctx, double *vals, int n, int op);
int cmf_comm_barrier(cmf_ctx *ctx);
/* accounting: calls and payload bytes since the last reset; with cmf_comm_timing(1) also the milliseconds the collectives
 * occupied the stream (events around each one, waiting for the slowest rank included)                                       */
int cmf_comm_timing(cmf_ctx *ctx, int enable);
int cmf_comm_stats(cmf_ctx *ctx, int64_t *calls, int64_t *bytes, double *ms, int reset);
/* the same per kind of collective (no reset: read before cmf_comm_stats(..., 1))                                            */
enum { CMF_COMM_ALLREDUCE_F32 = 0, CMF_COMM_ALLREDUCE_F64 = 1, CMF_COMM_ALLGATHER_F32 = 2, CMF_COMM_REDUCE_SCATTER_F32 = 3, CMF_COMM_GROUP = 4, CMF_COMM_KINDS = 5 };
int cmf_comm_stats_kind(cmf_ctx *ctx, int kind, int64_t *calls, int64_t *bytes, double *ms);
/* dev[0 .. n) *= factor on the context's stream (the measurement double of the collectives stands in for the peers with the
 * rank's own partial times the world size: finite iterates without a communicator)                                           */
int cmf_scale_f32(cmf_ctx *ctx, float *dev, int64_t n, double factor);
/* raw copies between caller-held device pointers (scratch, partial buffers) and host memory on the context's stream; both wait */
int cmf_copy_to_host(cmf_ctx *ctx, const void *dev, void *host, int64_t bytes);
int cmf_copy_from_host(cmf_ctx *ctx, void *dev, const void *host, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* CMFHIP_H */
