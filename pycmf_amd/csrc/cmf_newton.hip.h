// cmf_newton.hip.h -- Newton-Raphson sweeps U -> Z -> V (included by cmf_api.hip).
//
// Reference: NewtonSolver.update_step and helpers, pycmf/cmf_solvers.py:318-522
// (the live pure-Python path; the Cython twin pycmf/cmf_newton_solver.pyx:240-362
// computes the same quantities).  Every sweep is row-parallel (each row reads only
// its own pre-sweep value and the other, frozen factors), so a sweep becomes a
// handful of batched operations:
//   * "shared" form (linear link, no sampling): one k x k Hessian for all rows,
//       grad = s (F G - T O) + l1 sign F + l2 F,  F <- F - grad H^-1      (:396-410)
//   * "per-row" form (logit link and/or sg_sample_ratio < 1):
//       R = s m (f(F O^T) - T), W = s m w(F O^T)   one NT GEMM with fused epilogue
//       grad = R O + reg                            big NN/TN GEMM
//       H_i  = sum_j W_ij o_j o_j^T  (+ l2 I)       GEMM against the Khatri-Rao
//                                                   square KR(O)[j] = o_j (x) o_j
//       H_i^-1 by batched Jacobi, step_i = g_i H_i^-1                     (:412-430)
//   m is the 0/1 sample mask built from the index lists the host drew (:328-344).

template <typename... Args>
static int launch_ew(cmf_ctx *c, void (*kern)(Args...), int64_t n, Args... args) {
    Timed tm(c, CMF_K_ELEMWISE);
    if (n <= 0) return CMF_OK;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, c->stream, args...);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// Hout_i = safe_inverse(Hin_i) for nmat k_pad x k_pad matrices (valid order n).
// Cholesky fast path first (exact when lambda_min >= pert), Jacobi for the flagged rest.
static int ns_clamp_images(cmf_ctx *c, const float *Hc, const int *idx, int nf, int n, int kp, double pert, float **M_out);

// ONE matrix, |lambda| / clamp by the Hestenes-Jacobi sweep spread over the chip: one launch per round-robin step
// (cmf_eigen.hip.h), one 4-byte read-back per sweep
static int jacobi_chipwide(cmf_ctx *c, const float *Hin, float *Hout, int *flags, int n, int kp, double pert) {
    const int N = n + (n & 1);
    CHK(ensure(c, c->eigws, (size_t)(2 * (size_t)n * n + n) * sizeof(float)));
    float *Bw = (float *)c->eigws.p, *Vw = Bw + (size_t)n * n, *invw = Vw + (size_t)n * n;
    hipLaunchKernelGGL(jacobi_init_kernel, dim3(std::min(1024, (n * n + 63) / 64)), dim3(64), 0, c->stream, Hin, Bw, Vw, n, kp);
    for (int sweep = 0; sweep < 30; ++sweep) {
        HIPCHK(hipMemsetAsync(flags, 0, sizeof(int), c->stream));
        for (int st = 0; st < N - 1; ++st)
            hipLaunchKernelGGL(jacobi_pair_step_kernel, dim3(N / 2), dim3(64), 0, c->stream, Bw, Vw, n, N, st, flags);
        int rotated = 0;
        HIPCHK(hipMemcpyAsync(&rotated, flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!rotated) break;
    }
    hipLaunchKernelGGL(jacobi_sigma_kernel, dim3(n), dim3(64), 0, c->stream, (const float *)Bw, invw, n, (float)pert);
    hipLaunchKernelGGL(jacobi_compose_kernel, dim3((unsigned)(((size_t)kp * kp + 255) / 256)), dim3(256), 0, c->stream, (const float *)Vw,
                       (const float *)invw, Hout, n, kp);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// note the matrices the float32 spectral clamp is about to act on (see clamp_stats_kernel).  refine: those with ||H||_F / pert above
// the refinement ratio are not recorded but collected (chunk-relative, ascending) in c->bad_host for refine_rows64 -- one 4-byte
// read-back per call; a list that would take the sweep over opt_refine_max is declined and recorded instead.
// Frows (round 6): the factor rows the matrices update.  With it (and the float32 steps already in `step`) a matrix is listed only
// when the float32 error bound of its row's update exceeds opt_refine_tol (clamp_stats_kernel); the caller then ran the float32
// clamp path for every flagged matrix BEFORE this call, `flags` is its snapshot of the flags, and nothing is unflagged here.
static int clamp_stats(cmf_ctx *c, const float *Hc, int *flags, int64_t nmat, int n, int kp, int64_t stride, double pert,
                       bool refine = false, const float *condest = nullptr, float *step = nullptr, const float *Frows = nullptr) {
    c->bad_host.clear();
    if (nmat <= 0) return CMF_OK;
    if (!c->clampstat.p) {
        CHK(ensure(c, c->clampstat, 16));
        HIPCHK(hipMemsetAsync(c->clampstat.p, 0, 16, c->stream));
    }
    unsigned long long *cnt = (unsigned long long *)c->clampstat.p;
    unsigned *mx = (unsigned *)((char *)c->clampstat.p + 8);
    const float thr = (float)c->opt_refine_ratio, thrp = (float)c->opt_refine_cond;
    const float tol = Frows ? (float)c->opt_refine_tol : 0.f;
    const float *bstep = Frows ? step : nullptr;
    const bool want = refine && c->opt_refine && c->hess_psd && c->refined_sweep < c->opt_refine_max;
    if (!want) {
        hipLaunchKernelGGL(clamp_stats_kernel, dim3((unsigned)nmat), dim3(64), 0, c->stream, Hc, flags, n, kp, stride, (float)pert, cnt, mx, thr, 0,
                           (int *)nullptr, condest, thrp);
        HIPCHK(hipGetLastError());
        return CMF_OK;
    }
    CHK(ensure(c, c->badbuf, (size_t)(nmat + 1) * sizeof(int)));
    int *bad = (int *)c->badbuf.p;
    HIPCHK(hipMemsetAsync(bad, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(clamp_stats_kernel, dim3((unsigned)nmat), dim3(64), 0, c->stream, Hc, flags, n, kp, stride, (float)pert, cnt, mx, thr, 1, bad, condest, thrp, bstep, Frows, tol, Frows ? 1 : 0);
    HIPCHK(hipGetLastError());
    int nb = 0;
    HIPCHK(hipMemcpyAsync(&nb, bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nb <= 0) return CMF_OK;
    if (c->refined_sweep + nb > c->opt_refine_max) { // too many for this sweep: they stay float32 and are recorded as such
        hipLaunchKernelGGL(clamp_stats_kernel, dim3((unsigned)nmat), dim3(64), 0, c->stream, Hc, flags, n, kp, stride, (float)pert, cnt, mx, thr, 2,
                           (int *)nullptr, condest, thrp, bstep, Frows, tol, Frows ? 1 : 0);
        HIPCHK(hipGetLastError());
        c->refined_sweep = c->opt_refine_max;
        return CMF_OK;
    }
    c->bad_host.resize((size_t)nb);
    HIPCHK(hipMemcpyAsync(c->bad_host.data(), bad + 1, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    // the listed matrices skip the float32 spectral clamp (refine_rows64 redoes them from scratch; at C3 with l2 = 0 that clamp was
    // 0.75 s per iteration of work thrown away): only where the batched refinement will take them
    if (step && !Frows && c->opt_refine_batched && c->k > 64 && c->k <= 256 && c->hess_psd)
        hipLaunchKernelGGL(unflag_listed_kernel, dim3((unsigned)nb), dim3(64), 0, c->stream, flags, (const int *)(bad + 1), nb, step, kp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    std::sort(c->bad_host.begin(), c->bad_host.end());
    c->refined_sweep += nb;
    return CMF_OK;
}

// Rows whose Hessian went through the float32 spectral clamp since the last reset (and were not redone in float64), the largest
// ||H||_F / pert among them, and the rows redone in float64.
extern "C" int cmf_newton_clamp_stats(cmf_ctx *c, int64_t *rows, double *max_ratio, int64_t *refined, double *plain_cond, int reset) {
    if (!c) return fail(CMF_EINVAL, "null context");
    DeviceGuard dg(c->device);
    unsigned long long cnt = 0;
    float ratio = 0.f, plain = 0.f;
    if (c->clampstat.p) {
        unsigned char host[16];
        HIPCHK(hipMemcpyAsync(host, c->clampstat.p, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        memcpy(&cnt, host, 8);
        memcpy(&ratio, host + 8, 4);
        memcpy(&plain, host + 12, 4);
        if (reset) HIPCHK(hipMemsetAsync(c->clampstat.p, 0, 16, c->stream));
    }
    if (rows) *rows = (int64_t)cnt;
    if (max_ratio) *max_ratio = (double)ratio;
    if (refined) *refined = c->refined_total;
    if (plain_cond) *plain_cond = (double)plain;
    if (reset) c->refined_total = 0;
    return CMF_OK;
}

// Which route the flagged rows of the per-row sweeps took since the context was created: tridiagonal eigen-solve, rank-one shortcut.
extern "C" int cmf_newton_clamp_routes(cmf_ctx *c, int64_t *eigen_rows, int64_t *rank1_rows) {
    if (!c) return fail(CMF_EINVAL, "null context");
    if (eigen_rows) *eigen_rows = c->eig_clamp_rows;
    if (rank1_rows) *rank1_rows = c->rank1_rows;
    return CMF_OK;
}

static int safe_inverse_dev(cmf_ctx *c, const float *Hin, float *Hout, int nmat, int n, int kp, double pert, bool psd = false,
                            bool refine = false) {
    if (nmat <= 0) return CMF_OK;
    const int64_t stride = (int64_t)kp * kp;
    Timed tm(c, CMF_K_EIGEN);
    const int *need = nullptr;
    const size_t tri_bytes = (size_t)n * (n + 1) / 2 * sizeof(float);
    const bool inplace = (Hin == Hout);
    if (c->opt_chol && nmat == 1 && n <= 256 && c->have_problem && kp == c->kp && !inplace) {
        // one shared Hessian: let k_pad workgroups each factor it in registers and solve for one unit
        // vector -- H^-1 row by row (H^-1 is symmetric) at the latency of a single solve
        CHK(ensure(c, c->eigflag, (size_t)kp * sizeof(int)));
        int *flags = (int *)c->eigflag.p;
        hipLaunchKernelGGL(axpby_diag_kernel, dim3((kp * kp + 255) / 256), dim3(256), 0, c->stream, c->Eye, (const float *)Hin, 0.0f,
                           (const float *)nullptr, 0.f, 1.0f, kp, n);
        const dim3 grid((unsigned)kp), block(256);
        if (n <= 32) hipLaunchKernelGGL((chol_solve_kernel<2>), grid, block, 0, c->stream, Hin, (const float *)c->Eye, Hout, flags, n, kp, (int64_t)0, (float)pert, kp);
        else if (n <= 64) hipLaunchKernelGGL((chol_solve_kernel<4>), grid, block, 0, c->stream, Hin, (const float *)c->Eye, Hout, flags, n, kp, (int64_t)0, (float)pert, kp);
        else if (n <= 128) hipLaunchKernelGGL((chol_solve_kernel<8>), grid, block, 0, c->stream, Hin, (const float *)c->Eye, Hout, flags, n, kp, (int64_t)0, (float)pert, kp);
        else hipLaunchKernelGGL((chol_solve_kernel<16>), grid, block, 0, c->stream, Hin, (const float *)c->Eye, Hout, flags, n, kp, (int64_t)0, (float)pert, kp);
        HIPCHK(hipGetLastError());
        // did the Cholesky test accept it (lambda_min >= pert)?  one 4-byte read-back
        int flag0 = 0;
        HIPCHK(hipMemcpyAsync(&flag0, flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!flag0) return CMF_OK;
        if (psd && c->opt_ns && (kp == 256 || kp == 128)) {
            // positive semi-definite and clamped: spectral max M = max(H, pert I) by matrix polynomials (below), then the
            // same k_pad unit-vector solves on M
            CHK(ensure(c, c->nsidx, 4 * sizeof(int)));
            HIPCHK(hipMemsetAsync(c->nsidx.p, 0, 4 * sizeof(int), c->stream));
            float *M = nullptr;
            CHK(ns_clamp_images(c, Hin, (const int *)c->nsidx.p, 1, n, kp, pert, &M));
            const float *Mc = M;
            if (kp == 128) { // the matrix is the first diagonal block of a 256 x 256 image: compact it
                CHK(ensure(c, c->eigcopy, (size_t)kp * kp * sizeof(float)));
                HIPCHK(hipMemcpy2DAsync(c->eigcopy.p, kp * sizeof(float), M, 256 * sizeof(float), kp * sizeof(float), kp,
                                        hipMemcpyDeviceToDevice, c->stream));
                Mc = (const float *)c->eigcopy.p;
            }
            if (kp == 128) hipLaunchKernelGGL((chol_solve_kernel<8>), grid, block, 0, c->stream, Mc, (const float *)c->Eye, Hout, flags, n, kp, (int64_t)0, 0.0f, kp);
            else hipLaunchKernelGGL((chol_solve_kernel<16>), grid, block, 0, c->stream, Mc, (const float *)c->Eye, Hout, flags, n, kp, (int64_t)0, 0.0f, kp);
            HIPCHK(hipGetLastError());
            return CMF_OK;
        }
        if (n >= 48) return jacobi_chipwide(c, Hin, Hout, flags, n, kp, pert);
        HIPCHK(hipMemsetAsync(flags, 0xFF, sizeof(int), c->stream)); // small n: the one-workgroup kernel below
        need = (const int *)flags;
    } else if (nmat == 1 && n > 256 && !inplace) {
        // a single large matrix (n_components > 256 with the float64 treatment off or out of its range): the
        // one-workgroup kernels below would walk it out of L2 for seconds -- spread the sweep over the chip
        CHK(ensure(c, c->eigflag, sizeof(int)));
        return jacobi_chipwide(c, Hin, Hout, (int *)c->eigflag.p, n, kp, pert);
    } else if (c->opt_chol && tri_bytes <= 150 * 1024 && n <= 512) {
        CHK(ensure(c, c->eigflag, (size_t)nmat * sizeof(int)));
        const float *src = Hin;
        if (inplace) { // the fast path overwrites its output: keep the input for the Jacobi fallback
            CHK(ensure(c, c->eigcopy, (size_t)nmat * stride * sizeof(float)));
            HIPCHK(hipMemcpyAsync(c->eigcopy.p, Hin, (size_t)nmat * stride * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            src = (const float *)c->eigcopy.p;
        }
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&chol_safe_inverse_kernel), 150 * 1024));
        hipLaunchKernelGGL(chol_safe_inverse_kernel, dim3(nmat), dim3(256), tri_bytes, c->stream, src, Hout, (int *)c->eigflag.p, n, kp,
                           stride, (float)pert, nmat);
        HIPCHK(hipGetLastError());
        need = (const int *)c->eigflag.p;
        Hin = src;
    }
    CHK(clamp_stats(c, Hin, const_cast<int *>(need), nmat, n, kp, stride, pert, refine));
    const size_t lds_need = (size_t)(2 * n * n + n) * sizeof(float);
    if (lds_need <= 150 * 1024) {
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&jacobi_safe_inverse_kernel<true>), 150 * 1024));
        hipLaunchKernelGGL((jacobi_safe_inverse_kernel<true>), dim3(nmat), dim3(256), lds_need, c->stream, Hin, Hout,
                           (float *)nullptr, n, kp, stride, (float)pert, nmat, 30, need);
    } else {
        CHK(ensure(c, c->eigws, (size_t)nmat * 2 * stride * sizeof(float)));
        hipLaunchKernelGGL((jacobi_safe_inverse_kernel<false>), dim3(nmat), dim3(256), (size_t)n * sizeof(float), c->stream,
                           Hin, Hout, (float *)c->eigws.p, n, kp, stride, (float)pert, nmat, 30, need);
    }
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// C_b = alpha A_b B_b + beta D_b + gamma I for nb stacked 256 x 256 matrices (gemm_kernel ROLE 2)
static int gemm_blockdiag(cmf_ctx *c, const float *A, const float *B, float *C, const float *D, float alpha, float beta, float gamma,
                          int64_t nb) {
    using Cfg = GemmCfg<MODE_NN, 256>;
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.A = A; a.lda = 256; a.B = B; a.ldb = 256; a.b_batch = 256 * 256;
    a.C = C; a.ldc = 256; a.slab_stride = 0;
    a.Mout = nb * 256; a.Kred = 256; a.klen = 256;
    a.D = D; a.alpha = alpha; a.beta = beta; a.gamma = gamma;
    Timed tm(c, CMF_K_EIGEN, 2.0 * 256 * 256 * 256 * (double)nb);
    CHK(allow_big_lds(c, reinterpret_cast<const void *>(&gemm_kernel<MODE_NN, 256, 2, 4>), (int)Cfg::LDS_BYTES));
    hipLaunchKernelGGL((gemm_kernel<MODE_NN, 256, 2, 4>), dim3((unsigned)nb, 1, 1), dim3(512), Cfg::LDS_BYTES, c->stream, a);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// Flagged rows of a chunk (lambda_min(H) < pert), H positive semi-definite, k_pad = 256 (or 128: two per image):
//   safe_inverse(H) = max(H, pert I)^-1  with the spectral max  M = (H + pert I + |H - pert I|) / 2,
//   |B| = sign(B) B,  sign(B) by odd matrix polynomials of X0 = B / c, c >= rho(B) (Newton-Schulz family)
// -- nothing but 256^3 products (gemm_kernel ROLE 2, all flagged matrices of the chunk per launch).  An eigenvalue at
// distance delta from the threshold needs about log_3.44(c / delta) growth steps; the count below resolves
// delta = 1e-3 pert (closer ones keep an error <= delta in M, i.e. 1e-3 relative in that eigen-direction at worst:
// the clamp is continuous; measured 1e-6 .. 1e-4 on the solve).
// M >= pert I is then solved by the ordinary Cholesky kernel, which also clears the flag.
// M images of the nf matrices Hc[idx[b]] (see above); *M_out = nf / sub images of 256 x 256 floats
static int ns_clamp_images(cmf_ctx *c, const float *Hc, const int *idx, int nf, int n, int kp, double pert, float **M_out) {
    const int sub = 256 / kp;                       // matrices per 256 x 256 block-diagonal image (k_pad = 256: 1, 128: 2)
    const int64_t stride = (int64_t)kp * kp;        // of the input matrices
    const int64_t istride = 256 * 256;              // of the images
    const int64_t ni = (nf + sub - 1) / sub;        // images
    // workspaces: B, X, X', Y, Z (+ one spare image), then one word for max c
    CHK(ensure(c, c->nsws, (size_t)(5 * ni + 1) * istride * sizeof(float) + 16));
    float *Bm = (float *)c->nsws.p, *X = Bm + ni * istride, *X2 = X + ni * istride, *Y = X2 + ni * istride, *Z = Y + ni * istride;
    unsigned *cmax = (unsigned *)(Z + (ni + 1) * istride);
    HIPCHK(hipMemsetAsync(cmax, 0, sizeof(unsigned), c->stream));
    {
        Timed tm(c, CMF_K_EIGEN);
        hipLaunchKernelGGL(ns_prepare_kernel, dim3((unsigned)(ni * sub)), dim3(256), 0, c->stream, Hc, idx, Bm, X, n, kp, stride,
                           (float)pert, cmax, sub, nf);
        HIPCHK(hipGetLastError());
    }
    unsigned cbits = 0;
    HIPCHK(hipMemcpyAsync(&cbits, cmax, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float cm;
    memcpy(&cm, &cbits, sizeof cm);
    // growth phase: the odd quintic a x + b x^3 + c x^5 with (3.4445, -4.7750, 2.0315) multiplies small |x| by 3.44 per
    // step (3 products) and keeps [0, 1] inside about [0.7, 1.2]; six cubic steps then converge quadratically to +-1
    const double delta = 1e-3 * pert;
    int nq = (int)std::ceil(std::log(std::max((double)cm, pert) / delta) / std::log(3.4445));
    nq = std::min(std::max(nq, 4), 24);
    for (int it = 0; it < nq; ++it) {
        CHK(gemm_blockdiag(c, X, X, Y, nullptr, 1.0f, 0.f, 0.f, ni));            // Y = X^2
        CHK(gemm_blockdiag(c, Y, Y, Z, Y, 2.0315f, -4.7750f, 3.4445f, ni));      // Z = c Y^2 + b Y + a I
        CHK(gemm_blockdiag(c, X, Z, X2, nullptr, 1.0f, 0.f, 0.f, ni));           // X' = X Z
        std::swap(X, X2);
    }
    for (int it = 0; it < 6; ++it) {
        CHK(gemm_blockdiag(c, X, X, Y, nullptr, 1.0f, 0.f, 0.f, ni));            // Y = X^2
        CHK(gemm_blockdiag(c, X, Y, X2, X, -0.5f, 1.5f, 0.f, ni));               // X' = 1.5 X - 0.5 X Y
        std::swap(X, X2);
    }
    // M = (S B + B) / 2 + pert I   (H = B + pert I)
    CHK(gemm_blockdiag(c, X, Bm, Y, Bm, 0.5f, 0.5f, (float)pert, ni));
    *M_out = Y;
    return CMF_OK;
}

// step_b = grad_b * safe_inverse(H_b) for the matrices idx[0 .. nf) of a chunk (idx null: all of 0 .. nf) by the tridiagonal
// eigen-solve of cmf_eigclamp.hip.h (H is not modified).  Serves any symmetric H (|lambda| as
// pycmf/cmf_solvers.py:353), k_pad 128 / 256.  flags (nullable): cleared for every matrix served; one the QL iteration gave up on
// keeps its flag for the caller's fallback.
static bool eig_clamp_ok(const cmf_ctx *c, int n, int kp) { return (c->opt_eig_clamp & 1) && (kp == 256 || kp == 128) && n > 64; }
static int eig_clamp_solve(cmf_ctx *c, const float *Hc, const int *idx, int nf, const float *grad, float *step, int *flags, int n, int kp,
                           double pert, float *lam_out = nullptr, float *sens_out = nullptr) {
    if (nf <= 0) return CMF_OK;
    Timed tm(c, CMF_K_EIGEN);
    const int64_t stride = (int64_t)kp * kp;
    const int64_t cap = std::max<int64_t>(1024, (int64_t)n * n);       // PAIRS of sweep steps logged per wave of 64 matrices (measured: 0.3 .. 0.6 n^2)
    const int sw_cap = 8 * kp;                                          // sweeps per wave (measured: ~1.8 n)
    const int64_t per_mat = cap * (int64_t)sizeof(float4) + stride * (int64_t)sizeof(float);
    // scratch: reflectors (k_pad^2 floats) + rotation log (1 MB at n = 256) per matrix, kept under 12 GiB: a whole 8192-row chunk
    // of C3 is one batch
    const int bmax = (int)std::min<int64_t>(rup(nf, 64), std::max<int64_t>(64, ((int64_t)12 << 30) / per_mat / 64 * 64));
    const int64_t NB = bmax;
    CHK(ensure(c, c->eigcl_ws, eig_ws_floats(kp, NB) * sizeof(float)));
    CHK(ensure(c, c->eigcl_log, (size_t)NB * (size_t)per_mat + (size_t)(NB / 64) * sw_cap * sizeof(EigSweep)));
    CHK(ensure(c, c->eigcl_fail, (size_t)NB * sizeof(int)));
    float *ws = (float *)c->eigcl_ws.p;
    float *refl = (float *)c->eigcl_log.p;
    float4 *lg = (float4 *)(refl + (size_t)NB * stride);
    const size_t ql_lds = (size_t)2 * kp * 64 * sizeof(float);
    EigSweep *sw = (EigSweep *)(lg + (size_t)NB * cap);
    int *failf = (int *)c->eigcl_fail.p;
    static const bool want_stats = getenv("CMF_EIG_STATS") != nullptr; // (measurement: sweeps / trips per wave on stderr)
    long long *qstats = nullptr;
    if (want_stats) HIPCHK(hipMalloc((void **)&qstats, (size_t)(NB / 64) * 9 * sizeof(long long)));
    for (int b0 = 0; b0 < nf; b0 += bmax) {
        const int nb = std::min(bmax, nf - b0);
        const int *ib = idx ? idx + b0 : nullptr;
        const float *Hb = idx ? Hc : Hc + (int64_t)b0 * stride;
        const float *gb = idx ? grad : grad + (int64_t)b0 * kp;
        float *sb = idx ? step : step + (int64_t)b0 * kp;
        int *fb = (idx || !flags) ? flags : flags + b0;
        float *lam = lam_out ? lam_out + (int64_t)b0 * kp : (float *)nullptr;
        // (the QL kernel reads d, e, gt of all 64 lanes of its last wave: clear the slots past nb)
        if (nb % 64) HIPCHK(hipMemsetAsync(ws, 0, eig_ws_floats(kp, NB) * sizeof(float), c->stream));
        if (kp == 256) {
            CHK(allow_big_lds(c, reinterpret_cast<const void *>(&eig_ql_kernel<256>), (int)ql_lds));
            hipLaunchKernelGGL(eig_tridiag_kernel<256>, dim3((unsigned)nb), dim3(512), 0, c->stream, Hb, ib, gb, n, stride, ws, NB, refl);
            hipLaunchKernelGGL(eig_ql_kernel<256>, dim3((unsigned)((nb + 63) / 64)), dim3(64), ql_lds, c->stream, ws, NB, nb, n, (float)pert, lg, cap, sw, sw_cap, failf, lam, qstats, (c->opt_eig_clamp & 2) ? 0 : 1);
            hipLaunchKernelGGL(eig_backtransform_kernel<256>, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, c->stream, (const float *)refl, ib, nb, n,
                               (const float *)ws, NB, (const int *)failf, sb, fb, idx ? sens_out : (sens_out ? sens_out + b0 : (float *)nullptr));
        } else {
            CHK(allow_big_lds(c, reinterpret_cast<const void *>(&eig_ql_kernel<128>), (int)ql_lds));
            hipLaunchKernelGGL(eig_tridiag_kernel<128>, dim3((unsigned)nb), dim3(512), 0, c->stream, Hb, ib, gb, n, stride, ws, NB, refl);
            hipLaunchKernelGGL(eig_ql_kernel<128>, dim3((unsigned)((nb + 63) / 64)), dim3(64), ql_lds, c->stream, ws, NB, nb, n, (float)pert, lg, cap, sw, sw_cap, failf, lam, qstats, (c->opt_eig_clamp & 2) ? 0 : 1);
            hipLaunchKernelGGL(eig_backtransform_kernel<128>, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, c->stream, (const float *)refl, ib, nb, n,
                               (const float *)ws, NB, (const int *)failf, sb, fb, idx ? sens_out : (sens_out ? sens_out + b0 : (float *)nullptr));
        }
        HIPCHK(hipGetLastError());
        if (qstats) {
            const int nw = (nb + 63) / 64;
            std::vector<long long> hs((size_t)nw * 9);
            HIPCHK(hipMemcpyAsync(hs.data(), qstats, hs.size() * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            long long a = 0, t = 0, r = 0, tmax = 0, c1 = 0, c2 = 0, c3 = 0, n1 = 0, n2 = 0, ls = 0;
            for (int w = 0; w < nw; ++w) {
                const long long *q = hs.data() + 9 * w;
                a += q[0]; t += q[1]; r += q[2]; tmax = std::max(tmax, q[1]); c1 += q[3]; c2 += q[4]; c3 += q[5]; n1 += q[6]; n2 += q[7]; ls += q[8];
            }
            fprintf(stderr, "[cmf eig] %d matrices, %d waves: sweeps/wave %.1f, trips/wave %.0f (max %lld), own rotations/matrix %.0f; cycles/wave: QL %.0f, "
                            "forward %.0f, scale + backward %.0f; stopped early: %lld inside, %lld above, eigenvalues found per matrix %.1f\n",
                    nb, nw, (double)a / nw, (double)t / nw, tmax, (double)r / nb, (double)c1 / nw, (double)c2 / nw, (double)c3 / nw, n1, n2, (double)ls / nb);
        }
    }
    if (qstats) (void)hipFree(qstats);
    c->eig_clamp_rows += nf;
    return CMF_OK;
}

static int ns_clamp_solve_rows(cmf_ctx *c, const float *Hc, const float *grad, float *step, int *flags, int64_t nr, int n, int kp,
                               double pert) {
    CHK(ensure(c, c->nsidx, (size_t)(nr + 1) * sizeof(int)));
    int *idx = (int *)c->nsidx.p, *count = idx + nr;
    HIPCHK(hipMemsetAsync(count, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(compact_flags_kernel, dim3(32), dim3(256), 0, c->stream, (const int *)flags, (int)nr, idx, count);
    HIPCHK(hipGetLastError());
    int nf = 0;
    HIPCHK(hipMemcpyAsync(&nf, count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nf <= 0) return CMF_OK;
    float *M = nullptr;
    CHK(ns_clamp_images(c, Hc, (const int *)idx, nf, n, kp, pert, &M));
    Timed tm(c, CMF_K_EIGEN);
    const dim3 grid((unsigned)nf), block(256);
    const int64_t istride = 256 * 256;
    if (kp == 256 && c->opt_chol_mfma) {
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&chol_solve_mfma_kernel), (int)CholMfma::LDS_BYTES));
        hipLaunchKernelGGL(chol_solve_mfma_kernel, grid, block, CholMfma::LDS_BYTES, c->stream, (const float *)M, grad, step, flags, n, kp, istride, 0.0f, nf,
                           (const int *)idx, (const int *)nullptr, 1, 0, (float *)nullptr);
    } else if (kp == 256)
        hipLaunchKernelGGL((chol_solve_kernel<16>), grid, block, 0, c->stream, (const float *)M, grad, step, flags, n, kp, istride, 0.0f, nf, 0,
                           (const int *)idx, 1);
    else
        hipLaunchKernelGGL((chol_solve_kernel<8>), grid, block, 0, c->stream, (const float *)M, grad, step, flags, n, kp, istride, 0.0f, nf, 0,
                           (const int *)idx, 2);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// The flagged matrices of a chunk whose spectrum is "one eigenvalue above the threshold, the rest below" (cmf_rank1clamp.hip.h): power
// iteration, Cholesky certificate, closed-form step; a row that does not pass keeps its flag for the eigen-solve below.  Every decision
// is per row (a matrix without a dominant eigenvalue leaves the power iteration after one sweep; C3X: 0.4 ms per 8192 rows), so the
// route a row takes does not depend on the chunk or the rank it is served in; a chunk in which no row converged skips the certificate.
static bool rank1_clamp_ok(const cmf_ctx *c, int n, int kp) {
    return c->opt_rank1_clamp && c->hess_psd && kp == 256 && n > 128 && c->opt_chol_mfma && !c->opt_choldiag && eig_clamp_ok(c, n, kp);
}
static int rank1_clamp_solve_rows(cmf_ctx *c, const float *Hc, const float *grad, float *step, int *flags, int64_t nr, int n, int kp, double pert,
                                  float *sens_out) {
    CHK(ensure(c, c->nsidx, (size_t)(nr + 1) * sizeof(int)));
    int *idx = (int *)c->nsidx.p, *count = idx + nr;
    HIPCHK(hipMemsetAsync(count, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(compact_flags_kernel, dim3(32), dim3(256), 0, c->stream, (const int *)flags, (int)nr, idx, count);
    HIPCHK(hipGetLastError());
    int nf = 0;
    HIPCHK(hipMemcpyAsync(&nf, count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nf <= 0) return CMF_OK;
    Timed tm(c, CMF_K_EIGEN);
    const int64_t stride = (int64_t)kp * kp;
    CHK(ensure(c, c->eigcl_log, (size_t)nf * stride * sizeof(float)));                               // the certificate images
    CHK(ensure(c, c->r1_ws, (size_t)nf * (kp + 3) * sizeof(float) + 16));                             // q | lambda | ok | cert | served (8-byte aligned)
    float *A = (float *)c->eigcl_log.p;
    float *q = (float *)c->r1_ws.p, *lam = q + (size_t)nf * kp;
    int *ok = (int *)(lam + nf), *cert = ok + nf;
    unsigned long long *served = (unsigned long long *)(((uintptr_t)(cert + nf) + 7) & ~(uintptr_t)7);
    HIPCHK(hipMemsetAsync(cert, 1, (size_t)nf * sizeof(int), c->stream)); // non-zero: not certified
    HIPCHK(hipMemsetAsync(served, 0, sizeof(unsigned long long), c->stream));
    unsigned long long ns = 0;
    const int *ib = nf == nr ? (const int *)nullptr : (const int *)idx;
    // three products: the start (a normalised row of H) is one step already, two more converge when lambda_2 / lambda_1 < 1e-2, the third measures
    hipLaunchKernelGGL(rank1_power_kernel, dim3((unsigned)nf), dim3(256), 0, c->stream, Hc, ib, n, kp, stride, (float)pert, 3, 1.0e-5f, A, q, lam, ok, served);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&ns, served, sizeof(ns), hipMemcpyDeviceToHost, c->stream));   // how many converged
    HIPCHK(hipStreamSynchronize(c->stream));
    if (ns == 0) return CMF_OK;
    HIPCHK(hipMemsetAsync(served, 0, sizeof(unsigned long long), c->stream));
    CHK(allow_big_lds(c, reinterpret_cast<const void *>(&chol_solve_mfma_kernel), (int)CholMfma::LDS_BYTES));
    hipLaunchKernelGGL(chol_solve_mfma_kernel, dim3((unsigned)nf), dim3(256), CholMfma::LDS_BYTES, c->stream, (const float *)A, (const float *)nullptr,
                       (float *)nullptr, cert, n, kp, stride, 0.0f, nf, (const int *)nullptr, (const int *)nullptr, 1, 0, (float *)nullptr);
    hipLaunchKernelGGL(rank1_compose_kernel, dim3((unsigned)((nf + 3) / 4)), dim3(256), 0, c->stream, ib, nf, n, kp, (float)pert, (const float *)q,
                       (const float *)lam, (const int *)ok, (const int *)cert, grad, step, flags, sens_out, served);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&ns, served, sizeof(ns), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->rank1_rows += (int64_t)ns;
    return CMF_OK;
}

// the flagged matrices of a chunk through eig_clamp_solve (compacted list; one 4-byte read-back)
static int eig_clamp_solve_rows(cmf_ctx *c, const float *Hc, const float *grad, float *step, int *flags, int64_t nr, int n, int kp, double pert,
                                float *sens_out = nullptr) {
    CHK(ensure(c, c->nsidx, (size_t)(nr + 1) * sizeof(int)));
    int *idx = (int *)c->nsidx.p, *count = idx + nr;
    HIPCHK(hipMemsetAsync(count, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(compact_flags_kernel, dim3(32), dim3(256), 0, c->stream, (const int *)flags, (int)nr, idx, count);
    HIPCHK(hipGetLastError());
    int nf = 0;
    HIPCHK(hipMemcpyAsync(&nf, count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (nf <= 0) return CMF_OK;
    if (nf == nr) return eig_clamp_solve(c, Hc, nullptr, nf, grad, step, flags, n, kp, pert, nullptr, sens_out); // every row: no indirection
    return eig_clamp_solve(c, Hc, (const int *)idx, nf, grad, step, flags, n, kp, pert, nullptr, sens_out);
}

// step_i = grad_i * safe_inverse(H_i) for a chunk of per-row Hessians (H is clobbered):
// register-resident Cholesky solve for the rows with lambda_min >= pert, Jacobi + row product
// for the flagged rest.
struct RowCert { // per-group positive-definiteness certificates of a chunk (fused_rows_finish), or none
    const int *flags = nullptr; // [2 * groups]: 0 = the shared part alone already exceeds the threshold
    int rows = 1, split = 0;    // rows per group, rows in its first half
};
static int safe_solve_rows(cmf_ctx *c, float *Hc, const float *grad, float *step, int64_t nr, int n, int kp, double pert,
                           const RowCert &cert = RowCert(), bool refine = false, const float *Frows = nullptr) {
    if (nr <= 0) return CMF_OK;
    const int64_t stride = (int64_t)kp * kp;
    if (!c->opt_chol || n > 256) { // general path only
        CHK(safe_inverse_dev(c, Hc, Hc, (int)nr, n, kp, pert, false, refine));
        Timed tm(c, CMF_K_ELEMWISE);
        hipLaunchKernelGGL(rowvec_mat_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, c->stream, step, grad, (const float *)Hc, nr, kp, n);
        HIPCHK(hipGetLastError());
        return CMF_OK;
    }
    CHK(ensure(c, c->eigflag, (size_t)nr * (sizeof(int) + sizeof(float))));
    int *flags = (int *)c->eigflag.p;
    float *condest = (refine && c->opt_refine && c->hess_psd) ? (float *)(flags + nr) : nullptr; // per-row condition estimates of the plain solves
    {
        Timed tm(c, CMF_K_EIGEN);
        const dim3 grid((unsigned)nr), block(256);
        if (n <= 32) hipLaunchKernelGGL((chol_solve_kernel<2>), grid, block, 0, c->stream, (const float *)Hc, grad, step, flags, n, kp, stride, (float)pert, (int)nr, c->opt_choldiag,
                                         (const int *)nullptr, 1, cert.flags, cert.rows, cert.split, condest);
        else if (n <= 64) hipLaunchKernelGGL((chol_solve_kernel<4>), grid, block, 0, c->stream, (const float *)Hc, grad, step, flags, n, kp, stride, (float)pert, (int)nr, c->opt_choldiag,
                                              (const int *)nullptr, 1, cert.flags, cert.rows, cert.split, condest);
        else if (n <= 128) hipLaunchKernelGGL((chol_solve_kernel<8>), grid, block, 0, c->stream, (const float *)Hc, grad, step, flags, n, kp, stride, (float)pert, (int)nr, c->opt_choldiag,
                                               (const int *)nullptr, 1, cert.flags, cert.rows, cert.split, condest);
        else if (kp == 256 && c->opt_chol_mfma && !c->opt_choldiag) {
            // 128 < n <= 256: blocked Cholesky on the matrix pipe (cmf_chol_mfma.hip.h), two barriers per 32-column panel
            CHK(allow_big_lds(c, reinterpret_cast<const void *>(&chol_solve_mfma_kernel), (int)CholMfma::LDS_BYTES));
            hipLaunchKernelGGL(chol_solve_mfma_kernel, grid, block, CholMfma::LDS_BYTES, c->stream, (const float *)Hc, grad, step, flags, n, kp, stride, (float)pert, (int)nr,
                               (const int *)nullptr, cert.flags, cert.rows, cert.split, condest);
        }
        else hipLaunchKernelGGL((chol_solve_kernel<16>), grid, block, 0, c->stream, (const float *)Hc, grad, step, flags, n, kp, stride, (float)pert, (int)nr, c->opt_choldiag,
                                (const int *)nullptr, 1, cert.flags, cert.rows, cert.split, condest);
        HIPCHK(hipGetLastError());
        // Round 6: where the tridiagonal eigen-solve serves the flagged matrices (cheap: every one of them gets its float32 step)
        // the float64 refinement is decided AFTERWARDS, per row, by the error bound of that step (clamp_stats); otherwise (round 5)
        // by the ratio alone, before the clamp path, whose work on the listed matrices would be thrown away.
        const bool bound_first = refine && Frows && c->opt_refine && c->hess_psd && c->opt_refine_tol > 0.0 && eig_clamp_ok(c, n, kp);
        if (bound_first) {
            CHK(ensure(c, c->eigcl_snap, (size_t)nr * sizeof(int)));
            HIPCHK(hipMemcpyAsync(c->eigcl_snap.p, flags, (size_t)nr * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
            if (rank1_clamp_ok(c, n, kp)) CHK(rank1_clamp_solve_rows(c, Hc, grad, step, flags, nr, n, kp, pert, condest));
            CHK(eig_clamp_solve_rows(c, Hc, grad, step, flags, nr, n, kp, pert, condest)); // (condest of a clamped row <- the solve's sensitivity)
            CHK(clamp_stats(c, Hc, (int *)c->eigcl_snap.p, nr, n, kp, stride, pert, refine, condest, step, Frows));
        } else {
            CHK(clamp_stats(c, Hc, flags, nr, n, kp, stride, pert, refine, condest, step));
            if (rank1_clamp_ok(c, n, kp)) CHK(rank1_clamp_solve_rows(c, Hc, grad, step, flags, nr, n, kp, pert, nullptr));
            if (eig_clamp_ok(c, n, kp)) CHK(eig_clamp_solve_rows(c, Hc, grad, step, flags, nr, n, kp, pert));
        }
        // flagged matrices the eigen-solve did not serve (option, or its iteration gave up), k_pad = 128 / 256, Hessians positive
        // semi-definite by construction (weights >= 0): spectral clamp by Newton-Schulz (MFMA) + a second Cholesky solve
        if ((kp == 256 || kp == 128) && c->opt_ns && c->hess_psd) CHK(ns_clamp_solve_rows(c, Hc, grad, step, flags, nr, n, kp, pert));
        // whatever is still flagged: |lambda| / clamp by Jacobi, in place (the solve kernel does not modify H)
        const size_t lds_need = (size_t)(2 * n * n + n) * sizeof(float);
        if (lds_need <= 150 * 1024) {
            CHK(allow_big_lds(c, reinterpret_cast<const void *>(&jacobi_safe_inverse_kernel<true>), 150 * 1024));
            hipLaunchKernelGGL((jacobi_safe_inverse_kernel<true>), dim3((unsigned)nr), dim3(256), lds_need, c->stream, (const float *)Hc, Hc,
                               (float *)nullptr, n, kp, stride, (float)pert, (int)nr, 30, (const int *)flags);
        } else {
            CHK(ensure(c, c->eigws, (size_t)nr * 2 * stride * sizeof(float)));
            hipLaunchKernelGGL((jacobi_safe_inverse_kernel<false>), dim3((unsigned)nr), dim3(256), (size_t)n * sizeof(float), c->stream,
                               (const float *)Hc, Hc, (float *)c->eigws.p, n, kp, stride, (float)pert, (int)nr, 30, (const int *)flags);
        }
        HIPCHK(hipGetLastError());
    }
    Timed tm(c, CMF_K_ELEMWISE);
    hipLaunchKernelGGL(rowvec_mat_flagged_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, c->stream, step, grad, (const float *)Hc,
                       (const int *)flags, nr, kp, n);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

extern "C" int cmf_safe_invert_batch(cmf_ctx *c, const double *H, double *out, int n, int k, double pert) {
    if (!c || !H || !out || n < 0 || k <= 0) return fail(CMF_EINVAL, "bad argument");
    DeviceGuard dg(c->device);
    const int kp = pad_k(k);
    const size_t elems = (size_t)n * kp * kp;
    std::vector<float> host(elems, 0.f);
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < k; ++r)
            for (int q = 0; q < k; ++q) host[((size_t)i * kp + r) * kp + q] = (float)H[((size_t)i * k + r) * k + q];
    float *dH = nullptr;
    HIPCHK(hipMalloc((void **)&dH, std::max<size_t>(elems, 4) * sizeof(float)));
    int rc = CMF_OK;
    do {
        if (hipMemcpyAsync(dH, host.data(), elems * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(CMF_EHIP, "H2D failed"); break; }
        rc = safe_inverse_dev(c, dH, dH, n, k, kp, pert);
        if (rc != CMF_OK) break;
        if (hipMemcpyAsync(host.data(), dH, elems * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = fail(CMF_EHIP, "D2H failed"); break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(CMF_EHIP, "sync failed"); break; }
    } while (0);
    (void)hipFree(dH);
    if (rc != CMF_OK) return rc;
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < k; ++r)
            for (int q = 0; q < k; ++q) out[((size_t)i * k + r) * k + q] = (double)host[((size_t)i * kp + r) * kp + q];
    return CMF_OK;
}

// ---- shared-Hessian building blocks ------------------------------------------------------
// float64 Gram of a factor (cmf_shared64.hip.h): G64 = F^T F, optional float32 copy for the F G product
extern "C" int cmf_safe_solve_batch(cmf_ctx *c, const double *H, const double *g, double *out, double *lam, int n, int k, double pert, int method) {
    if (!c || !H || !g || !out || n < 0 || k <= 0 || method < 0 || method > 1) return fail(CMF_EINVAL, "bad argument");
    DeviceGuard dg(c->device);
    const int kp = pad_k(k);
    if (method == 1 && !((kp == 128 || kp == 256) && k > 64)) return fail(CMF_EINVAL, "the eigen-solve serves 64 < k <= 256");
    if (n == 0) return CMF_OK;
    const size_t me = (size_t)n * kp * kp, ve = (size_t)n * kp;
    std::vector<float> hm(me, 0.f), hv(3 * ve, 0.f);
    for (int i = 0; i < n; ++i) {
        for (int r = 0; r < k; ++r)
            for (int q = 0; q < k; ++q) hm[((size_t)i * kp + r) * kp + q] = (float)H[((size_t)i * k + r) * k + q];
        for (int q = 0; q < k; ++q) hv[(size_t)i * kp + q] = (float)g[(size_t)i * k + q];
    }
    float *dH = nullptr, *dv = nullptr;
    HIPCHK(hipMalloc((void **)&dH, me * sizeof(float)));
    if (hipMalloc((void **)&dv, 3 * ve * sizeof(float)) != hipSuccess) { (void)hipFree(dH); return fail(CMF_ENOMEM, "hipMalloc failed"); }
    float *dgr = dv, *dstep = dv + ve, *dlam = dv + 2 * ve;
    int rc = CMF_OK;
    do {
        if (hipMemcpyAsync(dH, hm.data(), me * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(dv, hv.data(), 3 * ve * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(CMF_EHIP, "H2D failed"); break; }
        const int save_refine = c->opt_refine;
        c->opt_refine = 0; // (no factors / data behind these matrices: nothing the float64 refinement could recompute)
        if (method == 1) rc = eig_clamp_solve(c, dH, nullptr, n, dgr, dstep, nullptr, k, kp, pert, dlam);
        else rc = safe_solve_rows(c, dH, dgr, dstep, n, k, kp, pert);
        c->opt_refine = save_refine;
        if (rc != CMF_OK) break;
        if (hipMemcpyAsync(hv.data(), dv, 3 * ve * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = fail(CMF_EHIP, "D2H failed"); break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(CMF_EHIP, "sync failed"); break; }
    } while (0);
    (void)hipFree(dH);
    (void)hipFree(dv);
    if (rc != CMF_OK) return rc;
    for (int i = 0; i < n; ++i)
        for (int q = 0; q < k; ++q) {
            out[(size_t)i * k + q] = (double)hv[ve + (size_t)i * kp + q];
            if (lam) lam[(size_t)i * k + q] = (double)hv[2 * ve + (size_t)i * kp + q];
        }
    return CMF_OK;
}

static int gram64(cmf_ctx *c, const float *F, int64_t rows_pad, double *G64, float *G32) {
    const int kp = c->kp;
    const int ts = kp >= 64 ? 64 : 32, T = kp / ts, ntile = T * (T + 1) / 2;
    int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>((4 * c->num_cu + ntile - 1) / ntile, rows_pad / 32));
    const int64_t chunk = rup((rows_pad + nsplit - 1) / nsplit, 32);
    nsplit = (rows_pad + chunk - 1) / chunk;
    CHK(ensure(c, c->gslab64, (size_t)nsplit * ntile * ts * ts * sizeof(double)));
    Timed tm(c, CMF_K_GEMM_SMALL, 2.0 * (double)rows_pad * kp * kp);
    const dim3 grid((unsigned)ntile, (unsigned)nsplit);
    if (ts == 64) hipLaunchKernelGGL((gram64_partial_kernel<64>), grid, dim3(256), 0, c->stream, F, kp, rows_pad, chunk, (double *)c->gslab64.p);
    else hipLaunchKernelGGL((gram64_partial_kernel<32>), grid, dim3(64), 0, c->stream, F, kp, rows_pad, chunk, (double *)c->gslab64.p);
    hipLaunchKernelGGL(gram64_reduce_kernel, dim3((unsigned)std::min(256, (kp * kp + 255) / 256)), dim3(256), 0, c->stream,
                       (const double *)c->gslab64.p, ts, kp, (int)nsplit, G64, G32);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

static int gemm64(cmf_ctx *c, bool trans_b, const double *A, const double *B, double *C, const double *D, double alpha, double beta,
                  double gamma, float *C32 = nullptr, int nvalid = 0) {
    const dim3 grid((unsigned)(c->kp / 32), (unsigned)(c->kp / 32));
    if (trans_b) hipLaunchKernelGGL((gemm64_kernel<true>), grid, dim3(256), 0, c->stream, A, B, C, D, alpha, beta, gamma, c->kp, C32, nvalid, (const int *)nullptr, 0);
    else hipLaunchKernelGGL((gemm64_kernel<false>), grid, dim3(256), 0, c->stream, A, B, C, D, alpha, beta, gamma, c->kp, C32, nvalid, (const int *)nullptr, 0);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// c->Hinv (float32, zero on the padding) = safe_inverse(H64) for the one shared Hessian, formed and inverted in float64
// (reference: _safe_invert, cmf_solvers.py:346-356, on a float64 matrix)
static int shared_inverse64(cmf_ctx *c, const double *H64, int n, double pert, bool psd, bool *plain = nullptr) {
    const int kp = c->kp;
    CHK(ensure(c, c->hinv64, (size_t)kp * kp * sizeof(double)));
    if (plain) *plain = false; // set when the host knows that lambda_min >= pert, i.e. the result is the plain inverse
    Timed tm(c, CMF_K_EIGEN);
    if (n <= 64) { // one launch, the branch on lambda_min is taken on the device
        const size_t lds = (size_t)(2 * n * n + n) * sizeof(double);
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&safe_inverse64_small_kernel), 80 * 1024));
        hipLaunchKernelGGL(safe_inverse64_small_kernel, dim3(1), dim3(256), lds, c->stream, H64, c->Hinv, n, kp, pert, (double *)c->hinv64.p);
        HIPCHK(hipGetLastError());
        return CMF_OK;
    }
    const size_t kk = (size_t)kp * kp;
    // workspaces: W0, W1 (Cholesky of H - pert I and of H), Xt (L^-1 transposed), then 2 flags
    CHK(ensure(c, c->w64, 3 * kk * sizeof(double) + 16));
    double *W0 = (double *)c->w64.p, *W1 = W0 + kk, *Xt = W1 + kk;
    int *flags = (int *)(Xt + kk);
    // rows of L per LDS panel of the triangular inverse: Y is n x 16, the panel rp x (n + 2) doubles, 150 KB in all
    const int rp = (int)std::max<int64_t>(1, std::min<int64_t>(32, ((int64_t)150 * 1024 - 128 * (int64_t)n) / (8 * ((int64_t)n + 2))));
    const size_t tri_lds = ((size_t)16 * n + (size_t)rp * (n + 2)) * sizeof(double);
    auto inverse_from = [&](const double *L) -> int { // Hinv = L^-T L^-1
        if (kp <= 128) {
            hipLaunchKernelGGL((tri_inverse64_reg_kernel<8>), dim3((unsigned)(kp / 16)), dim3(256), 0, c->stream, L, n, kp, Xt, kp, kp);
        } else if (kp <= 256) {
            hipLaunchKernelGGL((tri_inverse64_reg_kernel<16>), dim3((unsigned)(kp / 16)), dim3(256), 0, c->stream, L, n, kp, Xt, kp, kp);
        } else {
            CHK(allow_big_lds(c, reinterpret_cast<const void *>(&tri_inverse64_kernel), 152 * 1024));
            hipLaunchKernelGGL(tri_inverse64_kernel, dim3((unsigned)(kp / 16)), dim3(256), tri_lds, c->stream, L, n, kp, Xt, kp, kp, rp);
        }
        HIPCHK(hipGetLastError());
        // float64 image for the re-associated sweeps (identity on the padding), float32 rounding for everything else
        CHK(gemm64(c, true, Xt, Xt, (double *)c->hinv64.p, nullptr, 1.0, 0.0, 0.0, c->Hinv, n));
        hipLaunchKernelGGL(pad_identity64_kernel, dim3((unsigned)std::min(256, (kp * kp + 255) / 256)), dim3(256), 0, c->stream, (double *)c->hinv64.p, kp, n);
        HIPCHK(hipGetLastError());
        return CMF_OK;
    };
    auto launch_chol = [&](const double *Hm, int nwg, double sh0, double sh1) {
        if (n <= 128) hipLaunchKernelGGL((chol64_reg_kernel<4>), dim3(nwg), dim3(1024), 0, c->stream, Hm, n, kp, W0, (int64_t)kk, kp, sh0, sh1, flags);
        else if (n <= 256) hipLaunchKernelGGL((chol64_reg_kernel<8>), dim3(nwg), dim3(1024), 0, c->stream, Hm, n, kp, W0, (int64_t)kk, kp, sh0, sh1, flags);
        else hipLaunchKernelGGL(chol64_kernel, dim3(nwg), dim3(1024), 0, c->stream, Hm, n, kp, W0, (int64_t)kk, kp, sh0, sh1, flags);
    };
    launch_chol(H64, 2, pert, 0.0);
    HIPCHK(hipGetLastError());
    CHK(ensure(c, c->ns64, 6 * kk * sizeof(double) + 16));
    double *Bm = (double *)c->ns64.p, *X = Bm + kk, *X2 = X + kk, *Y = X2 + kk, *Z = Y + kk, *M = Z + kk;
    double *cnorm = M + kk;
    int hflags[2] = {0, 0};
    double hc = 1.0;
    HIPCHK(hipMemcpyAsync(hflags, flags, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (!hflags[0] && !hflags[1]) { // lambda_min >= pert: the clamp is the identity
        if (plain) *plain = true;
        return inverse_from(W1);
    }
    if (psd) {
        hipLaunchKernelGGL(ns64_prepare_kernel, dim3(1), dim3(1024), 0, c->stream, H64, Bm, X, n, kp, pert, cnorm);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&hc, cnorm, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        // M = max(H, pert I) = (sign(B) B + B) / 2 + pert I, sign(B) by odd polynomials of X0 = B / c (see ns_clamp_images);
        // float64 resolves eigenvalues down to 1e-6 pert from the threshold
        const double delta = 1e-6 * pert;
        int nq = (int)std::ceil(std::log(std::max(hc, pert) / delta) / std::log(3.4445));
        nq = std::min(std::max(nq, 4), 48);
        for (int it = 0; it < nq; ++it) {
            CHK(gemm64(c, false, X, X, Y, nullptr, 1.0, 0.0, 0.0));
            CHK(gemm64(c, false, Y, Y, Z, Y, 2.0315, -4.7750, 3.4445));
            CHK(gemm64(c, false, X, Z, X2, nullptr, 1.0, 0.0, 0.0));
            std::swap(X, X2);
        }
        for (int it = 0; it < 8; ++it) {
            CHK(gemm64(c, false, X, X, Y, nullptr, 1.0, 0.0, 0.0));
            CHK(gemm64(c, false, X, Y, X2, X, -0.5, 1.5, 0.0));
            std::swap(X, X2);
        }
        CHK(gemm64(c, false, X, Bm, M, Bm, 0.5, 0.5, pert));
        launch_chol((const double *)M, 1, 0.0, 0.0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(hflags, flags, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!hflags[0]) return inverse_from(W0);
    }
    return CMF_EUNSUPPORTED; // caller falls back to the float32 eigen-solver
}

// F <- clamp(F - grad * safe_inverse(H)); grad lives in c->den, scratch in c->num.  H: c->h64 (float64) when the
// float64 treatment is on, else c->Hm (float32).
static int shared_inverse(cmf_ctx *c, double pert, bool *plain = nullptr) {
    if (plain) *plain = false;
    bool done = false;
    if (c->opt_shared64 && c->kp <= 1024) {
        const int rc = shared_inverse64(c, (const double *)c->h64.p, c->k, pert, c->hess_psd, plain);
        if (rc == CMF_OK) done = true;
        else if (rc != CMF_EUNSUPPORTED) return rc;
        else { // not positive semi-definite by construction (alpha outside [0, 1]): |lambda| route of the float32 eigen-solver
            Timed tm(c, CMF_K_ELEMWISE);
            hipLaunchKernelGGL(axpby64_to_f32_kernel, dim3(256), dim3(256), 0, c->stream, c->Hm, (const double *)c->h64.p, 1.0,
                               (const double *)nullptr, 0.0, (int64_t)c->kp * c->kp);
            HIPCHK(hipGetLastError());
        }
    }
    if (!done) {
        CHK(safe_inverse_dev(c, c->Hm, c->Hinv, 1, c->k, c->kp, pert, c->hess_psd));
        if (c->opt_shared64 && c->kp <= 1024) { // float32 eigen-solver route: promote its result for the re-associated sweeps
            const int64_t kk = (int64_t)c->kp * c->kp;
            CHK(ensure(c, c->hinv64, (size_t)kk * sizeof(double)));
            Timed tm(c, CMF_K_ELEMWISE);
            hipLaunchKernelGGL(f32_to_f64_kernel, dim3(256), dim3(256), 0, c->stream, (double *)c->hinv64.p, (const float *)c->Hinv, kk);
            hipLaunchKernelGGL(pad_identity64_kernel, dim3(256), dim3(256), 0, c->stream, (double *)c->hinv64.p, c->kp, c->k);
            HIPCHK(hipGetLastError());
        }
    }
    return CMF_OK;
}

// ---- re-associated shared sweep ("pre-conditioned operand") -------------------------------------------------------
// The reference's shared-Hessian sweep  F <- clamp(F - grad Hinv),  grad = s (F G - T O) + l1 sign F + l2 F,  H = s G + l2 I
// (pycmf/cmf_solvers.py:396-410, :436-450, :321-326) is evaluated as
//     F <- clamp( F E  +  T (s O Hinv)  -  l1 sign(F) Hinv ),      E = I - H Hinv  (zero unless _safe_invert clamped)
// which is the same point in exact arithmetic.  In floating point it is not the same: grad is the small difference of two large
// products and Hinv multiplies its rounding error by up to cond(H); here O Hinv and E are formed in float64 from the float64
// Hessian (k x k work), and the float32 data contraction T (O') is the LAST operation, so its rounding error reaches the factor
// unamplified.  Measured on the clamped non-negative case of tests/test_gpu_shared64.py (cond 1e4): residual distance to the
// float64 CPU reference after 8 iterations 1.7e-3 -> 5e-7 (tests/tools/emul_newton_precision.py reproduces both on the CPU).
static bool use_reassoc(const cmf_ctx *c) { return c->opt_reassoc && c->opt_shared64 && c->kp <= 1024; }

// out[rows_pad x k_pad] = scale * O Hinv64  (float64 matrix pipe, one rounding)
static int factor_times_hinv(cmf_ctx *c, const float *O, int64_t rows_pad, double scale, float *out) {
    Timed tm(c, CMF_K_GEMM_SMALL, 2.0 * (double)rows_pad * c->kp * c->kp);
    if (c->kp == 32) hipLaunchKernelGGL((factor_times64_kernel<32>), dim3(1, (unsigned)(rows_pad / 64)), dim3(256), 0, c->stream, O, (const double *)c->hinv64.p, out, c->kp, scale);
    else if (c->kp >= 64 && c->opt_ft_tile == 256 && rows_pad % 128 == 0) hipLaunchKernelGGL((factor_times64_kernel<64, 128>), dim3((unsigned)(c->kp / 64), (unsigned)(rows_pad / 128)), dim3(512), 0, c->stream, O, (const double *)c->hinv64.p, out, c->kp, scale);
    else if (c->kp >= 128 && c->opt_ft_tile == 128) hipLaunchKernelGGL((factor_times64_kernel<128>), dim3((unsigned)(c->kp / 128), (unsigned)(rows_pad / 64)), dim3(256), 0, c->stream, O, (const double *)c->hinv64.p, out, c->kp, scale);
    else hipLaunchKernelGGL((factor_times64_kernel<64>), dim3((unsigned)(c->kp / 64), (unsigned)(rows_pad / 64)), dim3(256), 0, c->stream, O, (const double *)c->hinv64.p, out, c->kp, scale);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}
// c->Hm (float32) = E = I - H64 Hinv64 on the valid block, zero on the padding
static int clamp_defect(cmf_ctx *c) {
    Timed tm(c, CMF_K_EIGEN);
    return gemm64(c, false, (const double *)c->h64.p, (const double *)c->hinv64.p, nullptr, nullptr, -1.0, 0.0, 1.0, c->Hm, c->k);
}
// F <- clamp(F E + P - l1 sign(F) Hinv);  P (rows_pad x k_pad, clobbered) = T (s O Hinv);  `plain`: E = 0 is known on the host
static int reassoc_finish(cmf_ctx *c, int which, float *P, double l1, bool plain, bool nn) {
    const int64_t rows = c->frows_pad[which];
    float *F = c->F[which];
    if (l1 != 0.0) { // P += sign(F) (-l1 Hinv)
        float *sg = (P == c->den) ? c->num : c->den;
        CHK(launch_ew(c, sign_kernel, rows * c->kp / 4, sg, (const float *)F, rows * c->kp / 4));
        CHK(launch_ew(c, axpby_kernel, (int64_t)c->kp * c->kp, c->G, (const float *)c->Hinv, (float)-l1, (const float *)nullptr, 0.f, (int64_t)c->kp * c->kp));
        CHK(gemm(c, MODE_NN, sg, c->kp, c->G, c->kp, P, rows, c->kp, c->kp, true));
    }
    if (plain)
        return launch_ew(c, combine_clamp_kernel, rows * c->kp / 4, F, (const float *)nullptr, (const float *)P, 1.0f, c->frows[which], c->kp, c->k,
                         rows * c->kp / 4, nn ? 1 : 0);
    if (c->opt_fused_mu && c->kp <= 256) {
        Epilogue e;
        e.kind = EPI_COMBINE; e.F = F; e.P = P; e.out = F; e.a = 1.0; e.rows = c->frows[which]; e.kvalid = c->k; e.nn = nn ? 1 : 0;
        if (small_tile_ok(c, rows)) return factor_update(c, F, c->Hm, e, rows);
        return gemm(c, MODE_NN, F, c->kp, c->Hm, c->kp, F, rows, c->kp, c->kp, false, &e);
    }
    float *acc = (P == c->den) ? c->num : c->den;
    CHK(gemm(c, MODE_NN, F, c->kp, c->Hm, c->kp, acc, rows, c->kp, c->kp));
    return launch_ew(c, combine_clamp_kernel, rows * c->kp / 4, F, (const float *)acc, (const float *)P, 1.0f, c->frows[which], c->kp, c->k,
                     rows * c->kp / 4, nn ? 1 : 0);
}
static int shared_apply(cmf_ctx *c, int which, bool non_negative) {
    const int64_t rows = c->frows_pad[which];
    if (c->opt_fused_mu && c->kp <= 256) { // F <- clamp(F - grad H^-1) in the epilogue of the step product
        Epilogue e;
        e.kind = EPI_APPLY; e.F = c->F[which]; e.out = c->F[which]; e.rows = c->frows[which]; e.kvalid = c->k; e.nn = non_negative ? 1 : 0;
        if (small_tile_ok(c, rows)) return factor_update(c, c->den, c->Hinv, e, rows);
        return gemm(c, MODE_NN, c->den, c->kp, c->Hinv, c->kp, c->num, rows, c->kp, c->kp, false, &e);
    }
    CHK(gemm(c, MODE_NN, c->den, c->kp, c->Hinv, c->kp, c->num, rows, c->kp, c->kp));
    return launch_ew(c, newton_apply_kernel, rows * c->kp, c->F[which], (const float *)c->num, c->frows[which], c->kp, c->k,
                     rows * c->kp, non_negative ? 1 : 0);
}
static int shared_step(cmf_ctx *c, int which, double pert, bool non_negative) {
    CHK(shared_inverse(c, pert));
    return shared_apply(c, which, non_negative);
}

static bool use_shared64(const cmf_ctx *c) { return c->opt_shared64 && c->kp <= 1024; }
static int ensure_shared64(cmf_ctx *c) {
    const size_t kk = (size_t)c->kp * c->kp * sizeof(double);
    CHK(ensure(c, c->g64a, kk));
    CHK(ensure(c, c->g64b, kk));
    CHK(ensure(c, c->gmix64, kk));
    CHK(ensure(c, c->h64, kk));
    return CMF_OK;
}
static int launch_hess64(cmf_ctx *c, const double *A, double a, const double *B, double b, double diag) {
    Timed tm(c, CMF_K_ELEMWISE);
    hipLaunchKernelGGL(hess64_build_kernel, dim3((unsigned)std::min(256, (c->kp * c->kp + 255) / 256)), dim3(256), 0, c->stream,
                       (double *)c->h64.p, A, a, B, b, diag, c->kp, c->k);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// U (left=true: T = X, rows m) or Z (left=false: T = Y^T, rows p) with a shared Hessian.
// `vgram_ready`: c->g64a / c->G2 already hold V^T V (the caller's previous sweep used the same V).
static int sweep_side_shared(cmf_ctx *c, bool is_u, double scale, double l1, double l2, double pert, bool nn, bool vgram_ready = false) {
    const int which = is_u ? CMF_U : CMF_Z;
    const int64_t rows = c->frows_pad[which];
    float *F = c->F[which], *V = c->F[CMF_V];
    const bool f64 = use_shared64(c);
    if (f64) {
        CHK(ensure_shared64(c));
        if (!vgram_ready) CHK(gram64(c, V, c->dp, (double *)c->g64a.p, c->G2)); // V^T V, float64 + float32 copy
    } else {
        CHK(gemm(c, MODE_TN, V, c->kp, V, c->kp, c->G2, c->kp, c->kp, c->dp)); // V^T V
    }
    // H and its safe inverse first: when the host learns that the clamp did not act (lambda_min >= pert, the Cholesky route
    // of k > 64) and l1 = 0, the update is ONE product -- with H = s G + l2 I the step is (F H - s T O) H^-1 = F - s T O H^-1,
    // so F - step = s (T O) H^-1: no F G product, no cancellation against F
    if (f64) CHK(launch_hess64(c, (const double *)c->g64a.p, scale, nullptr, 0.0, l2));
    else CHK(launch_ew(c, axpby_diag_kernel, (int64_t)c->kp * c->kp, c->Hm, (const float *)c->G2, (float)scale,
                       (const float *)nullptr, 0.f, (float)l2, c->kp, c->k));
    bool plain = false;
    CHK(shared_inverse(c, pert, &plain));
    if (f64 && use_reassoc(c)) { // F <- clamp(F E + T (s V Hinv) - l1 sign(F) Hinv), see above
        if (!plain) CHK(clamp_defect(c));
        CHK(ensure(c, c->opr, (size_t)std::max(c->dp, c->mp + c->pp) * c->kp * sizeof(float)));
        float *Op = (float *)c->opr.p;
        CHK(factor_times_hinv(c, V, c->dp, scale, Op));
        {
            // unclamped inverse, no l1 term, data kept as blocked CSR: the product IS the new factor -- written (clamped) by the SpMM
            const int dw = is_u ? 0 : 1;
            const CsrDev &A = c->sp[dw][is_u ? 0 : 1];
            if (plain && l1 == 0.0 && c->sparse[dw] && !(dw == 0 ? c->X : c->Y) && spmm_can_update(c, A) && c->opt_arith == 0)
                return spmm(c, A, Op, F, rows, false, 0, nn ? 2 : 1);
        }
        if (is_u) CHK(data_times(c, 0, false, Op, c->num)); // X (s V Hinv)
        else CHK(data_times(c, 1, true, Op, c->num));       // Y^T (s V Hinv)
        return reassoc_finish(c, which, c->num, l1, plain, nn);
    }
    if (is_u) CHK(data_times(c, 0, false, V, c->num)); // X V
    else CHK(data_times(c, 1, true, V, c->num));       // Y^T V
    if (plain && l1 == 0.0 && c->opt_direct_step && c->opt_fused_mu && c->kp <= 256) {
        Epilogue e;
        e.kind = EPI_DIRECT; e.F = F; e.out = F; e.a = scale; e.rows = c->frows[which]; e.kvalid = c->k; e.nn = nn ? 1 : 0;
        if (small_tile_ok(c, rows)) return factor_update(c, c->num, c->Hinv, e, rows);
        return gemm(c, MODE_NN, c->num, c->kp, c->Hinv, c->kp, c->den, rows, c->kp, c->kp, false, &e);
    }
    if (c->opt_fused_mu && c->kp <= 256) { // grad = s (F G - T O) + l1 sign F + l2 F in the epilogue of F (V^T V)
        Epilogue e;
        e.kind = EPI_GRAD; e.F = F; e.P = c->num; e.out = c->den; e.a = scale; e.b = l1; e.c = l2;
        if (small_tile_ok(c, rows)) CHK(factor_update(c, F, c->G2, e, rows));
        else CHK(gemm(c, MODE_NN, F, c->kp, c->G2, c->kp, c->den, rows, c->kp, c->kp, false, &e));
    } else {
        CHK(gemm(c, MODE_NN, F, c->kp, c->G2, c->kp, c->den, rows, c->kp, c->kp));           // F (V^T V)
        CHK(launch_ew(c, newton_grad_kernel, rows * c->kp, c->den, (const float *)c->den, (float)scale, (const float *)c->num,
                      (float)-scale, (const float *)F, (float)l1, (float)l2, rows * c->kp));
    }
    return shared_apply(c, which, nn);
}

extern "C" int cmf_newton_v_partials(cmf_ctx *c, double alpha, float *buf) {
    NEED_PROBLEM(c);
    if (!buf) return fail(CMF_EINVAL, "null buffer");
    if (!have_data(c, 0) || !have_data(c, 1)) return fail(CMF_EINVAL, "X and Y must be set before a V update");
    DeviceGuard dg(c->device);
    float *P = buf, *Gs = buf + c->dp * c->kp;
    CHK(data_times(c, 0, true, c->F[CMF_U], c->num));  // X^T U
    CHK(data_times(c, 1, false, c->F[CMF_Z], c->den)); // Y Z
    CHK(launch_ew(c, axpby_kernel, c->dp * c->kp, P, (const float *)c->num, (float)alpha, (const float *)c->den,
                  (float)(1.0 - alpha), c->dp * c->kp));
    c->gmix64_valid = false;
    if (use_shared64(c)) {
        // Grams in float64; the (all-reducible) buffer carries their float32 rounding, the float64 mix stays behind for a
        // cmf_newton_v_apply that follows directly inside cmf_newton_step (single GPU: no collective in between)
        CHK(ensure_shared64(c));
        CHK(gram64(c, c->F[CMF_U], c->mp, (double *)c->g64a.p, nullptr));
        CHK(gram64(c, c->F[CMF_Z], c->pp, (double *)c->g64b.p, nullptr));
        Timed tm(c, CMF_K_ELEMWISE);
        const unsigned nb = (unsigned)std::min(256, (c->kp * c->kp + 255) / 256);
        hipLaunchKernelGGL(hess64_build_kernel, dim3(nb), dim3(256), 0, c->stream, (double *)c->gmix64.p, (const double *)c->g64a.p, alpha,
                           (const double *)c->g64b.p, 1.0 - alpha, 0.0, c->kp, c->kp);
        hipLaunchKernelGGL(axpby64_to_f32_kernel, dim3(nb), dim3(256), 0, c->stream, Gs, (const double *)c->gmix64.p, 1.0,
                           (const double *)nullptr, 0.0, (int64_t)c->kp * c->kp);
        HIPCHK(hipGetLastError());
        return CMF_OK;
    }
    CHK(gemm(c, MODE_TN, c->F[CMF_U], c->kp, c->F[CMF_U], c->kp, c->G, c->kp, c->kp, c->mp));
    CHK(gemm(c, MODE_TN, c->F[CMF_Z], c->kp, c->F[CMF_Z], c->kp, c->G2, c->kp, c->kp, c->pp));
    return launch_ew(c, axpby_kernel, (int64_t)c->kp * c->kp, Gs, (const float *)c->G, (float)alpha, (const float *)c->G2,
                     (float)(1.0 - alpha), (int64_t)c->kp * c->kp);
}

extern "C" int cmf_newton_v_apply(cmf_ctx *c, const float *buf, double l1, double l2, int nn_mask, double pert) {
    NEED_PROBLEM(c);
    if (!buf) return fail(CMF_EINVAL, "null buffer");
    DeviceGuard dg(c->device);
    const float *P = buf, *Gs = buf + c->dp * c->kp;
    float *V = c->F[CMF_V];
    const bool mix64 = c->gmix64_valid; // armed by cmf_newton_step only
    c->gmix64_valid = false;
    // H = Gmix + l2 I and its safe inverse first (see sweep_side_shared): an unclamped inverse and l1 = 0 make the sweep one
    // product, V - (V H - P) H^-1 = P H^-1
    if (use_shared64(c)) {
        CHK(ensure_shared64(c));
        if (mix64) {
            CHK(launch_hess64(c, (const double *)c->gmix64.p, 1.0, nullptr, 0.0, l2));
        } else { // the Gram arrives in float32 (summed over ranks by the caller's all-reduce): promote, factor in float64
            Timed tm(c, CMF_K_ELEMWISE);
            hipLaunchKernelGGL(hess64_from_f32_kernel, dim3((unsigned)std::min(256, (c->kp * c->kp + 255) / 256)), dim3(256), 0, c->stream,
                               (double *)c->h64.p, Gs, 1.0, l2, c->kp, c->k);
            HIPCHK(hipGetLastError());
        }
    } else {
        CHK(launch_ew(c, axpby_diag_kernel, (int64_t)c->kp * c->kp, c->Hm, Gs, 1.0f, (const float *)nullptr, 0.f, (float)l2,
                      c->kp, c->k));
    }
    bool plain = false;
    CHK(shared_inverse(c, pert, &plain));
    const bool nnv = (nn_mask & CMF_NN_V) != 0;
    if (plain && l1 == 0.0 && c->opt_direct_step && c->opt_fused_mu && c->kp <= 256) {
        Epilogue e;
        e.kind = EPI_DIRECT; e.F = V; e.out = V; e.a = 1.0; e.rows = c->frows[CMF_V]; e.kvalid = c->k; e.nn = nnv ? 1 : 0;
        if (small_tile_ok(c, c->dp)) return factor_update(c, P, c->Hinv, e, c->dp);
        return gemm(c, MODE_NN, P, c->kp, c->Hinv, c->kp, c->den, c->dp, c->kp, c->kp, false, &e);
    }
    if (c->opt_fused_mu && c->kp <= 256) {
        Epilogue e;
        e.kind = EPI_GRAD; e.F = V; e.P = P; e.out = c->den; e.a = 1.0; e.b = l1; e.c = l2;
        if (small_tile_ok(c, c->dp)) CHK(factor_update(c, V, Gs, e, c->dp));
        else CHK(gemm(c, MODE_NN, V, c->kp, Gs, c->kp, c->den, c->dp, c->kp, c->kp, false, &e)); // V Gmix - P + reg
    } else {
        CHK(gemm(c, MODE_NN, V, c->kp, Gs, c->kp, c->den, c->dp, c->kp, c->kp)); // V Gmix
        CHK(launch_ew(c, newton_grad_kernel, c->dp * c->kp, c->den, (const float *)c->den, 1.0f, P, -1.0f, (const float *)V,
                      (float)l1, (float)l2, c->dp * c->kp));
    }
    return shared_apply(c, CMF_V, nnv);
}

// ---- V sweep with the single shared Hessian, re-associated form, in three stages so that a row-sharded run can put its two
// collectives between them (the shards' Grams must be summed BEFORE the data pass: a float64 all-reduce of k_pad^2 doubles, 512 KB
// at k_pad = 256; then the usual all-reduce of the d x k partial):
//   cmf_newton_v_gram      gbuf = alpha U^T U + (1 - alpha) Z^T Z of the local rows (float64, device)
//   cmf_newton_v_products  H = gbuf + l2 I, safe inverse, pbuf = X^T (alpha U Hinv) + Y ((1 - alpha) Z Hinv) of the local rows
//   cmf_newton_v_finish    V <- clamp(V E + pbuf - l1 sign(V) Hinv)
// Reference: NewtonSolver._newton_update_V, sg_sample_ratio == 1, linear links (pycmf/cmf_solvers.py:436-450, :321-326).
extern "C" int cmf_newton_v_gram(cmf_ctx *c, double alpha, double *gbuf) {
    NEED_PROBLEM(c);
    if (!gbuf) return fail(CMF_EINVAL, "null buffer");
    if (!use_reassoc(c)) return fail(CMF_EUNSUPPORTED, "cmf_newton_v_gram needs the float64 shared Hessian (shared_hessian_f64, newton_reassoc) and k_pad <= 1024");
    DeviceGuard dg(c->device);
    CHK(ensure_shared64(c));
    CHK(gram64(c, c->F[CMF_U], c->mp, (double *)c->g64a.p, nullptr));
    CHK(gram64(c, c->F[CMF_Z], c->pp, (double *)c->g64b.p, nullptr));
    Timed tm(c, CMF_K_ELEMWISE);
    hipLaunchKernelGGL(hess64_build_kernel, dim3((unsigned)std::min(256, (c->kp * c->kp + 255) / 256)), dim3(256), 0, c->stream, gbuf,
                       (const double *)c->g64a.p, alpha, (const double *)c->g64b.p, 1.0 - alpha, 0.0, c->kp, c->kp);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

extern "C" int cmf_newton_v_products(cmf_ctx *c, double alpha, double l2, double pert, const double *gbuf, float *pbuf) {
    NEED_PROBLEM(c);
    if (!gbuf || !pbuf) return fail(CMF_EINVAL, "null buffer");
    if (!have_data(c, 0) || !have_data(c, 1)) return fail(CMF_EINVAL, "X and Y must be set before a V update");
    if (!use_reassoc(c)) return fail(CMF_EUNSUPPORTED, "cmf_newton_v_products needs the float64 shared Hessian and k_pad <= 1024");
    DeviceGuard dg(c->device);
    c->hess_psd = (alpha >= 0.0 && alpha <= 1.0 && l2 >= 0.0);
    CHK(ensure_shared64(c));
    CHK(launch_hess64(c, gbuf, 1.0, nullptr, 0.0, l2));
    bool plain = false;
    CHK(shared_inverse(c, pert, &plain));
    c->v_plain = plain;
    if (!plain) CHK(clamp_defect(c));
    CHK(ensure(c, c->opr, (size_t)std::max(c->dp, c->mp + c->pp) * c->kp * sizeof(float)));
    float *Ou = (float *)c->opr.p, *Oz = Ou + c->mp * c->kp;
    CHK(factor_times_hinv(c, c->F[CMF_U], c->mp, alpha, Ou));
    CHK(factor_times_hinv(c, c->F[CMF_Z], c->pp, 1.0 - alpha, Oz));
    CHK(data_times(c, 0, true, Ou, pbuf));         // X^T (alpha U Hinv)
    return data_times(c, 1, false, Oz, pbuf, true); // + Y ((1 - alpha) Z Hinv)
}

extern "C" int cmf_newton_v_finish(cmf_ctx *c, float *pbuf, double l1, int nn_mask) {
    NEED_PROBLEM(c);
    if (!pbuf) return fail(CMF_EINVAL, "null buffer");
    DeviceGuard dg(c->device);
    const bool plain = c->v_plain;
    c->v_plain = false;
    return reassoc_finish(c, CMF_V, pbuf, l1, plain, (nn_mask & CMF_NN_V) != 0);
}

extern "C" int cmf_newton_uz_update(cmf_ctx *c, double alpha, double l1, double l2, int nn_mask, int upd, double pert) {
    NEED_PROBLEM(c);
    DeviceGuard dg(c->device);
    c->hess_psd = (alpha >= 0.0 && alpha <= 1.0 && l2 >= 0.0);
    bool vgram = false;
    if (upd & CMF_UPD_U) {
        if (!have_data(c, 0)) return fail(CMF_EINVAL, "X must be set before a U update");
        CHK(sweep_side_shared(c, true, alpha, l1, l2, pert, (nn_mask & CMF_NN_U) != 0));
        vgram = use_shared64(c); // V has not moved: the Z sweep reuses V^T V
    }
    if (upd & CMF_UPD_Z) {
        if (!have_data(c, 1)) return fail(CMF_EINVAL, "Y must be set before a Z update");
        CHK(sweep_side_shared(c, false, 1.0 - alpha, l1, l2, pert, (nn_mask & CMF_NN_Z) != 0, vgram));
    }
    return CMF_OK;
}

// float64 path of the single shared inverse, exposed for tests: H (host, k x k float64, symmetric) -> safe_inverse(H)
extern "C" int cmf_safe_invert_f64(cmf_ctx *c, const double *H, double *out, int k, double pert) {
    NEED_PROBLEM(c);
    if (!H || !out || k != c->k) return fail(CMF_EINVAL, "cmf_safe_invert_f64: k must equal the problem's n_components");
    DeviceGuard dg(c->device);
    CHK(ensure_shared64(c));
    const int kp = c->kp;
    std::vector<double> host((size_t)kp * kp, 0.0);
    for (int r = 0; r < kp; ++r)
        for (int q = 0; q < kp; ++q) host[(size_t)r * kp + q] = (r < k && q < k) ? H[(size_t)r * k + q] : (r == q ? 1.0 : 0.0);
    HIPCHK(hipMemcpyAsync(c->h64.p, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int rc = shared_inverse64(c, (const double *)c->h64.p, k, pert, true);
    if (rc != CMF_OK) return rc == CMF_EUNSUPPORTED ? fail(CMF_EUNSUPPORTED, "float64 shared inverse did not converge (matrix not positive semi-definite?)") : rc;
    std::vector<float> res((size_t)kp * kp);
    HIPCHK(hipMemcpyAsync(res.data(), c->Hinv, res.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int r = 0; r < k; ++r)
        for (int q = 0; q < k; ++q) out[(size_t)r * k + q] = (double)res[(size_t)r * kp + q];
    return CMF_OK;
}

// ---- per-row machinery --------------------------------------------------------------------
// host-supplied index lists, already on the device: every entry must lie in [0, n)
static int check_index_lists(cmf_ctx *c, const int32_t *dev_idx, int64_t total, int64_t n) {
    int *flag = (int *)(c->dscalar + 6);
    HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)c->num_cu * 8);
    hipLaunchKernelGGL(index_range_kernel, dim3(grid), dim3(256), 0, c->stream, dev_idx, total, (int)n, flag);
    HIPCHK(hipGetLastError());
    int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (bad) return fail(CMF_EINVAL, "sample index list holds an index outside [0, %lld)", (long long)n);
    return CMF_OK;
}

// `n` = number of candidates per list (the extent sampled from)
static int build_mask(cmf_ctx *c, DevBuf &mb, int64_t rows_pad, int64_t cols_pad, const int32_t *idx, int64_t nlists,
                      int64_t per, bool by_row, int64_t n = 0, int salt = 0) {
    CHK(ensure(c, mb, (size_t)rows_pad * cols_pad));
    HIPCHK(hipMemsetAsync(mb.p, 0, (size_t)rows_pad * cols_pad, c->stream));
    if (nlists * per == 0) return CMF_OK;
    if (c->dev_sampling) { // draw the samples on the device (counter-based)
        Timed tm(c, CMF_K_ELEMWISE);
        hipLaunchKernelGGL(sample_select_kernel, dim3((unsigned)nlists), dim3(256), 0, c->stream, (uint8_t *)mb.p, cols_pad,
                           by_row ? 1 : 0, (int32_t *)nullptr, nlists, (int)n, (int)per, c->dev_seed * 4 + (uint64_t)salt,
                           c->sample_off[salt == 0 ? CMF_U : (salt == 1 ? CMF_Z : CMF_V)]);
        HIPCHK(hipGetLastError());
        return CMF_OK;
    }
    CHK(ensure(c, c->idxbuf, (size_t)nlists * per * sizeof(int32_t)));
    HIPCHK(hipMemcpyAsync(c->idxbuf.p, idx, (size_t)nlists * per * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    CHK(check_index_lists(c, (const int32_t *)c->idxbuf.p, nlists * per, n > 0 ? n : (by_row ? cols_pad : rows_pad))); // also waits: idx is caller memory
    return launch_ew(c, scatter_mask_kernel, nlists * per, (uint8_t *)mb.p, cols_pad, (const int32_t *)c->idxbuf.p, nlists, per,
                     by_row ? 1 : 0);
}

static int khatri_rao(cmf_ctx *c, DevBuf &kb, const float *F, int64_t rows_pad) {
    const size_t bytes = (size_t)rows_pad * c->kp * c->kp * sizeof(float);
    if (bytes > ((size_t)96 << 30))
        return fail(CMF_EUNSUPPORTED, "per-row Hessian path: Khatri-Rao image of %lld x %d^2 floats exceeds the 96 GiB budget",
                    (long long)rows_pad, c->kp);
    CHK(ensure(c, kb, bytes));
    return launch_ew(c, khatri_rao_kernel, rows_pad * c->kp * (c->kp / 4), (float *)kb.p, F, rows_pad, c->kp);
}

static int64_t hessian_chunk_rows(const cmf_ctx *c, int64_t rows_pad) {
    const int64_t per = (int64_t)c->kp * c->kp * sizeof(float);
    int64_t ch = (((int64_t)2 << 30) / per) / 256 * 256;
    ch = std::max<int64_t>(256, ch);
    return std::min(ch, rows_pad);
}

// residual / weight images of one data matrix: S = L Rt^T in the data's own layout
static int residual_images(cmf_ctx *c, bool x_side, int link, double scale, const uint8_t *mask, float *R, float *W,
                           bool w_slope) {
    NtOut o;
    o.link = link; o.scale_r = (float)scale; o.scale_w = (float)scale; o.w_is_slope = w_slope ? 1 : 0;
    o.R = R; o.W = W; o.mask = mask;
    if (x_side) {
        o.T = c->X; o.ldt = c->dp; o.ldr = c->dp; o.ldm = c->dp;
        return gemm_nt(c, c->F[CMF_U], c->mp, c->m, c->F[CMF_V], c->dp, c->d, o);
    }
    o.T = c->Y; o.ldt = c->pp; o.ldr = c->pp; o.ldm = c->pp;
    return gemm_nt(c, c->F[CMF_V], c->dp, c->d, c->F[CMF_Z], c->pp, c->p, o);
}

struct RowSide {
    bool active = false;
    const float *O = nullptr;   // other factor
    const int32_t *lists = nullptr;
    int64_t per = 0;            // samples per row
    const float *T = nullptr;
    int64_t t_row = 0, t_col = 0;
    double scale = 1.0;
    int link = 0;
    int cls = 0;                // > 0: shared partial sums over groups of `cls` rows (linear link, sampled; cmf_rowhess.hip.h)
    int64_t n = 0;              // candidates the lists draw from
    int slot = 0;               // which of the two class-list buffers
    // natively sparse data on this side: the row kernel runs with zero targets and the stored values of data row i that lie in
    // its sample enter the gradient afterwards (sparse_target_term); `sp` = CSR image whose row i belongs to factor row i
    const CsrDev *sp = nullptr;
    const int32_t *sorted = nullptr; // ascending copies of the lists (null: not sampled)
};

// the shared part of a sweep's Hessians as the refinement needs it: scale * F^T F, re-formed in float64
struct SharedPart {
    const float *F = nullptr;
    int64_t rows_pad = 0;
    double scale = 0.0;
};

static int sample_lists(cmf_ctx *c, DevBuf &lb, DevBuf &mb, const int32_t *host_idx, int64_t nlists, int64_t per, int64_t n,
                        int salt, const int32_t **out, DevBuf *sorted_buf = nullptr, const int32_t **sorted_out = nullptr);
static int refine_rows64(cmf_ctx *c, int which, const RowSide &s1, const RowSide &s2, const SharedPart &sh, const RowSide *shside,
                         double diag, double l1, double l2, int64_t r0, float *step, double pert);
// what the float64 refinement needs to know about a masked-dense sweep (the same sides, as index lists)
struct RefineSpec {
    bool on = false;
    RowSide s1, s2, shs;
    SharedPart shared;
    double l1 = 0.0, l2 = 0.0;
};

// finish a per-row sweep: for every chunk of rows build H_i, invert, step; then apply
struct RowHess {
    // H_i = [tn ? A1^T : A1] KR1  (+ A2 KR2)  + S + diag I
    const float *A1 = nullptr; int64_t lda1 = 0; bool a1_tn = false; const float *KR1 = nullptr; int64_t kred1 = 0;
    const float *A2 = nullptr; int64_t lda2 = 0; bool a2_tn = false; const float *KR2 = nullptr; int64_t kred2 = 0;
    const float *S = nullptr;
    double diag = 0.0;
};

static int per_row_finish(cmf_ctx *c, int which, const RowHess &h, double pert, bool nn, const RefineSpec &rf = RefineSpec()) {
    const int64_t rows_pad = c->frows_pad[which], rows = c->frows[which];
    const int64_t kk = (int64_t)c->kp * c->kp;
    c->refined_sweep = 0;
    const int64_t chunk = hessian_chunk_rows(c, rows_pad);
    CHK(ensure(c, c->hrows, (size_t)chunk * kk * sizeof(float)));
    float *Hc = (float *)c->hrows.p;
    float *grad = c->num, *step = c->den;
    for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
        const int64_t nr_pad = std::min(chunk, rows_pad - r0);
        const int64_t nr = std::min(nr_pad, rows - r0);
        bool have = false;
        if (h.A1) {
            const float *A = h.a1_tn ? h.A1 + r0 : h.A1 + r0 * h.lda1;
            CHK(gemm(c, h.a1_tn ? MODE_TN : MODE_NN, A, h.lda1, h.KR1, kk, Hc, nr_pad, kk, h.kred1));
            have = true;
        }
        if (h.A2) {
            const float *A = h.a2_tn ? h.A2 + r0 : h.A2 + r0 * h.lda2;
            CHK(gemm(c, h.a2_tn ? MODE_TN : MODE_NN, A, h.lda2, h.KR2, kk, Hc, nr_pad, kk, h.kred2, have));
            have = true;
        }
        CHK(launch_ew(c, hessian_finalize_kernel, nr_pad * kk, Hc, h.S, (float)h.diag, nr_pad, c->kp, c->k, have ? 1 : 0));
        CHK(safe_solve_rows(c, Hc, grad + r0 * c->kp, step + r0 * c->kp, nr, c->k, c->kp, pert, RowCert(), rf.on, c->F[which] + r0 * c->kp));
        if (rf.on) CHK(refine_rows64(c, which, rf.s1, rf.s2, rf.shared, rf.shs.active ? &rf.shs : nullptr, h.diag, rf.l1, rf.l2, r0, step + r0 * c->kp, pert));
    }
    return launch_ew(c, newton_apply_kernel, rows_pad * c->kp, c->F[which], (const float *)step, rows, c->kp, c->k,
                     rows_pad * c->kp, nn ? 1 : 0);
}

// U sweep (cmf_solvers.py:394-430) / Z sweep (:488-508) in per-row form
static int sweep_side_rows(cmf_ctx *c, bool is_u, int link, double scale, double l1, double l2, double pert, bool nn,
                           const int32_t *idx, int64_t per) {
    const int which = is_u ? CMF_U : CMF_Z;
    const int64_t rows_pad = c->frows_pad[which];
    const int64_t img = is_u ? c->mp * c->dp : c->dp * c->pp;
    const uint8_t *mask = nullptr;
    if (idx || c->dev_sampling) {
        // U: list i = columns (over d) of X row i;  Z: list i = rows (over d) of Y column i
        CHK(build_mask(c, c->mask1, is_u ? c->mp : c->dp, is_u ? c->dp : c->pp, idx, c->frows[which], per, is_u, c->d, is_u ? 0 : 1));
        mask = (const uint8_t *)c->mask1.p;
    }
    CHK(ensure(c, c->resid, (size_t)img * sizeof(float)));
    CHK(ensure(c, c->resid2, (size_t)img * sizeof(float)));
    float *R = (float *)c->resid.p, *W = (float *)c->resid2.p;
    CHK(residual_images(c, is_u, link, scale, mask, R, W, link == CMF_LINK_LOGIT));
    float *V = c->F[CMF_V];
    if (is_u) CHK(gemm(c, MODE_NN, R, c->dp, V, c->kp, c->num, c->mp, c->kp, c->dp));
    else CHK(gemm(c, MODE_TN, R, c->pp, V, c->kp, c->num, c->pp, c->kp, c->dp));
    CHK(launch_ew(c, newton_grad_kernel, rows_pad * c->kp, c->num, (const float *)c->num, 1.0f, (const float *)nullptr, 0.f,
                  (const float *)c->F[which], (float)l1, (float)l2, rows_pad * c->kp));
    CHK(khatri_rao(c, c->kr1, V, c->dp));
    RowHess h;
    h.A1 = W; h.lda1 = is_u ? c->dp : c->pp; h.a1_tn = !is_u; h.KR1 = (const float *)c->kr1.p; h.kred1 = c->dp;
    // the logit Hessian of U carries no l2 term (:427-428); Z always does (:501-506)
    h.diag = (link == CMF_LINK_LOGIT && (is_u || !c->opt_zlogit_l2)) ? 0.0 : l2;
    RefineSpec rf;
    if (c->opt_refine && c->hess_psd && (is_u ? c->X : c->Y)) {
        rf.on = true; rf.l1 = l1; rf.l2 = l2;
        RowSide &sd = rf.s1;
        sd.active = true; sd.O = V; sd.scale = scale; sd.link = link; sd.n = c->d;
        sd.per = mask ? per : c->d;
        if (mask) CHK(sample_lists(c, c->lists1, c->mask1, idx, c->frows[which], per, c->d, is_u ? 0 : 1, &sd.lists));
        if (is_u) { sd.T = c->X; sd.t_row = c->dp; sd.t_col = 1; }
        else { sd.T = c->Y; sd.t_row = 1; sd.t_col = c->pp; }
    }
    return per_row_finish(c, which, h, pert, nn, rf);
}

// V sweep in per-row form (cmf_solvers.py:432-486)
static int sweep_v_rows(cmf_ctx *c, double alpha, double l1, double l2, int x_link, int y_link, double pert, bool nn,
                        const int32_t *vx_idx, int64_t per_x, const int32_t *vy_idx, int64_t per_y) {
    const uint8_t *mx = nullptr, *my = nullptr;
    if (vx_idx || c->dev_sampling) { // list i = rows (over m) of X column i
        CHK(build_mask(c, c->mask1, c->mp, c->dp, vx_idx, c->d, per_x, false, c->m, 2));
        mx = (const uint8_t *)c->mask1.p;
    }
    if (vy_idx || c->dev_sampling) { // list i = columns (over p) of Y row i
        CHK(build_mask(c, c->mask2, c->dp, c->pp, vy_idx, c->d, per_y, true, c->p, 3));
        my = (const uint8_t *)c->mask2.p;
    }
    const bool x_shared = (x_link == CMF_LINK_LINEAR && !mx); // H_U = U^T U for every row
    const bool y_shared = (y_link == CMF_LINK_LINEAR && !my);
    const int64_t imgx = c->mp * c->dp, imgy = c->dp * c->pp;
    CHK(ensure(c, c->resid, (size_t)std::max(imgx, imgy) * sizeof(float)));
    float *R = (float *)c->resid.p, *WX = nullptr, *WY = nullptr;
    if (!x_shared) { CHK(ensure(c, c->resid2, (size_t)imgx * sizeof(float))); WX = (float *)c->resid2.p; }
    if (!y_shared) { CHK(ensure(c, c->resid3, (size_t)imgy * sizeof(float))); WY = (float *)c->resid3.p; }
    // gradient: alpha R_X^T U + (1-alpha) R_Y Z + reg
    CHK(residual_images(c, true, x_link, alpha, mx, R, WX, x_link == CMF_LINK_LOGIT));
    CHK(gemm(c, MODE_TN, R, c->dp, c->F[CMF_U], c->kp, c->num, c->dp, c->kp, c->mp));
    CHK(residual_images(c, false, y_link, 1.0 - alpha, my, R, WY, y_link == CMF_LINK_LOGIT));
    CHK(gemm(c, MODE_NN, R, c->pp, c->F[CMF_Z], c->kp, c->num, c->dp, c->kp, c->pp, true));
    CHK(launch_ew(c, newton_grad_kernel, c->dp * c->kp, c->num, (const float *)c->num, 1.0f, (const float *)nullptr, 0.f,
                  (const float *)c->F[CMF_V], (float)l1, (float)l2, c->dp * c->kp));
    RowHess h;
    h.diag = l2;
    const float *Gu = nullptr, *Gz = nullptr;
    if (x_shared) { CHK(gemm(c, MODE_TN, c->F[CMF_U], c->kp, c->F[CMF_U], c->kp, c->G, c->kp, c->kp, c->mp)); Gu = c->G; }
    if (y_shared) { CHK(gemm(c, MODE_TN, c->F[CMF_Z], c->kp, c->F[CMF_Z], c->kp, c->G2, c->kp, c->kp, c->pp)); Gz = c->G2; }
    if (Gu || Gz) {
        const float *A = Gu ? Gu : Gz;
        const float a = Gu ? (float)alpha : (float)(1.0 - alpha);
        const float *B = (Gu && Gz) ? Gz : nullptr;
        CHK(launch_ew(c, axpby_kernel, (int64_t)c->kp * c->kp, c->Hm, A, a, B, (float)(1.0 - alpha), (int64_t)c->kp * c->kp));
        h.S = c->Hm;
    }
    if (!x_shared) {
        CHK(khatri_rao(c, c->kr1, c->F[CMF_U], c->mp));
        h.A1 = WX; h.lda1 = c->dp; h.a1_tn = true; h.KR1 = (const float *)c->kr1.p; h.kred1 = c->mp;
    }
    if (!y_shared) {
        CHK(khatri_rao(c, c->kr2, c->F[CMF_Z], c->pp));
        const float **A = h.A1 ? &h.A2 : &h.A1;
        if (h.A1) { h.A2 = WY; h.lda2 = c->pp; h.a2_tn = false; h.KR2 = (const float *)c->kr2.p; h.kred2 = c->pp; }
        else { h.A1 = WY; h.lda1 = c->pp; h.a1_tn = false; h.KR1 = (const float *)c->kr2.p; h.kred1 = c->pp; }
        (void)A;
    }
    RefineSpec rf;
    if (c->opt_refine && c->hess_psd && c->X && c->Y && !(x_shared && y_shared)) {
        rf.on = true; rf.l1 = l1; rf.l2 = l2;
        RowSide sx, sy;
        sx.active = true; sx.O = c->F[CMF_U]; sx.scale = alpha; sx.link = x_link; sx.n = c->m; sx.per = mx ? per_x : c->m;
        sx.T = c->X; sx.t_row = 1; sx.t_col = c->dp;
        sy.active = true; sy.O = c->F[CMF_Z]; sy.scale = 1.0 - alpha; sy.link = y_link; sy.n = c->p; sy.per = my ? per_y : c->p;
        sy.T = c->Y; sy.t_row = c->pp; sy.t_col = 1;
        if (mx) CHK(sample_lists(c, c->lists1, c->mask1, vx_idx, c->d, per_x, c->m, 2, &sx.lists));
        if (my) CHK(sample_lists(c, c->lists2, c->mask2, vy_idx, c->d, per_y, c->p, 3, &sy.lists));
        // a shared side enters the refined Hessian as scale * Gram (float64) and the refined gradient as an unsampled linear side
        if (x_shared) { rf.shared.F = c->F[CMF_U]; rf.shared.rows_pad = c->mp; rf.shared.scale = alpha; rf.shs = sx; }
        else rf.s1 = sx;
        if (y_shared) { rf.shared.F = c->F[CMF_Z]; rf.shared.rows_pad = c->pp; rf.shared.scale = 1.0 - alpha; rf.shs = sy; }
        else rf.s2 = sy;
    }
    return per_row_finish(c, CMF_V, h, pert, nn, rf);
}

// ---- fused per-row path (cmf_rowhess.hip.h): only the sampled rows are touched ---------------
template <int KP>
static int launch_row_hess_kp(cmf_ctx *c, const RowHessArgs &a, int64_t nrows) {
    using Cfg = RowHessCfg<KP>;
#ifdef CMF_DIAG_BUILD // timing-only instantiations of the k_pad = 256 kernel (wrong results): python -m pycmf_amd.build --diag
    if (KP == 256 && c->opt_rowdiag > 0) { // DIAGNOSTIC builds of the k_pad = 256 kernel (wrong results)
#define CMF_ROWDIAG(D_, S_)                                                                                          \
    do {                                                                                                           \
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&row_hess_kernel<256, 1, D_, S_>), (int)Cfg::LDS_BYTES)); \
        hipLaunchKernelGGL((row_hess_kernel<256, 1, D_, S_>), dim3((unsigned)nrows), dim3(512), Cfg::LDS_BYTES, c->stream, a); \
    } while (0)
        if (c->opt_rowsym >= 3 && a.scale >= 0.f && !a.cls_cnt) { // the default kernel of the logit launches (tools/r06_rowdiag.sh)
            if (c->opt_rowdiag == 1) CMF_ROWDIAG(1, 4);
            else if (c->opt_rowdiag == 4) CMF_ROWDIAG(4, 4);
            else if (c->opt_rowdiag == 5) CMF_ROWDIAG(5, 4);
            else if (c->opt_rowdiag == 6) CMF_ROWDIAG(6, 4);
            else if (c->opt_rowdiag == 7) CMF_ROWDIAG(7, 4);
            else if (c->opt_rowdiag == 9) CMF_ROWDIAG(9, 4);
            else if (c->opt_rowdiag == 10) CMF_ROWDIAG(10, 4);
            else if (c->opt_rowdiag == 11) CMF_ROWDIAG(11, 4);
            else CMF_ROWDIAG(8, 4);
        } else if (c->opt_rowsym) {
            if (c->opt_rowdiag == 1) CMF_ROWDIAG(1, 1);
            else if (c->opt_rowdiag == 2) CMF_ROWDIAG(2, 1);
            else if (c->opt_rowdiag == 4) CMF_ROWDIAG(4, 1);
            else if (c->opt_rowdiag == 5) CMF_ROWDIAG(5, 1);
            else CMF_ROWDIAG(3, 1);
        } else {
            if (c->opt_rowdiag == 1) CMF_ROWDIAG(1, 0);
            else if (c->opt_rowdiag == 2) CMF_ROWDIAG(2, 0);
            else CMF_ROWDIAG(3, 0);
        }
#undef CMF_ROWDIAG
    } else
#endif
    if (KP == 256 && c->opt_rowsym >= 3 && a.scale >= 0.f && a.cls_cnt && a.cls_upper) { // class launch, upper blocks only
        constexpr int KS = KP == 256 ? 256 : 0;
        if constexpr (KS == 256) {
            constexpr size_t lds = Cfg::LDS_BYTES / 2; // one image per stage: two workgroups per CU
            if (c->opt_rowsym == 3) {
                CHK(allow_big_lds(c, reinterpret_cast<const void *>(&row_hess_kernel<256, 1, 0, 3, 1>), (int)lds));
                hipLaunchKernelGGL((row_hess_kernel<256, 1, 0, 3, 1>), dim3((unsigned)nrows), dim3(512), lds, c->stream, a);
            } else { // diagonal blocks on the 16-wide instruction
                CHK(allow_big_lds(c, reinterpret_cast<const void *>(&row_hess_kernel<256, 1, 0, 4, 1>), (int)lds));
                hipLaunchKernelGGL((row_hess_kernel<256, 1, 0, 4, 1>), dim3((unsigned)nrows), dim3(512), lds, c->stream, a);
            }
        }
    } else if (KP == 256 && c->opt_arith == 1 && c->opt_rowsym && a.scale >= 0.f) { // optional arithmetic: bf16 planes, six products
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&row_hess6_kernel), R6_LDS_BYTES));
        hipLaunchKernelGGL(row_hess6_kernel, dim3((unsigned)nrows), dim3(512), R6_LDS_BYTES, c->stream, a);
    } else if (KP == 256 && c->opt_rowsym >= 3 && a.scale >= 0.f) { // non-negative weights: single sqrt-weighted image
        constexpr int KS = KP == 256 ? 256 : 0;
        if constexpr (KS == 256) {
            if (c->opt_rowsym == 3) {
                CHK(allow_big_lds(c, reinterpret_cast<const void *>(&row_hess_kernel<256, 1, 0, 3>), (int)Cfg::LDS_BYTES));
                hipLaunchKernelGGL((row_hess_kernel<256, 1, 0, 3>), dim3((unsigned)nrows), dim3(512), Cfg::LDS_BYTES, c->stream, a);
            } else {
                CHK(allow_big_lds(c, reinterpret_cast<const void *>(&row_hess_kernel<256, 1, 0, 4>), (int)Cfg::LDS_BYTES));
                hipLaunchKernelGGL((row_hess_kernel<256, 1, 0, 4>), dim3((unsigned)nrows), dim3(512), Cfg::LDS_BYTES, c->stream, a);
            }
        }
    } else if (KP == 256 && c->opt_rowsym) {
        constexpr int KS = KP == 256 ? 256 : 0; // only the k_pad = 256 instantiation exists
        if constexpr (KS == 256) {
            CHK(allow_big_lds(c, reinterpret_cast<const void *>(&row_hess_kernel<256, 1, 0, 1>), (int)Cfg::LDS_BYTES));
            hipLaunchKernelGGL((row_hess_kernel<256, 1, 0, 1>), dim3((unsigned)nrows), dim3(512), Cfg::LDS_BYTES, c->stream, a);
        }
    } else if (c->opt_rowstagger) {
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&row_hess_kernel<KP, 1>), (int)Cfg::LDS_BYTES));
        hipLaunchKernelGGL((row_hess_kernel<KP, 1>), dim3((unsigned)nrows), dim3(512), Cfg::LDS_BYTES, c->stream, a);
    } else {
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&row_hess_kernel<KP, 0>), (int)Cfg::LDS_BYTES));
        hipLaunchKernelGGL((row_hess_kernel<KP, 0>), dim3((unsigned)nrows), dim3(512), Cfg::LDS_BYTES, c->stream, a);
    }
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

static int launch_row_hess(cmf_ctx *c, const RowHessArgs &a, int64_t nrows, double class_samples = -1.0) {
    if (nrows <= 0) return CMF_OK;
    // algorithmic credit: H_i is symmetric, so k_pad (k_pad + 1) / 2 multiply-adds per sample ("the
    // symmetric half may be credited as half"), plus the dot product and the gradient accumulation (2 k_pad each).
    // A class launch is credited with the sample rows of the factor rows it serves (Hessian part only: their gradient
    // is a pair of GEMMs); what it gathers is counted on the device (cmf_rowhess_samples).
    const double samples = class_samples >= 0.0 ? class_samples : (double)(a.nsplit > 1 ? nrows / a.nsplit : nrows) * (double)a.s;
    if (c->timing) {
        c->rh_credited += samples;
        if (class_samples < 0.0) c->rh_gathered += samples;
    }
    Timed tm(c, CMF_K_ROWHESS, samples * ((double)c->kp * (c->kp + 1.0) + (class_samples >= 0.0 ? 0.0 : 4.0 * c->kp)));
    switch (c->kp) {
    case 32: return launch_row_hess_kp<32>(c, a, nrows);
    case 64: return launch_row_hess_kp<64>(c, a, nrows);
    case 128: return launch_row_hess_kp<128>(c, a, nrows);
    case 256: return launch_row_hess_kp<256>(c, a, nrows);
    default: return fail(CMF_EUNSUPPORTED, "fused row kernel supports k_pad <= 256");
    }
}

// device index lists for one sweep side: from the host lists (parity mode) or from the device sampler
static int sample_lists(cmf_ctx *c, DevBuf &lb, DevBuf &mb, const int32_t *host_idx, int64_t nlists, int64_t per, int64_t n,
                        int salt, const int32_t **out, DevBuf *sorted_buf, const int32_t **sorted_out) {
    *out = nullptr;
    if (sorted_out) *sorted_out = nullptr;
    if (nlists * per == 0) return CMF_OK;
    CHK(ensure(c, lb, (size_t)nlists * per * sizeof(int32_t)));
    if (c->dev_sampling) {
        // exactly `per` winners per list, emitted in ascending index order
        (void)mb;
        Timed tm(c, CMF_K_ELEMWISE);
        hipLaunchKernelGGL(sample_select_kernel, dim3((unsigned)nlists), dim3(256), 0, c->stream, (uint8_t *)nullptr, (int64_t)0, 1,
                           (int32_t *)lb.p, nlists, (int)n, (int)per, c->dev_seed * 4 + (uint64_t)salt,
                           c->sample_off[salt == 0 ? CMF_U : (salt == 1 ? CMF_Z : CMF_V)]);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemcpyAsync(lb.p, host_idx, (size_t)nlists * per * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        CHK(check_index_lists(c, (const int32_t *)lb.p, nlists * per, n)); // also waits: host_idx is caller memory
        if (sorted_buf) { // the sparse target term looks samples up by binary search: an ascending copy of every list (NumPy's
                          // permutation order stays what the row kernel sums in)
            std::vector<int32_t> srt(host_idx, host_idx + nlists * per);
            for (int64_t l = 0; l < nlists; ++l) std::sort(srt.begin() + l * per, srt.begin() + (l + 1) * per);
            CHK(ensure(c, *sorted_buf, (size_t)nlists * per * sizeof(int32_t)));
            HIPCHK(hipMemcpyAsync(sorted_buf->p, srt.data(), (size_t)nlists * per * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream)); // srt is a local
            *sorted_out = (const int32_t *)sorted_buf->p;
        }
    }
    *out = (const int32_t *)lb.p;
    if (sorted_out && !*sorted_out) *sorted_out = *out; // the device sampler emits ascending lists
    return CMF_OK;
}


// zero targets for a native sparse side: one float of zeros read with strides 0
static int zero_targets(cmf_ctx *c, RowSide &sd, const CsrDev *sp) {
    CHK(ensure(c, c->zerobuf, 256));
    HIPCHK(hipMemsetAsync(c->zerobuf.p, 0, 256, c->stream));
    sd.T = (const float *)c->zerobuf.p; sd.t_row = 0; sd.t_col = 0; sd.sp = sp; sd.cls = 0;
    return CMF_OK;
}
// grad[r0 .. r0 + nr) -= s * sum over the stored values t_ij of data row i with j in S_i of t_ij o_j   (see csr_sampled_sub_kernel)
static int sparse_target_term(cmf_ctx *c, const RowSide &sd, float *grad_rows, int64_t r0, int64_t nr) {
    const CsrDev &A = *sd.sp;
    CsrView v{A.indptr, A.idx, A.val, A.rows};
    const int64_t nrows = std::min(nr, A.rows - r0);
    if (nrows <= 0) return CMF_OK;
    Timed tm(c, CMF_K_SPMM, 2.0 * (double)A.nnz * (double)c->kp * (double)nrows / (double)std::max<int64_t>(1, A.rows));
#define CMF_SUBK(GL_, CH_)                                                                                                     \
    do {                                                                                                                        \
        const int64_t waves = (nrows + (64 / GL_) - 1) / (64 / GL_);                                                            \
        hipLaunchKernelGGL((csr_sampled_sub_kernel<GL_, CH_>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c->stream, v, sd.O, c->kp, \
                           grad_rows, (float)sd.scale, r0, nrows, sd.sorted, sd.per);                                          \
    } while (0)
    switch (c->kp) {
    case 32: CMF_SUBK(8, 1); break;
    case 64: CMF_SUBK(16, 1); break;
    case 128: CMF_SUBK(32, 1); break;
    case 256: CMF_SUBK(64, 1); break;
    default: return fail(CMF_EUNSUPPORTED, "native sparse per-row sweeps support k_pad <= 256");
    }
#undef CMF_SUBK
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// Rows per group for the shared-partial-sum form of a linear, sampled side, or 0: worth it when the distinct samples of a
// group, n (1 - (1 - rho)^R), are well below the R rho n of the row-by-row form.
static size_t class_lists_lds(int64_t n, int R) { // class_lists_kernel: pattern words (transposed, W per thread) + hist[256][2^R + 1]
    const int64_t nw = (n + 3) / 4, W = (nw + 255) / 256;
    return (size_t)W * 256 * 4 + (size_t)256 * ((1 << R) + 1) * sizeof(unsigned short);
}
static int class_group_rows(const cmf_ctx *c, int link, bool sampled, int64_t per, int64_t n) {
    if (!sampled || link != CMF_LINK_LINEAR || c->opt_rowclasses == 0 || c->opt_rowclasses == 1 || per <= 0 || n <= 0 || n > 131072) return 0;
    const bool automatic = c->opt_rowclasses < 0;
    int R = std::min(c->opt_rowclasses, 6);
    if (automatic) { // as many rows per group as leave the 2^R - 1 classes ~256 samples each (8 K-steps of the row kernel per
        R = 2;       // class image written and read back: C3 measured 297 / 273 / 269 ms per iteration at R = 4 / 5 / 6)
        while (R < 6 && (n >> (R + 1)) >= 256) ++R;
    }
    while (R >= 2 && class_lists_lds(n, R) > (size_t)150 * 1024) --R; // the list kernel's LDS image of the candidates
    if (R < 2) return 0;
    if (automatic) { // worth it?  the class images, their sums, the lists and the gradient GEMMs cost 10-15 % of what the row kernel saves
        const double rho = (double)per / (double)n;
        const double factor = (1.0 - std::pow(1.0 - rho, R)) / (R * rho);
        if (factor >= 0.7) return 0;
    }
    return R;
}

// class lists (and pattern bytes) of a class side, once per sweep
static int build_class_lists(cmf_ctx *c, const RowSide &sd, int64_t rows) {
    const int R = sd.cls, NC1 = (1 << R) - 1;
    const int64_t ngroups = (rows + R - 1) / R, cap = std::min<int64_t>(sd.n, (int64_t)R * sd.per);
    DevBuf &ci = c->cls_idx[sd.slot], &co = c->cls_off[sd.slot], &cc = c->cls_cnt[sd.slot], &cp = c->cls_pat[sd.slot];
    CHK(ensure(c, ci, (size_t)ngroups * cap * sizeof(int32_t)));
    CHK(ensure(c, co, (size_t)ngroups * NC1 * sizeof(int64_t)));
    CHK(ensure(c, cc, (size_t)ngroups * NC1 * sizeof(int32_t)));
    CHK(ensure(c, cp, (size_t)ngroups * sd.n));
    const size_t lds = class_lists_lds(sd.n, R);
    CHK(allow_big_lds(c, reinterpret_cast<const void *>(&class_lists_kernel), (int)lds));
    Timed tm(c, CMF_K_ELEMWISE);
    hipLaunchKernelGGL(class_lists_kernel, dim3((unsigned)ngroups), dim3(256), lds, c->stream, sd.lists, sd.per, rows, (int)sd.n, R,
                       (int32_t *)ci.p, cap, (int64_t *)co.p, (int32_t *)cc.p, c->timing ? (unsigned long long *)(c->dscalar + 7) : nullptr,
                       (uint8_t *)cp.p);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// gradient part of a linear, sampled side as two GEMMs: s (mask o (L R^T - T)) times the other factor, into c->num.
// The byte mask comes from the pattern bytes build_class_lists left behind.
static int class_side_gradient(cmf_ctx *c, bool x_side, bool by_row, const RowSide &sd, int64_t nlists, double scale, int which,
                               bool accumulate) {
    DevBuf &mb = x_side ? c->mask1 : c->mask2;
    const int64_t rows_pad = x_side ? c->mp : c->dp, cols_pad = x_side ? c->dp : c->pp;
    CHK(ensure(c, mb, (size_t)rows_pad * cols_pad));
    HIPCHK(hipMemsetAsync(mb.p, 0, (size_t)rows_pad * cols_pad, c->stream));
    {
        Timed tm(c, CMF_K_ELEMWISE);
        const int R = sd.cls;
        const int64_t ngroups = (nlists + R - 1) / R;
        const uint8_t *patg = (const uint8_t *)c->cls_pat[sd.slot].p;
        if (by_row) {
            const unsigned grid = (unsigned)std::min<int64_t>((ngroups * sd.n + 255) / 256, (int64_t)c->num_cu * 32);
            hipLaunchKernelGGL(mask_rows_from_patterns_kernel, dim3(grid), dim3(256), 0, c->stream, (uint8_t *)mb.p, cols_pad, patg, ngroups, (int)sd.n,
                               R, nlists);
        } else {
            hipLaunchKernelGGL(mask_cols_from_patterns_kernel, dim3((unsigned)((sd.n + 63) / 64), (unsigned)((ngroups + 31) / 32)), dim3(256), 0,
                               c->stream, (uint8_t *)mb.p, cols_pad, patg, ngroups, (int)sd.n, R, nlists);
        }
        HIPCHK(hipGetLastError());
    }
    CHK(ensure(c, c->resid, (size_t)rows_pad * cols_pad * sizeof(float)));
    float *R = (float *)c->resid.p;
    CHK(residual_images(c, x_side, CMF_LINK_LINEAR, scale, (const uint8_t *)mb.p, R, nullptr, false));
    if (which == CMF_U) return gemm(c, MODE_NN, R, c->dp, c->F[CMF_V], c->kp, c->num, c->mp, c->kp, c->dp, accumulate);
    if (which == CMF_Z) return gemm(c, MODE_TN, R, c->pp, c->F[CMF_V], c->kp, c->num, c->pp, c->kp, c->dp, accumulate);
    if (x_side) return gemm(c, MODE_TN, R, c->dp, c->F[CMF_U], c->kp, c->num, c->dp, c->kp, c->mp, accumulate);
    return gemm(c, MODE_NN, R, c->pp, c->F[CMF_Z], c->kp, c->num, c->dp, c->kp, c->pp, accumulate);
}

// finish a sweep of factor `which` whose per-row parts come from up to two fused sides;
// c->num must already hold any shared-side gradient part if `grad_preloaded`

// Float64 refinement of the rows clamp_stats listed for this chunk (c->bad_host): per sample z = o_j . f_i, residual and Hessian
// weight in float64 (exact products of the float32 factor rows and data values, float64 sums); H_i = diag I + shared part + the
// per-row sums; g_i from the same residuals (stored values of natively sparse sides subtracted separately; the shared side of a
// V sweep enters the gradient as one more, unsampled, side); safe inverse by the float64 routine of the shared sweeps (Cholesky
// test, spectral clamp by float64 matrix polynomials; Jacobi for k <= 64); step = g_i H_i^-1 rounded once to float32.  One row
// at a time (~200 small launches for a clamped k = 256 row): the exception path of ill-conditioned problems, not a throughput
// path.  Reference: pycmf/cmf_solvers.py:394-508 with :346-356 on float64 Hessians.
// The batched form (cmf_refine64.hip.h) for 64 < n <= 256: the rows of the list travel together, `bs` at a time.
static int refine_rows64_batched(cmf_ctx *c, int which, const RowSide &s1, const RowSide &s2, const SharedPart &sh, const RowSide *shside,
                                 double diag, double l1, double l2, int64_t r0, float *step, double pert) {
    const int kp = c->kp, n = c->k;
    const int64_t kk = (int64_t)kp * kp;
    const RowSide *sides[3] = {&s1, &s2, shside};
    int64_t smax = 1;
    for (const RowSide *sd : sides)
        if (sd && sd->active) smax = std::max(smax, sd->per);
    const std::vector<int> bad = c->bad_host;
    c->bad_host.clear();
    if (!(s1.active || s2.active)) return CMF_OK; // no per-row side: nothing to redo
    // batch size: Hessian + two factor images per row, three weight / residual rows; the clamp's images (6 kk) only for its rows
    const int64_t per_row = (3 * kk + 4 * smax + kp) * (int64_t)sizeof(double);
    const int64_t bs = std::max<int64_t>(1, std::min<int64_t>((int64_t)bad.size(), std::min<int64_t>(4096, ((int64_t)6 << 30) / per_row)));
    CHK(ensure(c, c->rw64, (size_t)bs * 4 * smax * sizeof(double)));
    CHK(ensure(c, c->rh64, (size_t)(bs + 1) * kk * sizeof(double)));
    CHK(ensure(c, c->ref_w, (size_t)bs * 2 * kk * sizeof(double)));
    CHK(ensure(c, c->ref_g, (size_t)bs * kp * sizeof(double)));
    CHK(ensure(c, c->ref_i, (size_t)bs * 8 * sizeof(int)));
    double *w1 = (double *)c->rw64.p, *w2 = w1 + bs * smax, *r = w2 + bs * smax, *wx = r + bs * smax;
    double *H64 = (double *)c->rh64.p, *S64 = nullptr, *W = (double *)c->ref_w.p, *g = (double *)c->ref_g.p;
    int *dbad = (int *)c->ref_i.p, *dflag = dbad + bs, *dlidx = dflag + 2 * bs, *drows = dlidx + bs; // bad | flags (2 per row) | factor slot | batch row
    if (sh.F) {
        S64 = H64 + bs * kk;
        CHK(gram64(c, sh.F, sh.rows_pad, S64, nullptr));
    }
    auto side64 = [&](const RowSide *sd) {
        Ref64Side o;
        o.O = sd->O; o.lists = sd->lists; o.per = (int)sd->per; o.T = sd->T; o.t_row = sd->t_row; o.t_col = sd->t_col; o.scale = sd->scale;
        o.link = sd->link == CMF_LINK_LOGIT ? 1 : 0;
        return o;
    };
    Timed tm(c, CMF_K_EIGEN);
    for (size_t b0 = 0; b0 < bad.size(); b0 += (size_t)bs) {
        const int nb = (int)std::min<size_t>((size_t)bs, bad.size() - b0);
        HIPCHK(hipMemcpyAsync(dbad, bad.data() + b0, (size_t)nb * sizeof(int), hipMemcpyHostToDevice, c->stream));
        // ---- residuals, weights, gradients
        bool first_g = true;
        Ref64GramSide gs[2];
        for (int q = 0; q < 3; ++q) {
            const RowSide *sd = sides[q];
            if (!sd || !sd->active) continue;
            const Ref64Side o = side64(sd);
            double *wq = q == 0 ? w1 : (q == 1 ? w2 : wx); // (the shared side of a V sweep only enters the gradient: its weights are discarded)
            if (o.per > 0) {
                double *rq = r;
                hipLaunchKernelGGL(ref64_terms_kernel, dim3((unsigned)((o.per + 3) / 4), (unsigned)nb), dim3(256), 0, c->stream, o, (const float *)c->F[which], kp,
                                   (const int *)dbad, r0, rq, wq, smax);
                hipLaunchKernelGGL(ref64_grad_kernel, dim3((unsigned)(kp / 32), (unsigned)nb), dim3(256), 0, c->stream, o, kp, (const int *)dbad, r0,
                                   (const double *)rq, smax, g, first_g ? 1 : 0);
            } else if (first_g) {
                HIPCHK(hipMemsetAsync(g, 0, (size_t)nb * kp * sizeof(double), c->stream));
            }
            first_g = false;
            if (sd->sp)
                hipLaunchKernelGGL(ref64_sparse_grad_kernel, dim3((unsigned)((kp + 255) / 256), (unsigned)nb), dim3(256), 0, c->stream,
                                   (const int64_t *)sd->sp->indptr, (const int32_t *)sd->sp->idx, (const float *)sd->sp->val, sd->O, kp, sd->sorted,
                                   sd->per, sd->scale, (const int *)dbad, r0, g);
            if (q < 2) {
                gs[q].O = sd->O; gs[q].lists = sd->lists; gs[q].per = (int)sd->per; gs[q].scale = sd->scale;
                gs[q].w = sd->link == CMF_LINK_LOGIT ? wq : nullptr;
            }
            HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL(ref64_grad_finish_kernel, dim3((unsigned)((kp + 255) / 256), (unsigned)nb), dim3(256), 0, c->stream, g, (const float *)c->F[which],
                           (const int *)dbad, r0, l1, l2, n, kp);
        // ---- Hessians on the float64 matrix pipe
        {
            const int T = kp / 64, ntile = T * (T + 1) / 2;
            hipLaunchKernelGGL(ref64_wgram_kernel, dim3((unsigned)ntile, (unsigned)nb), dim3(256), 0, c->stream, gs[0], gs[1], smax, kp, n, (const int *)dbad, r0,
                               diag, (const double *)S64, sh.scale, H64);
        }
        // ---- threshold test and factor: H - pert I -> W[2 b], H -> W[2 b + 1]
        if (n <= 128) hipLaunchKernelGGL((chol64_reg_kernel<4>), dim3((unsigned)(2 * nb)), dim3(1024), 0, c->stream, (const double *)H64, n, kp, W, kk, kp, pert, 0.0, dflag, kk, 2);
        else hipLaunchKernelGGL((chol64_reg_kernel<8>), dim3((unsigned)(2 * nb)), dim3(1024), 0, c->stream, (const double *)H64, n, kp, W, kk, kp, pert, 0.0, dflag, kk, 2);
        HIPCHK(hipGetLastError());
        std::vector<int> flags((size_t)2 * nb);
        HIPCHK(hipMemcpyAsync(flags.data(), dflag, flags.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        std::vector<int> plain_l, plain_r, clamp_r;
        for (int b = 0; b < nb; ++b) {
            if (!flags[2 * b] && !flags[2 * b + 1]) { plain_l.push_back(2 * b + 1); plain_r.push_back(b); } // lambda_min >= pert: the clamp is the identity
            else clamp_r.push_back(b);
        }
        // ---- rows the clamp acts on: M = max(H, pert I) by the float64 matrix polynomials of shared_inverse64, batched
        // (in sub-batches: seven k_pad^2 float64 images per row -- 3.7 MB at k_pad = 256 -- are kept under 2 GiB whatever the
        // number of rows the clamp acts on: a sweep in which EVERY row of a 4096-row batch is clamped must not ask for 15 GiB)
        const size_t clamp_cap = std::max<size_t>(1, ((size_t)2 << 30) / (7 * kk * sizeof(double)));
        const std::vector<int> clamp_all = clamp_r;
        for (size_t c0 = 0; c->hess_psd && c0 < clamp_all.size(); c0 += clamp_cap) {
            clamp_r.assign(clamp_all.begin() + c0, clamp_all.begin() + std::min(clamp_all.size(), c0 + clamp_cap));
            const int ncl = (int)clamp_r.size();
            CHK(ensure(c, c->ref_ns, (size_t)ncl * 7 * kk * sizeof(double) + (size_t)ncl * sizeof(double)));
            double *Hc = (double *)c->ref_ns.p, *Bm = Hc + ncl * kk, *X = Bm + ncl * kk, *X2 = X + ncl * kk, *Y = X2 + ncl * kk, *Z = Y + ncl * kk, *M = Z + ncl * kk;
            double *cn = M + ncl * kk;
            HIPCHK(hipMemcpyAsync(drows, clamp_r.data(), (size_t)ncl * sizeof(int), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(ref64_gather_kernel, dim3(64, (unsigned)ncl), dim3(256), 0, c->stream, (const double *)H64, (const int *)drows, Hc, kk);
            hipLaunchKernelGGL(ns64_prepare_kernel, dim3((unsigned)ncl), dim3(1024), 0, c->stream, (const double *)Hc, Bm, X, n, kp, pert, cn, kk);
            HIPCHK(hipGetLastError());
            std::vector<double> hc((size_t)ncl);
            HIPCHK(hipMemcpyAsync(hc.data(), cn, hc.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            double hmax = pert;
            for (double v : hc) hmax = std::max(hmax, v);
            // eigenvalues closer than delta to the threshold keep an error <= delta in M (the clamp is continuous): 1e-5 relative in
            // that eigen-direction at worst, two growth steps fewer than the single-matrix path's 1e-6 (every step is three 256^3
            // float64 products PER ROW here)
            const double delta = 1e-5 * pert;
            const dim3 gg((unsigned)(kp / 32), (unsigned)(kp / 32), (unsigned)ncl);
            // Spectral map in front of the sign iteration (option refine_spectral_map).  H is positive semi-definite, so B = H - pert I
            // has its spectrum in [-pert, ||H||]: ||H|| / pert is 1e3 ... 1e4 here, and the growth phase spends log_3.44 of that ratio
            // in steps just to bring the eigenvalues near the threshold up from delta / ||H||.  f(lambda) = (lambda - pert) /
            // (lambda + pert) is increasing with f(pert) = 0, so sign(f(H)) = sign(B), and it maps [0, inf) into [-1, 1): the start
            // X0 = f(H) = I - 2 pert (H + pert I)^-1 needs no scaling and resolves |lambda - pert| >= delta as |x| >= delta / (2 pert).
            // Cost: one batched register Cholesky of H + pert I, its triangular inverse, one product -- against six growth steps
            // (eighteen products) saved at ||H|| / pert = 2600.
            bool mapped = false;
            if (c->opt_refine_map && n > 64) {
                int *mflag0 = dflag; // (the flags of the threshold test were read above)
                if (n <= 128) hipLaunchKernelGGL((chol64_reg_kernel<4>), dim3((unsigned)ncl), dim3(1024), 0, c->stream, (const double *)Hc, n, kp, Y, kk, kp, -pert, -pert, mflag0, kk, 1);
                else hipLaunchKernelGGL((chol64_reg_kernel<8>), dim3((unsigned)ncl), dim3(1024), 0, c->stream, (const double *)Hc, n, kp, Y, kk, kp, -pert, -pert, mflag0, kk, 1);
                HIPCHK(hipGetLastError());
                std::vector<int> f0((size_t)ncl);
                HIPCHK(hipMemcpyAsync(f0.data(), mflag0, f0.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                bool all_ok = true;
                for (int v : f0) all_ok = all_ok && v == 0;
                if (all_ok) {
                    if (kp <= 128) hipLaunchKernelGGL((tri_inverse64_reg_kernel<8>), dim3((unsigned)(kp / 16), (unsigned)ncl), dim3(256), 0, c->stream, (const double *)Y, n, kp, Z, kp, kp, (int64_t)kk, (int64_t)kk);
                    else hipLaunchKernelGGL((tri_inverse64_reg_kernel<16>), dim3((unsigned)(kp / 16), (unsigned)ncl), dim3(256), 0, c->stream, (const double *)Y, n, kp, Z, kp, kp, (int64_t)kk, (int64_t)kk);
                    // X0 = I - 2 pert Xt Xt^T  (Xt = L^-1 transposed, zero beyond the valid block: identity on the padding)
                    hipLaunchKernelGGL((gemm64_kernel<true>), gg, dim3(256), 0, c->stream, (const double *)Z, (const double *)Z, X, (const double *)nullptr, -2.0 * pert, 0.0, 1.0, kp,
                                       (float *)nullptr, 0, (const int *)nullptr, 0, kk);
                    HIPCHK(hipGetLastError());
                    mapped = true;
                }
            }
            int nq = (int)std::ceil(std::log(mapped ? 2.0 * pert / delta : hmax / delta) / std::log(3.4445));
            nq = std::min(std::max(nq, 4), 48);
            auto mm = [&](const double *A, const double *B, double *Cc, const double *D, double al, double be, double ga) {
                if (c->opt_gemm64_tile128 && kp % 128 == 0 &&
                    allow_big_lds(c, reinterpret_cast<const void *>(&gemm64_tile128_kernel), GEMM64_TILE128_LDS) == CMF_OK)
                    hipLaunchKernelGGL(gemm64_tile128_kernel, dim3((unsigned)(kp / 128), (unsigned)(kp / 128), (unsigned)ncl), dim3(512), GEMM64_TILE128_LDS, c->stream, A, B, Cc, D, al, be, ga, kp, (int64_t)kk);
                else
                    hipLaunchKernelGGL((gemm64_kernel<false>), gg, dim3(256), 0, c->stream, A, B, Cc, D, al, be, ga, kp, (float *)nullptr, 0, (const int *)nullptr, 0, kk);
            };
            for (int it = 0; it < nq; ++it) {
                mm(X, X, Y, nullptr, 1.0, 0.0, 0.0);
                mm(Y, Y, Z, Y, 2.0315, -4.7750, 3.4445);
                mm(X, Z, X2, nullptr, 1.0, 0.0, 0.0);
                std::swap(X, X2);
            }
            for (int it = 0; it < 5; ++it) { // from |x| in [0.7, 1.2]: 0.3 -> 0.14 -> 2.7e-2 -> 1.1e-3 -> 1.8e-6 -> 5e-12 of sign(B)
                mm(X, X, Y, nullptr, 1.0, 0.0, 0.0);
                mm(X, Y, X2, X, -0.5, 1.5, 0.0);
                std::swap(X, X2);
            }
            mm(X, Bm, M, Bm, 0.5, 0.5, pert);
            HIPCHK(hipGetLastError());
            // factor M into the rows' first factor slot (the failed H - pert I attempt left it unused)
            CHK(ensure(c, c->ref_w2, (size_t)ncl * kk * sizeof(double) + (size_t)ncl * sizeof(int)));
            double *WM = (double *)c->ref_w2.p;
            int *mflag = (int *)(WM + (int64_t)ncl * kk);
            if (n <= 128) hipLaunchKernelGGL((chol64_reg_kernel<4>), dim3((unsigned)ncl), dim3(1024), 0, c->stream, (const double *)M, n, kp, WM, kk, kp, 0.0, 0.0, mflag, kk, 1);
            else hipLaunchKernelGGL((chol64_reg_kernel<8>), dim3((unsigned)ncl), dim3(1024), 0, c->stream, (const double *)M, n, kp, WM, kk, kp, 0.0, 0.0, mflag, kk, 1);
            HIPCHK(hipGetLastError());
            std::vector<int> mf((size_t)ncl);
            HIPCHK(hipMemcpyAsync(mf.data(), mflag, mf.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            std::vector<int> cl, cr;
            for (int q = 0; q < ncl; ++q)
                if (!mf[q]) { cl.push_back(q); cr.push_back(clamp_r[q]); } // (a failed factorisation of M: not reached for PSD Hessians; the float32 step stands)
            if (!cl.empty()) {
                HIPCHK(hipMemcpyAsync(dlidx, cl.data(), cl.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
                HIPCHK(hipMemcpyAsync(drows, cr.data(), cr.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(ref64_solve_kernel, dim3((unsigned)cl.size()), dim3(256), 0, c->stream, (const double *)WM, kk, (const int *)dlidx, (const int *)drows,
                                   (int)cl.size(), (const double *)g, (const int *)dbad, step, n, kp);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(c->stream)); // dlidx / drows are reused below
                c->refined_total += (int64_t)cl.size();
            }
        }
        if (!plain_l.empty()) {
            HIPCHK(hipMemcpyAsync(dlidx, plain_l.data(), plain_l.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(drows, plain_r.data(), plain_r.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(ref64_solve_kernel, dim3((unsigned)plain_l.size()), dim3(256), 0, c->stream, (const double *)W, kk, (const int *)dlidx, (const int *)drows,
                               (int)plain_l.size(), (const double *)g, (const int *)dbad, step, n, kp);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(c->stream)); // the host vectors above go out of scope
            c->refined_total += (int64_t)plain_l.size();
        }
    }
    return CMF_OK;
}

static int refine_rows64(cmf_ctx *c, int which, const RowSide &s1, const RowSide &s2, const SharedPart &sh, const RowSide *shside,
                         double diag, double l1, double l2, int64_t r0, float *step, double pert) {
    if (c->bad_host.empty()) return CMF_OK;
    if (c->opt_refine_batched && c->k > 64 && c->k <= 256 && c->hess_psd)
        return refine_rows64_batched(c, which, s1, s2, sh, shside, diag, l1, l2, r0, step, pert);
    const int kp = c->kp, n = c->k;
    const size_t kk = (size_t)kp * kp;
    const RowSide *sides[3] = {&s1, &s2, shside};
    int64_t smax = 1;
    for (const RowSide *sd : sides)
        if (sd && sd->active) smax = std::max(smax, sd->per);
    CHK(ensure(c, c->rw64, (size_t)(2 * smax + kp) * sizeof(double)));
    CHK(ensure(c, c->rh64, 2 * kk * sizeof(double)));
    double *H64 = (double *)c->rh64.p, *S64 = nullptr, *w = (double *)c->rw64.p, *r = w + smax, *g = r + smax;
    if (sh.F) {
        S64 = H64 + kk;
        CHK(gram64(c, sh.F, sh.rows_pad, S64, nullptr));
    }
    const std::vector<int> bad = c->bad_host;
    c->bad_host.clear();
    for (int b : bad) {
        const int64_t i = r0 + b;
        const float *f = c->F[which] + i * kp;
        bool first_h = true, first_g = true;
        for (int q = 0; q < 3; ++q) {
            const RowSide *sd = sides[q];
            if (!sd || !sd->active) continue;
            const int32_t *list = sd->lists ? sd->lists + i * sd->per : nullptr;
            const int s = (int)sd->per;
            if (s > 0)
                hipLaunchKernelGGL(row_terms64_kernel, dim3((unsigned)((s + 3) / 4)), dim3(256), 0, c->stream, sd->O, kp, f, list, s,
                                   sd->link == CMF_LINK_LOGIT ? 1 : 0, sd->T, i * sd->t_row, sd->t_col, r, w);
            hipLaunchKernelGGL(row_grad64_kernel, dim3((unsigned)(kp / 32)), dim3(256), 0, c->stream, sd->O, kp, list, s, (const double *)r, sd->scale,
                               g, first_g ? 1 : 0);
            first_g = false;
            if (sd->sp)
                hipLaunchKernelGGL(row_sparse_grad64_kernel, dim3((unsigned)((kp + 255) / 256)), dim3(256), 0, c->stream, (const int64_t *)sd->sp->indptr,
                                   (const int32_t *)sd->sp->idx, (const float *)sd->sp->val, i, sd->O, kp,
                                   sd->sorted ? sd->sorted + i * sd->per : (const int32_t *)nullptr, sd->per, sd->scale, g);
            if (q < 2) { // the shared side's Hessian part is S64
                hipLaunchKernelGGL(weighted_gram64_kernel, dim3((unsigned)(kp / 32), (unsigned)(kp / 32)), dim3(256), 0, c->stream, sd->O, kp, n, list, s,
                                   (const double *)w, sd->scale, H64, first_h ? 1 : 0, diag, (const double *)S64, sh.scale);
                first_h = false;
            }
            HIPCHK(hipGetLastError());
        }
        if (first_h) continue; // no per-row side: nothing to redo
        hipLaunchKernelGGL(row_grad_finish64_kernel, dim3((unsigned)((kp + 255) / 256)), dim3(256), 0, c->stream, g, f, l1, l2, n, kp);
        const int rc = shared_inverse64(c, H64, n, pert, true);
        if (rc == CMF_EUNSUPPORTED) continue; // (not reached for positive semi-definite Hessians) the float32 step stands
        CHK(rc);
        hipLaunchKernelGGL(rowvec_mat64_kernel, dim3((unsigned)((kp + 255) / 256)), dim3(256), 0, c->stream, (const double *)g,
                           (const double *)c->hinv64.p, step + (int64_t)b * kp, n, kp);
        HIPCHK(hipGetLastError());
        ++c->refined_total;
    }
    return CMF_OK;
}

// class images of the k_pad = 256 symmetric kernel hold their 36 upper 32 x 32 blocks only (cmf_rowhess.hip.h)
static bool class_images_upper(const cmf_ctx *c, double scale) { return c->kp == 256 && c->opt_rowsym >= 3 && scale >= 0.0 && c->opt_rowdiag == 0; }
static int64_t class_image_floats(const cmf_ctx *c, double scale) { return class_images_upper(c, scale) ? 36 * CLS_BLOCK : (int64_t)c->kp * c->kp; }

static int fused_rows_finish(cmf_ctx *c, int which, const RowSide &s1, const RowSide &s2, const float *S, double diag, bool grad_preloaded,
                             double l1, double l2, double pert, bool nn, const SharedPart &shared = SharedPart(),
                             const RowSide *shside = nullptr) {
    c->refined_sweep = 0;
    const int64_t rows_pad = c->frows_pad[which], rows = c->frows[which];
    const int64_t kk = (int64_t)c->kp * c->kp;
    int64_t chunk = hessian_chunk_rows(c, rows_pad);
    if (c->opt_rowchunk > 0) chunk = std::min<int64_t>(chunk, rup(c->opt_rowchunk, 256)); // tests: force several chunks
    {
        // class images of a chunk: (2^R - 1) / R images per row (36 upper blocks each at k_pad = 256); keep them under 16 GiB (C3, R = 6:
        // 8192 rows per chunk, the cap of the Hessian chunk itself).
        // The memory cap is taken over BOTH class sides first; then ONE rounding to a common group boundary, lcm(256, R1, R2):
        // every chunk must start on a group boundary of every class side (g0 = r0 / R below)
        int64_t q = 256;
        for (const RowSide *sd : {&s1, &s2})
            if (sd->active && sd->cls) {
                const int64_t per_row = (((int64_t)1 << sd->cls) - 1) * class_image_floats(c, sd->scale) * (int64_t)sizeof(float) / sd->cls;
                chunk = std::min(chunk, std::max<int64_t>(256, (((int64_t)16 << 30) / per_row) / 256 * 256));
                q = std::lcm(q, (int64_t)sd->cls);
            }
        if (chunk < rows_pad) chunk = std::max<int64_t>(q, chunk / q * q);
    }
    CHK(ensure(c, c->hrows, (size_t)chunk * kk * sizeof(float)));
    float *Hc = (float *)c->hrows.p;
    float *grad = c->num, *step = c->den;
    // global certificate: every H_i of this sweep is  (sums of w o o^T, w >= 0) + (positive semi-definite shared part) + diag I,
    // so diag >= pert alone puts lambda_min(H_i) at or above the threshold of _safe_invert (cmf_solvers.py:346-356): the safe
    // inverse is the plain inverse for EVERY row and nobody runs the Cholesky test of H_i - pert I (half of the solve kernel).
    // The reference's own sparse Newton settings (l2_reg = 5, hessian_pertubation = 0.2) are such a case.
    const bool global_cert = c->hess_psd && c->opt_rowcert && !c->opt_choldiag && diag >= pert && c->k <= 256;
    if (global_cert) {
        CHK(ensure(c, c->certflag, 2 * sizeof(int)));
        HIPCHK(hipMemsetAsync(c->certflag.p, 0, 2 * sizeof(int), c->stream));
    }
    for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
        const int64_t nr = std::min(chunk, rows - r0);
        bool have_h = false, have_g = grad_preloaded;
        RowCert cert;
        if (global_cert) { cert.flags = (const int *)c->certflag.p; cert.rows = 0x7fffffff; cert.split = 0; }
        for (const RowSide *sd : {&s1, &s2}) {
            if (!sd->active) continue;
            RowHessArgs a;
            memset(&a, 0, sizeof a);
            a.O = sd->O; a.F = c->F[which];
            a.scale = (float)sd->scale; a.link = sd->link;
            if (sd->cls) {
                // one launch forms the class images of the chunk's groups, a second one adds up each row's classes
                const int R = sd->cls, NC1 = (1 << R) - 1;
                const int64_t g0 = r0 / R, ng = (nr + R - 1) / R;
                CHK(ensure(c, c->hclass, (size_t)((chunk + R - 1) / R) * NC1 * class_image_floats(c, sd->scale) * sizeof(float)));
                a.idx = (const int32_t *)c->cls_idx[sd->slot].p;
                a.cls_off = (const int64_t *)c->cls_off[sd->slot].p + g0 * NC1;
                a.cls_cnt = (const int32_t *)c->cls_cnt[sd->slot].p + g0 * NC1;
                a.T = c->F[which]; a.t_row = 0; a.t_col = 0; // never used: the gradient of a class side comes from GEMMs
                a.H = (float *)c->hclass.p; a.G = nullptr; a.accumulate = 0; a.row0 = 0; a.nrows = ng * NC1;
                a.kvalid = c->k;
                // k_pad = 256 with the single-image symmetric kernel: class images hold their 36 upper blocks only
                const bool upper = class_images_upper(c, sd->scale);
                a.cls_upper = upper ? 1 : 0; a.cls_nc1 = NC1;
                CHK(launch_row_hess(c, a, ng * NC1, (double)nr * (double)sd->per));
                {
                    Timed tm(c, CMF_K_ELEMWISE);
                    if (upper) {
                        // certificates: the samples common to all rows of half a group form a positive semi-definite part of
                        // each of those rows' Hessians (every other contribution is one too when the weights are >= 0).  If
                        // part + diag I already exceeds the perturbation threshold, so does each row: one test for the half
                        // group instead of one per row.  Worth it when the part has enough samples to be well conditioned.
                        const int split = (R + 1) / 2;
                        const bool want_cert = c->hess_psd && c->opt_rowcert && !cert.flags && diag >= 0.0 && !c->opt_choldiag &&
                                               (double)sd->n * std::pow((double)sd->per / (double)sd->n, split) >= 4.0 * c->k;
                        float *certimg = nullptr;
                        if (want_cert) {
                            CHK(ensure(c, c->certimg, (size_t)((chunk + R - 1) / R) * 2 * kk * sizeof(float)));
                            CHK(ensure(c, c->certflag, (size_t)((chunk + R - 1) / R) * 2 * sizeof(int)));
                            certimg = (float *)c->certimg.p;
                        }
                        const unsigned grid = (unsigned)std::min<int64_t>(ng * 36, (int64_t)c->num_cu * 16);
#define CMF_CLASS_SUM(NB_)                                                                                                     \
    hipLaunchKernelGGL(class_sum_blocks_kernel<NB_>, dim3(grid), dim3(256), 0, c->stream, Hc, (const float *)c->hclass.p,      \
                       have_h ? nullptr : S, have_h ? 0.f : (float)diag, nr, R, c->k, have_h ? 1 : 0, certimg, split)
                        if (c->opt_class_depth >= 16) CMF_CLASS_SUM(16);
                        else if (c->opt_class_depth >= 8) CMF_CLASS_SUM(8);
                        else CMF_CLASS_SUM(4);
#undef CMF_CLASS_SUM
                        if (want_cert) {
                            cert.flags = (const int *)c->certflag.p; cert.rows = R; cert.split = split;
                        }
                    } else {
                        const int64_t total = nr * kk / 4;
                        const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)c->num_cu * 32);
                        hipLaunchKernelGGL(class_sum_kernel, dim3(grid), dim3(256), 0, c->stream, Hc, (const float *)c->hclass.p, have_h ? nullptr : S,
                                           have_h ? 0.f : (float)diag, nr, r0, R, c->kp, c->k, have_h ? 1 : 0);
                    }
                    HIPCHK(hipGetLastError());
                }
                have_h = true;
                continue;
            }
            a.idx = sd->lists; a.idx_stride = sd->per; a.s = (int)sd->per;
            a.T = sd->T; a.t_row = sd->t_row; a.t_col = sd->t_col;
            a.H = Hc; a.G = grad; a.accumulate = 0; a.row0 = r0; a.nrows = nr;
            a.kvalid = c->k;
            // few rows with long lists (the Z sweep of a 64-column Y over 1e5 rows of V: 64 workgroups on 256 CUs): split every
            // row's samples over several workgroups that write partial sums, added up in chunk order afterwards
            int nsplit = 1;
            if (nr < c->num_cu && sd->per >= 4096 && !(c->opt_arith == 1 && c->kp == 256) && c->opt_rowsplit)
                nsplit = (int)std::min<int64_t>((2 * c->num_cu + nr - 1) / nr, sd->per / 1024);
            if (nsplit > 1) {
                const int split_len = (int)rup((sd->per + nsplit - 1) / nsplit, 32);
                nsplit = (int)((sd->per + split_len - 1) / split_len);
                CHK(ensure(c, c->hpart, ((size_t)nsplit * nr * kk + (size_t)nsplit * rows_pad * c->kp) * sizeof(float)));
                float *Hp = (float *)c->hpart.p, *Gp = Hp + (size_t)nsplit * nr * kk;
                a.H = Hp; a.G = Gp; a.accumulate = 0; a.S = nullptr; a.diag = 0.f;
                a.nsplit = nsplit; a.split_len = split_len; a.g_split_stride = rows_pad * c->kp;
                CHK(launch_row_hess(c, a, nr * nsplit));
                CHK(sum_slabs(c, Hc, Hp, nr * kk, nsplit, nr * kk, have_h));
                if (!have_h) CHK(launch_ew(c, hessian_finalize_kernel, nr * kk, Hc, S, (float)diag, nr, c->kp, c->k, 1));
                CHK(sum_slabs(c, grad + r0 * c->kp, Gp + r0 * c->kp, nr * c->kp, nsplit, rows_pad * c->kp, have_g));
                have_h = have_g = true;
                continue;
            }
            // H and G accumulate independently: encode as two flags in one int (bit0: H, bit1: G)
            a.accumulate = (have_h ? 1 : 0) | (have_g ? 2 : 0);
            a.S = S; a.diag = (float)diag; // folded into the epilogue of the launch that starts H_i
            CHK(launch_row_hess(c, a, nr));
            have_h = have_g = true;
        }
        if (!have_h) CHK(launch_ew(c, hessian_finalize_kernel, nr * kk, Hc, S, (float)diag, nr, c->kp, c->k, 0));
        for (const RowSide *sd : {&s1, &s2})
            if (sd->active && sd->sp) CHK(sparse_target_term(c, *sd, grad + r0 * c->kp, r0, nr));
        CHK(launch_ew(c, newton_grad_kernel, nr * c->kp, grad + r0 * c->kp, (const float *)(grad + r0 * c->kp), 1.0f, (const float *)nullptr, 0.f,
                      (const float *)(c->F[which] + r0 * c->kp), (float)l1, (float)l2, nr * c->kp));
        if (cert.flags && !global_cert) { // test the certificates (threshold test only): part + diag I - pert I positive definite?
            Timed tm(c, CMF_K_EIGEN);
            const int64_t ncert = 2 * ((nr + cert.rows - 1) / cert.rows);
            if (c->kp == 256 && c->opt_chol_mfma) {
                CHK(allow_big_lds(c, reinterpret_cast<const void *>(&chol_solve_mfma_kernel), (int)CholMfma::LDS_BYTES));
                hipLaunchKernelGGL(chol_solve_mfma_kernel, dim3((unsigned)ncert), dim3(256), CholMfma::LDS_BYTES, c->stream, (const float *)c->certimg.p,
                                   (const float *)nullptr, (float *)nullptr, (int *)c->certflag.p, c->k, c->kp, kk, (float)(pert - diag), (int)ncert,
                                   (const int *)nullptr, (const int *)nullptr, 1, 0, (float *)nullptr);
            } else
            hipLaunchKernelGGL((chol_solve_kernel<16>), dim3((unsigned)ncert), dim3(256), 0, c->stream, (const float *)c->certimg.p, (const float *)nullptr,
                               (float *)nullptr, (int *)c->certflag.p, c->k, c->kp, kk, (float)(pert - diag), (int)ncert, 1);
            HIPCHK(hipGetLastError());
        }
        CHK(safe_solve_rows(c, Hc, grad + r0 * c->kp, step + r0 * c->kp, nr, c->k, c->kp, pert, cert, true, c->F[which] + r0 * c->kp));
        CHK(refine_rows64(c, which, s1, s2, shared, shside, diag, l1, l2, r0, step + r0 * c->kp, pert));
    }
    return launch_ew(c, newton_apply_kernel, rows_pad * c->kp, c->F[which], (const float *)step, rows, c->kp, c->k, rows_pad * c->kp,
                     nn ? 1 : 0);
}

static int sweep_side_fused(cmf_ctx *c, bool is_u, int link, double scale, double l1, double l2, double pert, bool nn,
                            const int32_t *idx, int64_t per, bool sampled) {
    const int which = is_u ? CMF_U : CMF_Z;
    RowSide sd;
    sd.active = true;
    sd.O = c->F[CMF_V];
    sd.per = sampled ? per : c->d;
    const int dw = is_u ? 0 : 1;
    const bool native = (dw == 0 ? c->X : c->Y) == nullptr && c->sparse[dw]; // the data of this sweep stays CSR
    if (sampled) CHK(sample_lists(c, c->lists1, c->mask1, idx, c->frows[which], per, c->d, is_u ? 0 : 1, &sd.lists,
                                  native ? &c->lists1s : nullptr, native ? &sd.sorted : nullptr));
    if (is_u) { sd.T = c->X; sd.t_row = c->dp; sd.t_col = 1; }
    else { sd.T = c->Y; sd.t_row = 1; sd.t_col = c->pp; }
    sd.scale = scale; sd.link = link;
    sd.n = c->d; sd.cls = class_group_rows(c, link, sampled, per, c->d);
    // U rows pair with rows of X, Z rows with rows of Y^T
    if (native) CHK(zero_targets(c, sd, &c->sp[dw][is_u ? 0 : 1]));
    if (sd.cls) {
        CHK(build_class_lists(c, sd, c->frows[which]));
        CHK(class_side_gradient(c, is_u, is_u, sd, c->frows[which], scale, which, false));
    }
    // the logit Hessian of U carries no l2 term (:427-428); Z always does (:501-506)
    const double diag = (link == CMF_LINK_LOGIT && (is_u || !c->opt_zlogit_l2)) ? 0.0 : l2;
    return fused_rows_finish(c, which, sd, RowSide(), nullptr, diag, sd.cls != 0, l1, l2, pert, nn);
}

static int sweep_v_fused(cmf_ctx *c, double alpha, double l1, double l2, int x_link, int y_link, double pert, bool nn,
                         const int32_t *vx_idx, int64_t per_x, const int32_t *vy_idx, int64_t per_y, bool sampled) {
    const bool x_shared = (x_link == CMF_LINK_LINEAR && !sampled);
    const bool y_shared = (y_link == CMF_LINK_LINEAR && !sampled);
    RowSide sx, sy, shs;
    SharedPart shp;
    const float *S = nullptr;
    bool preloaded = false;
    float *V = c->F[CMF_V];
    if (x_shared || y_shared) {
        // shared side: gradient part s (V G - T-product) and Hessian part s G, like the shared sweep
        const bool xs = x_shared;
        const double sc = xs ? alpha : 1.0 - alpha;
        const float *Fo = xs ? c->F[CMF_U] : c->F[CMF_Z];
        const int64_t orow = xs ? c->mp : c->pp;
        CHK(gemm(c, MODE_TN, Fo, c->kp, Fo, c->kp, c->G, c->kp, c->kp, orow));            // Gram of the other factor
        CHK(data_times(c, xs ? 0 : 1, xs, Fo, c->num));                                     // X^T U  or  Y Z
        CHK(gemm(c, MODE_NN, V, c->kp, c->G, c->kp, c->den, c->dp, c->kp, c->kp));          // V G
        CHK(launch_ew(c, axpby_kernel, c->dp * c->kp, c->num, (const float *)c->den, (float)sc, (const float *)c->num, (float)-sc,
                      c->dp * c->kp));
        CHK(launch_ew(c, axpby_kernel, (int64_t)c->kp * c->kp, c->Hm, (const float *)c->G, (float)sc, (const float *)nullptr, 0.f,
                      (int64_t)c->kp * c->kp));
        S = c->Hm;
        shp.F = Fo; shp.rows_pad = orow; shp.scale = sc;
        // the same side as the float64 refinement sees it: an unsampled linear per-row side (gradient only)
        shs.active = true; shs.O = Fo; shs.per = xs ? c->m : c->p; shs.scale = sc; shs.link = CMF_LINK_LINEAR;
        if (xs) { shs.T = c->X; shs.t_row = 1; shs.t_col = c->dp; }
        else { shs.T = c->Y; shs.t_row = c->pp; shs.t_col = 1; }
        if ((xs ? c->X : c->Y) == nullptr && c->sparse[xs ? 0 : 1]) CHK(zero_targets(c, shs, xs ? &c->sp[0][1] : &c->sp[1][0]));
        preloaded = true;
    }
    if (!x_shared) {
        sx.active = true; sx.O = c->F[CMF_U];
        sx.per = sampled ? per_x : c->m;
        const bool native = c->X == nullptr && c->sparse[0];
        if (sampled) CHK(sample_lists(c, c->lists1, c->mask1, vx_idx, c->d, per_x, c->m, 2, &sx.lists, native ? &c->lists1s : nullptr,
                                      native ? &sx.sorted : nullptr));
        sx.T = c->X; sx.t_row = 1; sx.t_col = c->dp; sx.scale = alpha; sx.link = x_link;
        sx.n = c->m; sx.slot = 0; sx.cls = class_group_rows(c, x_link, sampled, per_x, c->m);
        if (native) CHK(zero_targets(c, sx, &c->sp[0][1])); // V row i pairs with column i of X = row i of X^T
        if (sx.cls) {
            CHK(build_class_lists(c, sx, c->d));
            CHK(class_side_gradient(c, true, false, sx, c->d, alpha, CMF_V, preloaded));
            preloaded = true;
        }
    }
    if (!y_shared) {
        sy.active = true; sy.O = c->F[CMF_Z];
        sy.per = sampled ? per_y : c->p;
        const bool native = c->Y == nullptr && c->sparse[1];
        if (sampled) CHK(sample_lists(c, c->lists2, c->mask2, vy_idx, c->d, per_y, c->p, 3, &sy.lists, native ? &c->lists2s : nullptr,
                                      native ? &sy.sorted : nullptr));
        sy.T = c->Y; sy.t_row = c->pp; sy.t_col = 1; sy.scale = 1.0 - alpha; sy.link = y_link;
        sy.n = c->p; sy.slot = 1; sy.cls = class_group_rows(c, y_link, sampled, per_y, c->p);
        if (native) CHK(zero_targets(c, sy, &c->sp[1][0])); // V row i pairs with row i of Y
        if (sy.cls) {
            CHK(build_class_lists(c, sy, c->d));
            CHK(class_side_gradient(c, false, true, sy, c->d, 1.0 - alpha, CMF_V, preloaded));
            preloaded = true;
        }
    }
    return fused_rows_finish(c, CMF_V, sx, sy, S, l2, preloaded, l1, l2, pert, nn, shp, shs.active ? &shs : nullptr);
}

// V sweep with a shared X side (linear link, no sampling) and a per-row Y side of FEWER samples than components (p <= 64 < k),
// every H_i certified above the threshold by l2 >= pert: the Woodbury form of cmf_eigen.hip.h ("low-rank per-row side") -- five
// d x k x k-sized products and one p x p solve per row instead of a k x k Hessian and factorisation per row.  Same step as
// sweep_v_fused up to float32 rounding (pycmf/cmf_solvers.py:432-486; the reference's own sparse Newton shape: 6 label columns,
// samples/toxic_comments.ipynb).  *done = false: a precondition failed at run time (the shared part dipped under the threshold),
// the caller takes the general path.
static int sweep_v_lowrank(cmf_ctx *c, double alpha, double l1, double l2, int y_link, double pert, bool nn, const int32_t *vy_idx,
                           int64_t per_y, bool sampled, bool *done) {
    *done = false;
    const int kp = c->kp;
    const int64_t dp = c->dp, pp = c->pp, d = c->d;
    const int p = (int)c->p, pk = p <= 32 ? 32 : 64;
    float *V = c->F[CMF_V], *U = c->F[CMF_U], *Z = c->F[CMF_Z];
    // S = alpha U^T U + l2 I in float64, S^-1 (plain inverse expected)
    CHK(ensure_shared64(c));
    CHK(gram64(c, U, c->mp, (double *)c->g64a.p, c->G));                         // U^T U: float64 + float32 copy (c->G)
    CHK(launch_hess64(c, (const double *)c->g64a.p, alpha, nullptr, 0.0, l2));
    bool plain = false;
    CHK(shared_inverse(c, pert, &plain));
    if (!plain && c->k > 64) return CMF_OK;                                        // clamped shared part: general path
    if (c->k <= 64) return CMF_OK;                                                 // p < k <= 64: nothing to gain, and `plain` is unknown on the host
    // workspaces: B (pp x kp), Zt (kp x pp), K (pk^2), M (d x pk^2), rhs / y (d x pk), images R, W, b, Q (dp x pp)
    const size_t img = (size_t)dp * pp * sizeof(float);
    CHK(ensure(c, c->lr_small, ((size_t)pp * kp * 2 + (size_t)pk * pk) * sizeof(float)));
    float *B = (float *)c->lr_small.p, *Zt = B + pp * kp, *K = Zt + pp * kp;
    const int64_t lr_chunk = std::min<int64_t>(d, c->opt_rowchunk > 0 ? rup(c->opt_rowchunk, 4) : 262144);                         // rows per batch of p x p systems (4.3 GB at pk = 64)
    const bool in_regs = c->opt_lowrank == 2;   // A/B: one wave per system in registers (measured slower: cmf_eigen.hip.h)
    CHK(ensure(c, c->lr_rows, in_regs ? (size_t)256 : (size_t)lr_chunk * pk * (pk + 2) * sizeof(float)));
    float *M = (float *)c->lr_rows.p, *rhs = M + (size_t)lr_chunk * pk * pk, *y = rhs + (size_t)lr_chunk * pk;
    CHK(ensure(c, c->resid, img));
    CHK(ensure(c, c->resid2, img));
    CHK(ensure(c, c->resid3, img));
    float *R = (float *)c->resid.p, *W = (float *)c->resid2.p, *bq = (float *)c->resid3.p;
    CHK(factor_times_hinv(c, Z, pp, 1.0, B));                                     // B = Z S^-1
    {
        Timed tm(c, CMF_K_ELEMWISE);
        hipLaunchKernelGGL(lowrank_k_kernel, dim3((unsigned)pk), dim3(256), 0, c->stream, (const float *)B, (const float *)Z, K, p, pk, kp);
        hipLaunchKernelGGL(transpose_small_kernel, dim3(64), dim3(256), 0, c->stream, (const float *)Z, Zt, (int)pp, kp, kp, (int)pp);
        HIPCHK(hipGetLastError());
    }
    // gradient: shared X part alpha (V G_U - X^T U), per-row Y part R Z with R = (1 - alpha) m o (f(V Z^T) - Y), regularisation
    CHK(data_times(c, 0, true, U, c->num));                                         // X^T U
    CHK(gemm(c, MODE_NN, V, kp, c->G, kp, c->den, dp, kp, kp));                     // V G_U
    CHK(launch_ew(c, axpby_kernel, dp * kp, c->num, (const float *)c->den, (float)alpha, (const float *)c->num, (float)-alpha, dp * kp));
    const uint8_t *my = nullptr;
    if (sampled) { // list i = columns (over p) of Y row i
        CHK(build_mask(c, c->mask2, dp, pp, vy_idx, d, per_y, true, c->p, 3));
        my = (const uint8_t *)c->mask2.p;
    }
    CHK(residual_images(c, false, y_link, 1.0 - alpha, my, R, W, y_link == CMF_LINK_LOGIT));
    CHK(gemm(c, MODE_NN, R, pp, Z, kp, c->num, dp, kp, pp, true));                  // += R Z
    CHK(launch_ew(c, newton_grad_kernel, dp * kp, c->num, (const float *)c->num, 1.0f, (const float *)nullptr, 0.f, (const float *)V,
                  (float)l1, (float)l2, dp * kp));
    CHK(gemm(c, MODE_NN, c->num, kp, c->Hinv, kp, c->den, dp, kp, kp));             // a = g S^-1
    CHK(gemm(c, MODE_NN, c->den, kp, Zt, pp, bq, dp, pp, kp));                      // b = a Z^T
    {
        Timed tm(c, CMF_K_EIGEN);
        CHK(ensure(c, c->eigflag, (size_t)lr_chunk * sizeof(int)));
        CHK(ensure(c, c->certflag, 2 * sizeof(int)));
        HIPCHK(hipMemsetAsync(c->certflag.p, 0, 2 * sizeof(int), c->stream));
        if (in_regs) { // one wave per row, the p x p system in registers
            if (pk == 32) hipLaunchKernelGGL((lowrank_solve_kernel<32>), dim3((unsigned)((d + 3) / 4)), dim3(256), 0, c->stream, (const float *)W, bq, pp, (const float *)K, d, p);
            else hipLaunchKernelGGL((lowrank_solve_kernel<64>), dim3((unsigned)((d + 3) / 4)), dim3(256), 0, c->stream, (const float *)W, bq, pp, (const float *)K, d, p);
            HIPCHK(hipGetLastError());
        } else // default: the systems through memory and chol_solve_kernel
        for (int64_t r0 = 0; r0 < d; r0 += lr_chunk) {
            const int64_t nr = std::min(lr_chunk, d - r0);
            hipLaunchKernelGGL(lowrank_build_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, c->stream, (const float *)(W + r0 * pp),
                               (const float *)(bq + r0 * pp), pp, (const float *)K, M, rhs, nr, p, pk);
            // I + sqrt(C) K sqrt(C) is positive definite with lambda_min >= 1: certified, one factorisation + solve per row
            const dim3 grid((unsigned)nr), block(256);
            if (pk == 32) hipLaunchKernelGGL((chol_solve_kernel<2>), grid, block, 0, c->stream, (const float *)M, (const float *)rhs, y, (int *)c->eigflag.p, p, pk,
                                             (int64_t)pk * pk, 0.0f, (int)nr, 0, (const int *)nullptr, 1, (const int *)c->certflag.p, 0x7fffffff, 0);
            else hipLaunchKernelGGL((chol_solve_kernel<4>), grid, block, 0, c->stream, (const float *)M, (const float *)rhs, y, (int *)c->eigflag.p, p, pk,
                                    (int64_t)pk * pk, 0.0f, (int)nr, 0, (const int *)nullptr, 1, (const int *)c->certflag.p, 0x7fffffff, 0);
            // the b image is consumed row block by row block: its rows become sqrt(C) y in place
            hipLaunchKernelGGL(lowrank_scale_kernel, dim3((unsigned)std::min<int64_t>((nr * pp + 255) / 256, (int64_t)c->num_cu * 32)), dim3(256), 0, c->stream,
                               (const float *)(W + r0 * pp), (const float *)y, bq + r0 * pp, pp, nr, p, pk);
            HIPCHK(hipGetLastError());
        }
        if (dp > d) HIPCHK(hipMemsetAsync(bq + d * pp, 0, (size_t)(dp - d) * pp * sizeof(float), c->stream));
    }
    CHK(gemm(c, MODE_NN, bq, pp, B, kp, c->num, dp, kp, pp));                        // (sqrt(C) y)^T B
    CHK(launch_ew(c, axpby_kernel, dp * kp, c->den, (const float *)c->den, 1.0f, (const float *)c->num, -1.0f, dp * kp)); // step
    CHK(launch_ew(c, newton_apply_kernel, dp * kp, V, (const float *)c->den, d, kp, c->k, dp * kp, nn ? 1 : 0));
    *done = true;
    return CMF_OK;
}

static int newton_step_impl(cmf_ctx *c, double alpha, double l1, double l2, int x_link, int y_link, int nn_mask, int upd,
                            double pert, double ratio, const int32_t *u_idx, const int32_t *z_idx, const int32_t *vx_idx,
                            const int32_t *vy_idx);

extern "C" int cmf_newton_step(cmf_ctx *c, double alpha, double l1, double l2, int x_link, int y_link, int nn_mask, int upd,
                               double pert, double ratio, const int32_t *u_idx, const int32_t *z_idx, const int32_t *vx_idx,
                               const int32_t *vy_idx) {
    NEED_PROBLEM(c);
    if (x_link == CMF_LINK_LINEAR && y_link == CMF_LINK_LINEAR && ratio >= 1.0 && have_data(c, 0) && have_data(c, 1) &&
        c->opt_graph > 0 && use_shared64(c) && c->k <= 64) {
        // all three sweeps take the shared-Hessian form and the float64 inverse of k <= 64 decides everything on the device:
        // a fixed launch sequence with no host round trip -> graph replay.  (Larger k read two flags back per inverse:
        // those steps are not captured.)
        DeviceGuard dg(c->device);
        const double key[6] = {alpha, l1, l2, (double)nn_mask, (double)upd, pert};
        return run_graphed(c, c->newton_graph, key, 6, [&]() {
            return newton_step_impl(c, alpha, l1, l2, x_link, y_link, nn_mask, upd, pert, ratio, nullptr, nullptr, nullptr, nullptr);
        });
    }
    return newton_step_impl(c, alpha, l1, l2, x_link, y_link, nn_mask, upd, pert, ratio, u_idx, z_idx, vx_idx, vy_idx);
}

static int newton_step_impl(cmf_ctx *c, double alpha, double l1, double l2, int x_link, int y_link, int nn_mask, int upd,
                            double pert, double ratio, const int32_t *u_idx, const int32_t *z_idx, const int32_t *vx_idx,
                            const int32_t *vy_idx) {
    NEED_PROBLEM(c);
    if ((x_link != 0 && x_link != 1) || (y_link != 0 && y_link != 1)) return fail(CMF_EINVAL, "bad link id");
    DeviceGuard dg(c->device);
    const bool sampled = ratio < 1.0;
    struct FlopScope { // sampled sweeps execute masked-dense GEMMs: credit only the sampled share
        cmf_ctx *c;
        ~FlopScope() { c->flop_scale = 1.0; }
    } fscope{c};
    struct SamplingScope { // device sampling is armed by cmf_newton_step_device_sampled only
        cmf_ctx *c; bool keep;
        ~SamplingScope() { if (!keep) c->dev_sampling = false; }
    } scope{c, false};
    if (!sampled) c->dev_sampling = false;
    c->hess_psd = (alpha >= 0.0 && alpha <= 1.0 && l2 >= 0.0); // Gram-like sums with non-negative weights
    const bool fused = c->opt_rowkernel && c->kp <= 256; // fused gather kernel vs masked-dense GEMMs
    const int64_t su = (int64_t)((double)c->d * ratio);  // int(n * ratio), cmf_solvers.py:331
    const int64_t sm = (int64_t)((double)c->m * ratio), sp = (int64_t)((double)c->p * ratio);
    if (sampled && !c->dev_sampling) {
        if (((upd & CMF_UPD_U) && !u_idx) || ((upd & CMF_UPD_Z) && !z_idx) || ((upd & CMF_UPD_V) && (!vx_idx || !vy_idx)))
            return fail(CMF_EINVAL, "sg_sample_ratio < 1 needs the sample index lists of every updated factor");
    }
    bool vgram = false; // V^T V (float64) left behind by a shared U sweep for the Z sweep that follows (V unchanged)
    if (upd & CMF_UPD_U) {
        if (!have_data(c, 0)) return fail(CMF_EINVAL, "X must be set before a U update");
        // the fused per-row sweeps keep a natively sparse side sparse (zero-target row kernel + sparse target term)
        if (!(x_link == CMF_LINK_LINEAR && !sampled) && !fused) CHK(need_dense(c, 0));
        if (x_link == CMF_LINK_LINEAR && !sampled) {
            CHK(sweep_side_shared(c, true, alpha, l1, l2, pert, (nn_mask & CMF_NN_U) != 0));
            vgram = use_shared64(c);
        }
        else if (fused) {
            CHK(sweep_side_fused(c, true, x_link, alpha, l1, l2, pert, (nn_mask & CMF_NN_U) != 0, u_idx, su, sampled));
        } else {
            c->flop_scale = sampled ? ratio : 1.0;
            CHK(sweep_side_rows(c, true, x_link, alpha, l1, l2, pert, (nn_mask & CMF_NN_U) != 0, (sampled && !c->dev_sampling) ? u_idx : nullptr, su));
            c->flop_scale = 1.0;
        }
    }
    if (upd & CMF_UPD_Z) {
        if (!have_data(c, 1)) return fail(CMF_EINVAL, "Y must be set before a Z update");
        if (!(y_link == CMF_LINK_LINEAR && !sampled) && !fused) CHK(need_dense(c, 1));
        if (y_link == CMF_LINK_LINEAR && !sampled)
            CHK(sweep_side_shared(c, false, 1.0 - alpha, l1, l2, pert, (nn_mask & CMF_NN_Z) != 0, vgram));
        else if (fused) {
            CHK(sweep_side_fused(c, false, y_link, 1.0 - alpha, l1, l2, pert, (nn_mask & CMF_NN_Z) != 0, z_idx, su, sampled));
        } else {
            c->flop_scale = sampled ? ratio : 1.0;
            CHK(sweep_side_rows(c, false, y_link, 1.0 - alpha, l1, l2, pert, (nn_mask & CMF_NN_Z) != 0,
                                (sampled && !c->dev_sampling) ? z_idx : nullptr, su));
            c->flop_scale = 1.0;
        }
    }
    if (upd & CMF_UPD_V) {
        if (!have_data(c, 0) || !have_data(c, 1)) return fail(CMF_EINVAL, "X and Y must be set before a V update");
        // a natively sparse side stays sparse in the fused sweeps: a shared part (linear link, no sampling) is served from the
        // SpMM gradient + the shared Gram, a per-row part by the zero-target row kernel + the sparse target term -- the
        // reference's own sparse Newton workloads (samples/toxic_comments.ipynb:853-856, benchmarks/benchmark_cmf.py:72-82;
        // cmf_solvers.py:328-344, :432-486)
        {
            const bool xs = (x_link == CMF_LINK_LINEAR && !sampled), ys = (y_link == CMF_LINK_LINEAR && !sampled);
            if (!(xs && ys) && !fused) {
                CHK(need_dense(c, 0));
                CHK(need_dense(c, 1));
            }
        }
        if (x_link == CMF_LINK_LINEAR && y_link == CMF_LINK_LINEAR && !sampled && use_reassoc(c)) {
            CHK(ensure_shared64(c));
            CHK(cmf_newton_v_gram(c, alpha, (double *)c->gmix64.p));
            CHK(cmf_newton_v_products(c, alpha, l2, pert, (const double *)c->gmix64.p, c->num));
            CHK(cmf_newton_v_finish(c, c->num, l1, nn_mask));
        } else if (x_link == CMF_LINK_LINEAR && y_link == CMF_LINK_LINEAR && !sampled) {
            CHK(cmf_newton_v_partials(c, alpha, c->vbuf));
            c->gmix64_valid = use_shared64(c); // no collective between the two halves: keep the float64 Gram mix
            CHK(cmf_newton_v_apply(c, c->vbuf, l1, l2, nn_mask, pert));
        } else if (fused) {
            bool done = false;
            // shared X side + per-row Y side of fewer samples than components, all H_i certified (l2 >= pert): Woodbury form
            if (c->opt_lowrank && x_link == CMF_LINK_LINEAR && !sampled && c->p <= 64 && c->p < c->k && c->k > 64 && c->Y && c->hess_psd &&
                c->opt_rowcert && l2 >= pert && use_shared64(c))
                CHK(sweep_v_lowrank(c, alpha, l1, l2, y_link, pert, (nn_mask & CMF_NN_V) != 0, vy_idx, sp, sampled, &done));
            if (!done)
            CHK(sweep_v_fused(c, alpha, l1, l2, x_link, y_link, pert, (nn_mask & CMF_NN_V) != 0, vx_idx, sm, vy_idx, sp, sampled));
        } else {
            c->flop_scale = sampled ? ratio : 1.0;
            CHK(sweep_v_rows(c, alpha, l1, l2, x_link, y_link, pert, (nn_mask & CMF_NN_V) != 0,
                             (sampled && !c->dev_sampling) ? vx_idx : nullptr, sm, (sampled && !c->dev_sampling) ? vy_idx : nullptr, sp));
        }
    }
    return CMF_OK;
}

// Same step with the per-row samples drawn on the device from a counter-based generator keyed by
// (seed, sweep, row): statistically equivalent to the reference's sampler (exactly int(n*ratio)
// distinct candidates per row, uniform), not stream-identical to NumPy's MT19937.
extern "C" int cmf_newton_step_device_sampled(cmf_ctx *c, double alpha, double l1, double l2, int x_link, int y_link, int nn_mask,
                                              int upd, double pert, double ratio, uint64_t seed) {
    NEED_PROBLEM(c);
    c->dev_sampling = ratio < 1.0;
    c->dev_seed = seed;
    return cmf_newton_step(c, alpha, l1, l2, x_link, y_link, nn_mask, upd, pert, ratio, nullptr, nullptr, nullptr, nullptr);
}

// The index lists the device sampler draws for `nrows` consecutive rows of one sweep (sweep 0: U rows, lists over d; 1: Z rows,
// over d; 2: V rows, X side, over m; 3: V rows, Y side, over p) under `seed` -- exactly what
// cmf_newton_step_device_sampled(seed) uses for those rows, ascending inside a list.  For parity tests that recompute single
// rows in float64 at sizes where the whole sweep cannot be (pycmf/cmf_solvers.py:328-344 is the sampler being replaced).
extern "C" int cmf_sample_lists(cmf_ctx *c, int sweep, uint64_t seed, double ratio, int64_t row0, int64_t nrows, int32_t *out) {
    NEED_PROBLEM(c);
    if (sweep < 0 || sweep > 3 || !out || nrows < 0 || row0 < 0 || !(ratio > 0.0 && ratio < 1.0)) return fail(CMF_EINVAL, "bad argument");
    DeviceGuard dg(c->device);
    const int64_t n = (sweep <= 1) ? c->d : (sweep == 2 ? c->m : c->p);
    const int64_t per = (int64_t)((double)n * ratio);
    if (nrows * per == 0) return CMF_OK;
    int32_t *dl = nullptr;
    HIPCHK(hipMalloc((void **)&dl, (size_t)nrows * per * sizeof(int32_t)));
    const int64_t off = c->sample_off[sweep == 0 ? CMF_U : (sweep == 1 ? CMF_Z : CMF_V)] + row0;
    hipLaunchKernelGGL(sample_select_kernel, dim3((unsigned)nrows), dim3(256), 0, c->stream, (uint8_t *)nullptr, (int64_t)0, 1, dl, nrows, (int)n,
                       (int)per, seed * 4 + (uint64_t)sweep, off);
    int rc = CMF_OK;
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out, dl, (size_t)nrows * per * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        rc = fail(CMF_EHIP, "cmf_sample_lists failed");
    (void)hipFree(dl);
    return rc;
}

// ---- generic "data times a host matrix" product (initialisers: randomized range finder) --------------
// out (host, float64, rows_out x ncols) = op(A) * B,  A = X (which 0) or Y (which 1), op = transpose if trans.
extern "C" int cmf_data_matmul_f64(cmf_ctx *c, int which, int trans, const double *B, int64_t b_rows, int ncols, double *out) {
    NEED_PROBLEM(c);
    if ((which != 0 && which != 1) || !B || !out || ncols <= 0) return fail(CMF_EINVAL, "bad argument");
    if (!have_data(c, which)) return fail(CMF_EINVAL, "%s has not been set", which == 0 ? "X" : "Y");
    DeviceGuard dg(c->device);
    const int64_t ar = which == 0 ? c->m : c->d, ac = which == 0 ? c->d : c->p;       // A is ar x ac
    const int64_t arp = which == 0 ? c->mp : c->dp, acp = which == 0 ? c->dp : c->pp;
    const int64_t need_rows = trans ? ar : ac, out_rows = trans ? ac : ar;
    const int64_t need_pad = trans ? arp : acp, out_pad = trans ? acp : arp;
    if (b_rows != need_rows) return fail(CMF_EINVAL, "operand has %lld rows, expected %lld", (long long)b_rows, (long long)need_rows);
    const int np = ncols <= 32 ? 32 : ncols <= 64 ? 64 : ncols <= 128 ? 128 : (int)rup(ncols, 256);
    float *dB = nullptr, *dO = nullptr;
    HIPCHK(hipMalloc((void **)&dB, (size_t)need_pad * np * sizeof(float)));
    if (hipMalloc((void **)&dO, (size_t)out_pad * np * sizeof(float)) != hipSuccess) { (void)hipFree(dB); return fail(CMF_ENOMEM, "out of device memory"); }
    int rc = CMF_OK;
    do {
        if (hipMemsetAsync(dB, 0, (size_t)need_pad * np * sizeof(float), c->stream) != hipSuccess) { rc = fail(CMF_EHIP, "memset failed"); break; }
        rc = upload_strided<double>(c, dB, np, need_rows, ncols, B, ncols, 1);
        if (rc != CMF_OK) break;
        const float *A = which == 0 ? c->X : c->Y;
        if (!A) { // native CSR
            rc = spmm(c, c->sp[which][trans ? 1 : 0], dB, dO, out_pad, false, np);
        } else if (!trans) {
            rc = gemm(c, MODE_NN, A, acp, dB, np, dO, arp, np, acp);
        } else {
            rc = gemm(c, MODE_TN, A, acp, dB, np, dO, acp, np, arp);
        }
        if (rc != CMF_OK) break;
        std::vector<float> host((size_t)out_rows * ncols);
        if (hipMemcpy2DAsync(host.data(), ncols * sizeof(float), dO, np * sizeof(float), ncols * sizeof(float), out_rows, hipMemcpyDeviceToHost,
                             c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(CMF_EHIP, "D2H failed"); break; }
        for (size_t i = 0; i < host.size(); ++i) out[i] = (double)host[i];
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(dB);
    (void)hipFree(dO);
    return rc;
}


// ---- the whole outer loop: _IterativeCMFSolver.fit_iterative_update, pycmf/cmf_solvers.py:132-195 ---------------------------
// error at init (:170), then for n_iter = 1 .. max_iter: update_step (:172); every `check_every`-th iteration when tol > 0 the
// error (:175-176) and the stopping test (previous - error) / error_at_init < tol (:183-186).  The error is
// alpha_err ||X - f(U V^T)|| + (1 - alpha_err) ||Y - f(V Z^T)|| (:128-130; a side that is not set counts as 0) -- one 16-byte
// read-back per check, nothing else crosses the boundary while the loop runs.  The step body of MU and of the graph-capturable
// Newton configurations is replayed from a hipGraph (option "graph": -1 = inside cmf_run only, the default).
// err_trace (nullable, trace_cap entries): [0] = error at init, then one entry per check; time_trace (nullable): seconds since
// the call started at the same points (what the reference's verbose lines print, :178-181).
// One MU iteration AND the error metric of the factors it leaves (compute_factorization_error, pycmf/cmf_solvers.py:36-42, linear
// link) from what the iteration forms anyway -- the expansion sklearn's own sparse path takes (:40):
//   ||X - U V^T||^2 = ||X||^2 - 2 <U, X V> + <U^T U, V^T V>,
// X V being the numerator of the U update (:232) taken with the NEW V, <.,.> float64 sums over float32 operands, the three k x k
// Grams on the float64 matrix pipe, ||X||^2 once per data set.  No pass over X or Y: the NT error pass costs 38.7 ms at C4 (6 % of
// a fit at the reference's default tol = 1e-4), this costs two dot products of m k elements and three Grams.  The expansion
// subtracts numbers of size ||X||^2: where the fit is close (e^2 < 1e-3 ||X||^2: float32 rounding of X V would show in the fifth
// digit of the error) -- or the side is not dense, its factor not part of this update -- the side falls back to the NT pass.
// *ex2, *ey2: squared Frobenius residuals as cmf_residual_sq returns them.
extern "C" int cmf_mu_step_error(cmf_ctx *c, double l1, double l2, int mask, double *ex2, double *ey2) {
    NEED_PROBLEM(c);
    DeviceGuard dg(c->device);
    const bool tx = ex2 && c->X && (mask & CMF_UPD_U) && !c->opt_trace_error_off, ty = ey2 && c->Y && (mask & CMF_UPD_Z) && !c->opt_trace_error_off;
    c->mu_dots = tx || ty;
    const int rc = mu_step_eager(c, l1, l2, mask);
    c->mu_dots = false;
    CHK(rc);
    double sq[2] = {-1.0, -1.0};
    if (tx || ty) {
        for (int w = 0; w < 2; ++w)
            if ((w == 0 ? tx : ty) && !c->dsq_valid[w]) {
                double x2 = 0.0, y2 = 0.0;
                CHK(cmf_data_sq(c, &x2, &y2));
                c->dsq_cache[0] = x2; c->dsq_cache[1] = y2;
                c->dsq_valid[0] = c->X != nullptr; c->dsq_valid[1] = c->Y != nullptr;
            }
        const int64_t kk = (int64_t)c->kp * c->kp;
        CHK(ensure(c, c->trace64, (size_t)3 * kk * sizeof(double)));
        double *GU = (double *)c->trace64.p, *GV = GU + kk, *GZ = GV + kk;
        CHK(gram64(c, c->F[CMF_V], c->dp, GV, nullptr));
        if (tx) CHK(gram64(c, c->F[CMF_U], c->mp, GU, nullptr));
        if (ty) CHK(gram64(c, c->F[CMF_Z], c->pp, GZ, nullptr));
        {
            Timed tm(c, CMF_K_ELEMWISE);
            if (tx) hipLaunchKernelGGL(frob_inner64_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)GU, (const double *)GV, (int)kk, c->dscalar + 4);
            if (ty) hipLaunchKernelGGL(frob_inner64_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)GZ, (const double *)GV, (int)kk, c->dscalar + 5);
            HIPCHK(hipGetLastError());
        }
        double h[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(h, c->dscalar + 2, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (tx) { const double e2 = c->dsq_cache[0] - 2.0 * h[0] + h[2]; if (e2 >= 1e-3 * c->dsq_cache[0]) sq[0] = e2; }
        if (ty) { const double e2 = c->dsq_cache[1] - 2.0 * h[1] + h[3]; if (e2 >= 1e-3 * c->dsq_cache[1]) sq[1] = e2; }
    }
    const bool nx = ex2 && sq[0] < 0.0 && have_data(c, 0), ny = ey2 && sq[1] < 0.0 && have_data(c, 1);
    if (nx || ny) {
        double a = 0.0, b = 0.0;
        CHK(cmf_residual_sq(c, CMF_LINK_LINEAR, CMF_LINK_LINEAR, nx ? &a : nullptr, ny ? &b : nullptr));
        if (nx) sq[0] = a;
        if (ny) sq[1] = b;
    }
    if (ex2) *ex2 = sq[0] < 0.0 ? 0.0 : sq[0];
    if (ey2) *ey2 = sq[1] < 0.0 ? 0.0 : sq[1];
    return CMF_OK;
}

extern "C" int cmf_run(cmf_ctx *c, const cmf_run_params *p, int max_iter, double tol, int check_every, int *n_iter_out,
                       double *err_trace, double *time_trace, int trace_cap, int *n_trace) {
    NEED_PROBLEM(c);
    if (!p || max_iter < 0 || check_every < 1) return fail(CMF_EINVAL, "cmf_run: bad argument");
    if (p->solver != CMF_SOLVER_MU && p->solver != CMF_SOLVER_NEWTON) return fail(CMF_EINVAL, "cmf_run: solver must be CMF_SOLVER_MU or CMF_SOLVER_NEWTON");
    DeviceGuard dg(c->device);
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    const bool hx = have_data(c, 0), hy = have_data(c, 1);
    int nt = 0;
    auto error = [&](double *e) -> int {
        double ex2 = 0.0, ey2 = 0.0;
        CHK(cmf_residual_sq(c, p->x_link, p->y_link, hx ? &ex2 : nullptr, hy ? &ey2 : nullptr));
        *e = p->alpha_err * (hx ? std::sqrt(ex2) : 0.0) + (1.0 - p->alpha_err) * (hy ? std::sqrt(ey2) : 0.0);
        if (nt < trace_cap) {
            if (err_trace) err_trace[nt] = *e;
            if (time_trace) time_trace[nt] = elapsed();
        }
        ++nt;
        return CMF_OK;
    };
    struct GraphScope { // "graph" = -1: replay inside the loop, eager for single steps
        cmf_ctx *c; int saved;
        ~GraphScope() { c->opt_graph = saved; }
    } gs{c, c->opt_graph};
    if (c->opt_graph < 0) c->opt_graph = 1;
    double prev = 0.0, at_init = 0.0;
    CHK(error(&at_init));
    prev = at_init;
    int it = 0;
    auto record = [&](double e) {
        if (nt < trace_cap) {
            if (err_trace) err_trace[nt] = e;
            if (time_trace) time_trace[nt] = elapsed();
        }
        ++nt;
    };
    for (it = 1; it <= max_iter; ++it) {
        if (p->solver == CMF_SOLVER_MU && tol > 0.0 && it % check_every == 0) {
            // the check iteration: the step and the error metric of its result in one call, without a pass over X and Y
            double ex2 = 0.0, ey2 = 0.0;
            CHK(cmf_mu_step_error(c, p->l1, p->l2, p->update_mask, hx ? &ex2 : nullptr, hy ? &ey2 : nullptr));
            const double e = p->alpha_err * (hx ? std::sqrt(ex2) : 0.0) + (1.0 - p->alpha_err) * (hy ? std::sqrt(ey2) : 0.0);
            record(e);
            if ((prev - e) / at_init < tol) break;
            prev = e;
            continue;
        }
        if (p->solver == CMF_SOLVER_MU) CHK(cmf_mu_step(c, p->l1, p->l2, p->update_mask));
        else if (p->sg_ratio < 1.0)
            CHK(cmf_newton_step_device_sampled(c, p->alpha, p->l1, p->l2, p->x_link, p->y_link, p->nn_mask, p->update_mask, p->hessian_pertubation,
                                               p->sg_ratio, p->seed + (uint64_t)it));
        else
            CHK(cmf_newton_step(c, p->alpha, p->l1, p->l2, p->x_link, p->y_link, p->nn_mask, p->update_mask, p->hessian_pertubation, 1.0,
                                nullptr, nullptr, nullptr, nullptr));
        if (tol > 0.0 && it % check_every == 0) {
            double e = 0.0;
            CHK(error(&e));
            if ((prev - e) / at_init < tol) break;
            prev = e;
        }
    }
    if (it > max_iter) it = max_iter; // the loop ran out: Python's `for n_iter in range(1, max_iter + 1)` leaves max_iter
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_iter_out) *n_iter_out = it;
    if (n_trace) *n_trace = nt;
    return CMF_OK;
}
