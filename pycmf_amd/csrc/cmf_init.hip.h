// cmf_init.hip.h -- device side of the initialisers (included by cmf_api.hip).
//
// Reference: _initialize_mf (pycmf/cmf.py:41-202).  'svd' and 'nndsvd*' call sklearn's randomized_svd (:126, :149), whose
// cost is a dozen passes over the data matrix plus tall-skinny orthonormalisations; 'random' needs M.mean() (:111).  Both
// run here on the copy of X / Y that the solver uses anyway:
//   * cmf_rsvd: Halko-Martinsson-Tropp range finder with power iterations; every product with the data is the solver's
//     own MFMA GEMM (or CSR SpMM), every orthonormalisation is CholeskyQR2 -- float64 Gram on the float64 matrix pipe
//     (gram64), float64 Cholesky and triangular inverse (cmf_shared64.hip.h), one GEMM with R^-1 -- instead of sklearn's
//     pivoted LU: the same subspace, no tall-skinny matrix ever leaves the device.  The small matrix B = Q^T A is
//     decomposed through its (size x size) float64 Gram, whose eigen-decomposition (cyclic Jacobi) is the only host
//     arithmetic.  The Gaussian test matrix comes from the caller (NumPy's RandomState, so that a given random_state
//     spans sklearn's subspace).
//   * cmf_data_sum: sum of all entries (the mean of 'random' / 'nndsvda' / 'nndsvdar').

__global__ __launch_bounds__(256) void sum_kernel(const float *A, int64_t n4, double *partials) {
    double v = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4 *>(A)[i];
        v += (double)a[0] + (double)a[1] + (double)a[2] + (double)a[3];
    }
    __shared__ double red[256];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

extern "C" int cmf_data_sum(cmf_ctx *c, double *sx, double *sy) {
    NEED_PROBLEM(c);
    DeviceGuard dg(c->device);
    double host[2] = {0, 0};
    HIPCHK(hipMemsetAsync(c->dscalar, 0, 2 * sizeof(double), c->stream));
    const float *src[2] = {c->X, c->Y};
    const int64_t n[2] = {c->mp * c->dp, c->dp * c->pp};
    for (int w = 0; w < 2; ++w) {
        if (!src[w]) continue;
        const int blocks = 1024;
        CHK(ensure(c, c->dpart, blocks * sizeof(double)));
        Timed tm(c, CMF_K_ELEMWISE);
        hipLaunchKernelGGL(sum_kernel, dim3(blocks), dim3(256), 0, c->stream, src[w], n[w] / 4, (double *)c->dpart.p);
        hipLaunchKernelGGL(sum_doubles_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)c->dpart.p, (int64_t)blocks, c->dscalar + w);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(host, c->dscalar, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int w = 0; w < 2; ++w)
        if (!src[w] && c->sparse[w]) host[w] = c->sp_sum[w];
    if (sx) *sx = host[0];
    if (sy) *sy = host[1];
    return CMF_OK;
}

// ---- block sums of the dense device image of X / Y in float64: the per-tile checksums of the full-size parity tests.  For a product
// P = A F the column sums of a 256-row tile of P are (sums of A over the tile's rows)^T F, for P = A^T F they are (sums of A over
// the tile's columns)^T F -- k_pad-sized float64 work on the host against every output tile of a data pass.  Plain reductions,
// nothing shared with the GEMM kernels they check.
__global__ __launch_bounds__(256) void block_sums_cols_kernel(const float *A, int64_t ld, int64_t rows, int64_t nblk, double *out) {
    // one wave per (row, block of 256 columns): out[row * nblk + t]
    const int64_t id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (id >= rows * nblk) return;
    const int64_t row = id / nblk, t = id % nblk;
    const f32x4 a = reinterpret_cast<const f32x4 *>(A + row * ld + t * 256)[threadIdx.x & 63];
    double v = (double)a[0] + (double)a[1] + (double)a[2] + (double)a[3];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) out[id] = v;
}
__global__ __launch_bounds__(256) void block_sums_rows_kernel(const float *A, int64_t ld, int64_t cols, double *out) {
    // one thread per column, 256 rows each: out[blockIdx.y * cols + col]
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    const float *a = A + (int64_t)blockIdx.y * 256 * ld + col;
    double v = 0.0;
    for (int r = 0; r < 256; ++r) v += (double)a[(int64_t)r * ld];
    out[(int64_t)blockIdx.y * cols + col] = v;
}
// axis 0: out[rows_pad / 256][cols_pad] = sums over blocks of 256 rows; axis 1: out[rows_pad][cols_pad / 256] = sums over blocks of
// 256 columns (padded extents: cmf_get_geometry; the padding is zero).  Host array of float64.
extern "C" int cmf_data_block_sums_f64(cmf_ctx *c, int which, int axis, double *out) {
    NEED_PROBLEM(c);
    if (!out || (axis != 0 && axis != 1)) return fail(CMF_EINVAL, "bad argument (axis 0: row blocks, 1: column blocks)");
    DeviceGuard dg(c->device);
    int64_t r, cc, rp, cp; float **slot;
    CHK(data_dims(c, which, &r, &cc, &rp, &cp, &slot));
    if (!*slot) return fail(CMF_EINVAL, "%s has no dense device image (native sparse input or not set)", which == 0 ? "X" : "Y");
    const int64_t n = axis == 0 ? (rp / 256) * cp : rp * (cp / 256);
    void *dev = nullptr;
    CHK(dev_alloc(c, &dev, (size_t)n * sizeof(double), false));
    if (axis == 0) hipLaunchKernelGGL(block_sums_rows_kernel, dim3((unsigned)(cp / 256), (unsigned)(rp / 256)), dim3(256), 0, c->stream, (const float *)*slot, cp, cp, (double *)dev);
    else hipLaunchKernelGGL(block_sums_cols_kernel, dim3((unsigned)((rp * (cp / 256) + 3) / 4)), dim3(256), 0, c->stream, (const float *)*slot, cp, rp, cp / 256, (double *)dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(c, dev);
    if (e != hipSuccess) return fail(CMF_EHIP, "cmf_data_block_sums_f64: %s", hipGetErrorString(e));
    return CMF_OK;
}

// ---- host: eigen-decomposition of a symmetric n x n float64 matrix by cyclic Jacobi; eigenvalues descending, eigenvectors
// in the COLUMNS of V (row-major n x n)
static void jacobi_eigh_host(std::vector<double> &A, int n, std::vector<double> &V, std::vector<double> &lam) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += A[(size_t)i * n + i] * A[(size_t)i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        }
        if (off <= 1e-30 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) { // columns p, q
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = cs * akp - sn * akq;
                    A[(size_t)k * n + q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; ++k) { // rows p, q
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = cs * apk - sn * aqk;
                    A[(size_t)q * n + k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = cs * vkp - sn * vkq;
                    V[(size_t)k * n + q] = sn * vkp + cs * vkq;
                }
            }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return A[(size_t)a * n + a] > A[(size_t)b * n + b]; });
    lam.resize(n);
    std::vector<double> Vs((size_t)n * n);
    for (int j = 0; j < n; ++j) {
        lam[j] = A[(size_t)order[j] * n + order[j]];
        for (int i = 0; i < n; ++i) Vs[(size_t)i * n + j] = V[(size_t)i * n + order[j]];
    }
    V.swap(Vs);
}

struct RsvdWork {
    float *Q = nullptr, *Y = nullptr, *T = nullptr, *S32 = nullptr; // tall-skinny iterates (rmax_pad x np) and a np x np float32 matrix
    double *G = nullptr, *H = nullptr, *W = nullptr, *Xt = nullptr; // np x np float64: Gram, shifted Gram, Cholesky factor, L^-T
    int *flag = nullptr;
};

// float64 Gram of a tall-skinny float32 matrix of width np (gram64 with an explicit width)
static int gram64_w(cmf_ctx *c, const float *F, int64_t rows_pad, int np, double *G64) {
    const int ts = np >= 64 ? 64 : 32, T = np / ts, ntile = T * (T + 1) / 2;
    int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>((4 * c->num_cu + ntile - 1) / ntile, rows_pad / 32));
    const int64_t chunk = rup((rows_pad + nsplit - 1) / nsplit, 32);
    nsplit = (rows_pad + chunk - 1) / chunk;
    CHK(ensure(c, c->gslab64, (size_t)nsplit * ntile * ts * ts * sizeof(double)));
    Timed tm(c, CMF_K_GEMM_SMALL, 2.0 * (double)rows_pad * np * np);
    const dim3 grid((unsigned)ntile, (unsigned)nsplit);
    if (ts == 64) hipLaunchKernelGGL((gram64_partial_kernel<64>), grid, dim3(256), 0, c->stream, F, np, rows_pad, chunk, (double *)c->gslab64.p);
    else hipLaunchKernelGGL((gram64_partial_kernel<32>), grid, dim3(64), 0, c->stream, F, np, rows_pad, chunk, (double *)c->gslab64.p);
    hipLaunchKernelGGL(gram64_reduce_kernel, dim3((unsigned)std::min(256, (np * np + 255) / 256)), dim3(256), 0, c->stream,
                       (const double *)c->gslab64.p, ts, np, (int)nsplit, G64, (float *)nullptr);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// C[rows_pad x np] = A[rows_pad x np] * B[np x np] (all float32, row-major, pitch np)
static int gemm_tall(cmf_ctx *c, const float *A, const float *B, float *C, int64_t rows_pad, int np) {
    return gemm(c, MODE_NN, A, np, B, np, C, rows_pad, np, np);
}

// One CholeskyQR pass: Y <- Y R^-1 with Y^T Y = R^T R (float64 Gram, Cholesky and inverse).  `size` valid columns of np.
// A rank-deficient block (fewer independent columns than `size`) is regularised by a relative diagonal shift.
static int cholqr_pass(cmf_ctx *c, RsvdWork &w, float *&Y, float *&T, int64_t rows_pad, int size, int np) {
    CHK(gram64_w(c, Y, rows_pad, np, w.G));
    const size_t kk = (size_t)np * np;
    const unsigned nb = (unsigned)std::min(256, (np * np + 255) / 256);
    double shift = 0.0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        hipLaunchKernelGGL(hess64_build_kernel, dim3(nb), dim3(256), 0, c->stream, w.H, (const double *)w.G, 1.0, (const double *)nullptr, 0.0, shift, np, size);
        if (size <= 128) hipLaunchKernelGGL((chol64_reg_kernel<4>), dim3(1), dim3(1024), 0, c->stream, (const double *)w.H, size, np, w.W, (int64_t)kk, np, 0.0, 0.0, w.flag);
        else if (size <= 256) hipLaunchKernelGGL((chol64_reg_kernel<8>), dim3(1), dim3(1024), 0, c->stream, (const double *)w.H, size, np, w.W, (int64_t)kk, np, 0.0, 0.0, w.flag);
        else hipLaunchKernelGGL(chol64_kernel, dim3(1), dim3(1024), 0, c->stream, (const double *)w.H, size, np, w.W, (int64_t)kk, np, 0.0, 0.0, w.flag);
        HIPCHK(hipGetLastError());
        int hf = 0;
        HIPCHK(hipMemcpyAsync(&hf, w.flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!hf) break;
        if (attempt == 2) return fail(CMF_EUNSUPPORTED, "range finder: Gram matrix is not positive definite (rank-deficient block)");
        // trace of the Gram scales the shift
        std::vector<double> hg(kk);
        HIPCHK(hipMemcpy(hg.data(), w.G, kk * sizeof(double), hipMemcpyDeviceToHost));
        double tr = 0.0;
        for (int i = 0; i < size; ++i) tr += hg[(size_t)i * np + i];
        shift = (attempt == 0 ? 1e-12 : 1e-8) * std::max(tr, 1e-300);
    }
    if (np <= 128) hipLaunchKernelGGL((tri_inverse64_reg_kernel<8>), dim3((unsigned)(np / 16)), dim3(256), 0, c->stream, (const double *)w.W, size, np, w.Xt, np, np);
    else if (np <= 256) hipLaunchKernelGGL((tri_inverse64_reg_kernel<16>), dim3((unsigned)(np / 16)), dim3(256), 0, c->stream, (const double *)w.W, size, np, w.Xt, np, np);
    else {
        const int rp = (int)std::max<int64_t>(1, std::min<int64_t>(32, ((int64_t)150 * 1024 - 128 * (int64_t)size) / (8 * ((int64_t)size + 2))));
        const size_t lds = ((size_t)16 * size + (size_t)rp * (size + 2)) * sizeof(double);
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&tri_inverse64_kernel), 152 * 1024));
        hipLaunchKernelGGL(tri_inverse64_kernel, dim3((unsigned)(np / 16)), dim3(256), lds, c->stream, (const double *)w.W, size, np, w.Xt, np, np, rp);
    }
    // Xt[c][i] = (L^-1)[i][c] = (R^-1)[c][i]: the right factor as it stands
    hipLaunchKernelGGL(axpby64_to_f32_kernel, dim3(nb), dim3(256), 0, c->stream, w.S32, (const double *)w.Xt, 1.0, (const double *)nullptr, 0.0, (int64_t)kk);
    HIPCHK(hipGetLastError());
    CHK(gemm_tall(c, Y, w.S32, T, rows_pad, np));
    std::swap(Y, T);
    return CMF_OK;
}

extern "C" int cmf_rsvd(cmf_ctx *c, int which, int transpose, int k, int size, int n_iter, const double *omega, double *U_out,
                        double *S_out, double *Vt_out) {
    NEED_PROBLEM(c);
    if ((which != 0 && which != 1) || !omega || !U_out || !S_out || !Vt_out || k <= 0 || size < k || n_iter < 0)
        return fail(CMF_EINVAL, "bad argument");
    if (!have_data(c, which)) return fail(CMF_EINVAL, "%s has not been set", which == 0 ? "X" : "Y");
    if (size > 1024) return fail(CMF_EUNSUPPORTED, "randomized SVD on the device supports n_components + oversampling <= 1024");
    DeviceGuard dg(c->device);
    const int64_t mr = which == 0 ? c->m : c->d, mc = which == 0 ? c->d : c->p;          // M is mr x mc
    const int64_t mrp = which == 0 ? c->mp : c->dp, mcp = which == 0 ? c->dp : c->pp;
    // A = M^T when transpose (sklearn's 'auto': n_samples < n_features), else M; A is R x C
    const int64_t R = transpose ? mc : mr, C = transpose ? mr : mc, Rp = transpose ? mcp : mrp, Cp = transpose ? mrp : mcp;
    const int np = pad_k(size);
    const float *Md = which == 0 ? c->X : c->Y;
    auto times = [&](bool with_At, const float *B, float *out) -> int { // out = A B (R rows) or A^T B (C rows), width np
        const bool use_mt = with_At != (transpose != 0);                // product with M^T ?
        if (!Md) return spmm(c, c->sp[which][use_mt ? 1 : 0], B, out, use_mt ? mcp : mrp, false, np);
        if (!use_mt) return gemm(c, MODE_NN, Md, mcp, B, np, out, mrp, np, mcp);
        return gemm(c, MODE_TN, Md, mcp, B, np, out, mcp, np, mrp);
    };
    const int64_t tall = std::max(Rp, Cp);
    RsvdWork w;
    const size_t tb = (size_t)tall * np * sizeof(float), kk = (size_t)np * np;
    void *blk = nullptr;
    const size_t bytes = 3 * tb + kk * sizeof(float) + 4 * kk * sizeof(double) + 64;
    HIPCHK(hipMalloc(&blk, bytes));
    struct Free { void *p; ~Free() { (void)hipFree(p); } } guard{blk};
    char *p = (char *)blk;
    w.Q = (float *)p; p += tb; w.Y = (float *)p; p += tb; w.T = (float *)p; p += tb;
    w.S32 = (float *)p; p += kk * sizeof(float);
    w.G = (double *)p; p += kk * sizeof(double); w.H = (double *)p; p += kk * sizeof(double);
    w.W = (double *)p; p += kk * sizeof(double); w.Xt = (double *)p; p += kk * sizeof(double);
    w.flag = (int *)p;
    HIPCHK(hipMemsetAsync(blk, 0, bytes, c->stream));
    // Q <- omega (C x size)
    CHK(upload_strided<double>(c, w.Q, np, C, size, omega, size, 1));
    // {Q, Y, T} is always a permutation of the three tall buffers; cholqr_pass(A, scratch) leaves its result in A (the
    // two pointers trade places)
    float *Q = w.Q, *Y = w.Y, *T = w.T;
    for (int it = 0; it < n_iter; ++it) {
        CHK(times(false, Q, Y));                       // Y = A Q            (R x size)
        CHK(cholqr_pass(c, w, Y, T, Rp, size, np));
        CHK(cholqr_pass(c, w, Y, T, Rp, size, np));
        CHK(times(true, Y, Q));                        // Q = A^T Y          (C x size)
        CHK(cholqr_pass(c, w, Q, T, Cp, size, np));
        CHK(cholqr_pass(c, w, Q, T, Cp, size, np));
    }
    CHK(times(false, Q, Y));                           // final range basis: Qf = orth(A Q)
    CHK(cholqr_pass(c, w, Y, T, Rp, size, np));
    CHK(cholqr_pass(c, w, Y, T, Rp, size, np));
    float *Bt = T;
    CHK(times(true, Y, Bt));                           // Bt = A^T Qf = B^T   (C x size), B = Qf^T A
    // B B^T = Bt^T Bt (size x size, float64) -> eigen-decomposition on the host: B = Uh S Vh^T
    CHK(gram64_w(c, Bt, Cp, np, w.G));
    std::vector<double> hg(kk), Gs((size_t)size * size), Vh, lam;
    HIPCHK(hipMemcpyAsync(hg.data(), w.G, kk * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < size; ++i)
        for (int j = 0; j < size; ++j) Gs[(size_t)i * size + j] = hg[(size_t)i * np + j];
    jacobi_eigh_host(Gs, size, Vh, lam);
    std::vector<double> s(size);
    for (int j = 0; j < size; ++j) s[j] = std::sqrt(std::max(lam[j], 0.0));
    // U = Qf Uh (R x size);  V = Bt Uh S^-1 (C x size)
    std::vector<float> uh(kk, 0.f), uhs(kk, 0.f);
    for (int i = 0; i < size; ++i)
        for (int j = 0; j < size; ++j) {
            uh[(size_t)i * np + j] = (float)Vh[(size_t)i * size + j];
            uhs[(size_t)i * np + j] = s[j] > 1e-300 ? (float)(Vh[(size_t)i * size + j] / s[j]) : 0.f;
        }
    float *Uf = Q;                                     // the last iterate is no longer needed
    HIPCHK(hipMemcpyAsync(w.S32, uh.data(), kk * sizeof(float), hipMemcpyHostToDevice, c->stream));
    CHK(gemm_tall(c, Y, w.S32, Uf, Rp, np));
    HIPCHK(hipStreamSynchronize(c->stream));
    // hand back the first k triplets in M's orientation (M = U S Vt): with A = M^T the roles of the two sides swap.
    // Vt is k x mc row-major: element (j, i) lives at j * mc + i, i.e. "row" i of the device matrix has stride 1, column j stride mc
    if (!transpose) CHK(download_strided<double>(c, Uf, np, mr, k, U_out, k, 1));     // U  = left vectors of A  (mr x k)
    else CHK(download_strided<double>(c, Uf, np, mc, k, Vt_out, 1, mc));               // Vt = left vectors of A, transposed
    HIPCHK(hipMemcpyAsync(w.S32, uhs.data(), kk * sizeof(float), hipMemcpyHostToDevice, c->stream));
    CHK(gemm_tall(c, Bt, w.S32, Y, Cp, np));           // right vectors of A (C x size), into the buffer of Qf
    HIPCHK(hipStreamSynchronize(c->stream));
    if (!transpose) CHK(download_strided<double>(c, Y, np, mc, k, Vt_out, 1, mc));
    else CHK(download_strided<double>(c, Y, np, mr, k, U_out, k, 1));
    for (int j = 0; j < k; ++j) S_out[j] = s[j];
    return CMF_OK;
}
