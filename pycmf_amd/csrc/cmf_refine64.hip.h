// cmf_refine64.hip.h -- BATCHED float64 refinement of ill-conditioned per-row Newton steps (cmf_newton.hip.h: refine_rows64).
//
// Round 3 redid every listed row on its own: ~15 small launches and two host synchronisations when its Hessian passes the
// threshold test, ~200 when the spectral clamp acts, capped at 16 384 rows per sweep.  Here the rows of a chunk's list travel
// together -- grid dimension y (or z) is the position in the list -- through
//   ref64_terms / ref64_grad / ref64_sparse_grad / ref64_grad_finish   residuals, weights and gradients in float64
//   ref64_wgram_kernel      H_b = diag I + s_S S + sum over up to two sides of scale * sum_j w_j o_j o_j^T on v_mfma_f64_16x16x4_f64
//                           (gathered rows of the other factor through LDS, 64 x 64 upper tiles, mirrored on store)
//   chol64_reg_kernel       (cmf_shared64.hip.h, batched by a matrix stride) H_b - pert I and H_b: the threshold test and the factor
//   ref64_solve_kernel      step_b = g_b H_b^-1 by blocked substitution with the float64 factor, rounded once to float32
// and, for the rows the clamp acts on, the Newton-Schulz images of shared_inverse64 as batched float64 matrix products.
// Host synchronisations: one per batch (the flags of the threshold test), one more when some rows are clamped.
// Reference arithmetic: pycmf/cmf_solvers.py:394-508 with _safe_invert (:346-356) on float64 Hessians.
#pragma once

namespace cmfk {

// One side of a per-row sweep as the float64 kernels see it (cmf_newton.hip.h: RowSide)
struct Ref64Side {
    const float *O = nullptr;        // other factor, pitch kp
    const int32_t *lists = nullptr;  // [rows][per] sample lists (null: all `per` candidates in order)
    int per = 0;
    const float *T = nullptr;        // targets; element of factor row i and sample j: T[i * t_row + j * t_col]
    int64_t t_row = 0, t_col = 0;
    double scale = 1.0;
    int link = 0;
};

// r[b][q] = link(o_j . f_i) - t_ij, w[b][q] = 1 | sigma'(z): one wave per sample, grid (ceil(per / 4), rows of the batch)
__global__ __launch_bounds__(256) void ref64_terms_kernel(Ref64Side sd, const float *F, int kp, const int *bad, int64_t r0, double *r, double *w,
                                                          int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= sd.per) return;
    const int b = blockIdx.y;
    const int64_t i = r0 + bad[b];
    const int64_t j = sd.lists ? sd.lists[i * sd.per + q] : q;
    const float *o = sd.O + j * kp, *f = F + i * kp;
    double z = 0.0;
    for (int e = lane; e < kp; e += 64) z += (double)o[e] * (double)f[e];
    for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
    if (lane == 0) {
        const double t = (double)sd.T[i * sd.t_row + j * sd.t_col];
        if (sd.link) {
            const double sg = 1.0 / (1.0 + exp(-z));
            r[b * ld + q] = sg - t;
            w[b * ld + q] = sg * (1.0 - sg);
        } else {
            r[b * ld + q] = z - t;
            w[b * ld + q] = 1.0;
        }
    }
}

// g[b][col] (+)= scale * sum_q r[b][q] o_{j(q)}[col]: 32 columns per workgroup, eight partial sums per column added in a fixed order
__global__ __launch_bounds__(256) void ref64_grad_kernel(Ref64Side sd, int kp, const int *bad, int64_t r0, const double *r, int64_t ld, double *g,
                                                         int first) {
    __shared__ float so[64][33];
    __shared__ double sr[64];
    __shared__ double part[8][32];
    const int t = threadIdx.x, c0 = blockIdx.x * 32, col = t & 31, ph = t >> 5, b = blockIdx.y;
    const int64_t i = r0 + bad[b];
    const int32_t *list = sd.lists ? sd.lists + i * sd.per : nullptr;
    const int s = sd.per;
    double acc = 0.0;
    for (int j0 = 0; j0 < s; j0 += 64) {
        __syncthreads();
        for (int e = t; e < 64 * 32; e += 256) {
            const int jj = e >> 5, q = e & 31;
            so[jj][q] = (j0 + jj < s) ? sd.O[(int64_t)(list ? list[j0 + jj] : j0 + jj) * kp + c0 + q] : 0.f;
        }
        if (t < 64) sr[t] = (j0 + t < s) ? r[b * ld + j0 + t] : 0.0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += sr[ph + 8 * u] * (double)so[ph + 8 * u][col];
    }
    part[ph][col] = acc;
    __syncthreads();
    if (t < 32) {
        double v = 0.0;
        for (int u = 0; u < 8; ++u) v += part[u][t];
        v *= sd.scale;
        double *gp = g + (int64_t)b * kp + c0 + t;
        *gp = first ? v : *gp + v;
    }
}

// natively sparse side: g[b][col] -= scale * sum over the stored values t_ij of data row i with j in the row's (ascending) list of t_ij o_j[col]
__global__ __launch_bounds__(256) void ref64_sparse_grad_kernel(const int64_t *indptr, const int32_t *idx, const float *val, const float *O, int kp,
                                                                const int32_t *sorted, int64_t per, double scale, const int *bad, int64_t r0,
                                                                double *g) {
    const int col = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (col >= kp) return;
    const int64_t row = r0 + bad[b];
    const int32_t *L = sorted ? sorted + row * per : nullptr;
    const int64_t beg = indptr[row], end = indptr[row + 1];
    double acc = 0.0;
    for (int64_t q = beg; q < end; ++q) {
        const int32_t j = idx[q];
        if (L) {
            int64_t lo = 0, hi = per;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (L[mid] < j) lo = mid + 1;
                else hi = mid;
            }
            if (lo >= per || L[lo] != j) continue;
        }
        acc += (double)val[q] * (double)O[(int64_t)j * kp + col];
    }
    g[(int64_t)b * kp + col] -= scale * acc;
}

// g[b] += l1 sign(f_i) + l2 f_i on the valid columns, zero on the padding                                        (:399-400, :420, :498)
__global__ __launch_bounds__(256) void ref64_grad_finish_kernel(double *g, const float *F, const int *bad, int64_t r0, double l1, double l2, int n, int kp) {
    const int col = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (col >= kp) return;
    double *gp = g + (int64_t)b * kp + col;
    if (col >= n) { *gp = 0.0; return; }
    const double fv = (double)F[(r0 + bad[b]) * kp + col];
    *gp += l1 * (fv > 0.0 ? 1.0 : (fv < 0.0 ? -1.0 : 0.0)) + l2 * fv;
}

// H[b] (kp x kp float64, both triangles) = diag I (valid block) + sscale S + sum over the sides of scale * sum_q w[b][q] o_j o_j^T on
// the float64 matrix pipe.  One upper 64 x 64 tile per workgroup (four waves, wave tile 32 x 32 = 2 x 2 MFMA blocks of 16 x 16), the
// gathered rows o_j pass through LDS 32 samples at a time (both column panels of the tile), the weight rides on the A-operand.
// Samples are added in list order within a workgroup: deterministic.  grid (tiles, batch); padding rows / columns of H are zero.
struct Ref64GramSide {
    const float *O = nullptr;
    const int32_t *lists = nullptr;
    int per = 0;
    const double *w = nullptr;   // [batch][ld] weights (null: 1)
    double scale = 0.0;
};
__global__ __launch_bounds__(256) void ref64_wgram_kernel(Ref64GramSide s1, Ref64GramSide s2, int64_t ldw, int kp, int n, const int *bad, int64_t r0,
                                                          double diag, const double *S, double sscale, double *H) {
    constexpr int TS = 64, LD = TS + 16;
    __shared__ __attribute__((aligned(16))) float sA[32 * LD];
    __shared__ __attribute__((aligned(16))) float sB[32 * LD];
    __shared__ double sw[32];
    const int T = kp / TS;
    int tb = blockIdx.x, ti = 0;
    while (tb >= T - ti) { tb -= T - ti; ++ti; }
    const int tj = ti + tb;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wi = w >> 1, wj = w & 1;
    const int l15 = lane & 15, lk = lane >> 4;
    const int b = blockIdx.y;
    const int64_t i = r0 + bad[b];
    f64x4 tot[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) tot[x][y] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int side = 0; side < 2; ++side) {
        const Ref64GramSide &sd = side == 0 ? s1 : s2;
        if (!sd.O || sd.per <= 0) continue;
        const int32_t *list = sd.lists ? sd.lists + i * sd.per : nullptr;
        f64x4 acc[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = f64x4{0.0, 0.0, 0.0, 0.0};
        for (int j0 = 0; j0 < sd.per; j0 += 32) {
            __syncthreads();
            for (int idx = t; idx < 8 * TS; idx += 256) {
                const int row = idx / (TS / 4), c4 = idx % (TS / 4);
                f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
                if (j0 + row < sd.per) {
                    const float *src = sd.O + (int64_t)(list ? list[j0 + row] : j0 + row) * kp + 4 * c4;
                    va = *reinterpret_cast<const f32x4 *>(src + ti * TS);
                    vb = *reinterpret_cast<const f32x4 *>(src + tj * TS);
                }
                *reinterpret_cast<f32x4 *>(sA + row * LD + 4 * c4) = va;
                *reinterpret_cast<f32x4 *>(sB + row * LD + 4 * c4) = vb;
            }
            if (t < 32) sw[t] = (j0 + t < sd.per) ? (sd.w ? sd.w[b * ldw + j0 + t] : 1.0) : 0.0;
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const double wk = sw[4 * s + lk];
                double a[2], bb[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    a[q] = wk * (double)sA[(4 * s + lk) * LD + wi * 32 + 16 * q + l15];
                    bb[q] = (double)sB[(4 * s + lk) * LD + wj * 32 + 16 * q + l15];
                }
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], bb[y], acc[x][y], 0, 0, 0);
            }
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) tot[x][y] += sd.scale * acc[x][y];
    }
    double *Hb = H + (int64_t)b * kp * kp;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = ti * TS + wi * 32 + 16 * x + lk + 4 * reg, cidx = tj * TS + wj * 32 + 16 * y + l15;
                double v = 0.0;
                if (r < n && cidx < n) {
                    v = tot[x][y][reg];
                    if (r == cidx) v += diag;
                    if (S) v += sscale * S[(int64_t)r * kp + cidx];
                }
                Hb[(int64_t)r * kp + cidx] = v;
                if (ti != tj) Hb[(int64_t)cidx * kp + r] = v;
            }
}

// step[sel] = g[sel] H^-1 with H = L L^T (L: lower triangle, row-major, pitch kp, float64): blocked forward and backward substitution,
// one 256-thread workgroup per system.  Lidx[q] = which of the factor images (stride lstride) holds the factor of system q,
// rows[q] = its position in the batch (gradient row) and bad[rows[q]] its chunk-relative step row.
__global__ __launch_bounds__(256) void ref64_solve_kernel(const double *Lbase, int64_t lstride, const int *Lidx, const int *rows, int nsys, const double *g,
                                                          const int *bad, float *step, int n, int kp) {
    __shared__ double v[256 + 32];       // the right-hand side, overwritten by y, then x
    __shared__ double dblk[32][33];
    __shared__ double part[8][32];
    const int q = blockIdx.x;
    if (q >= nsys) return;
    const double *L = Lbase + (int64_t)Lidx[q] * lstride;
    const int brow = rows[q];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 256 + 32; i += 256) v[i] = (i < n) ? g[(int64_t)brow * kp + i] : 0.0;
    __syncthreads();
    const int nb = (n + 31) / 32;
    // ---- forward: L y = g.  Block row I: rows 32 I + r (r = t >> 3), eight threads per row stride over the solved columns
    for (int I = 0; I < nb; ++I) {
        const int r = t >> 3, e = t & 7, row = 32 * I + r;
        double s = 0.0;
        if (row < n)
            for (int c = e; c < 32 * I; c += 8) s += L[(int64_t)row * kp + c] * v[c];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        for (int idx = t; idx < 32 * 32; idx += 256) {
            const int rr = idx >> 5, cc = idx & 31;
            dblk[rr][cc] = (32 * I + rr < n && 32 * I + cc < n && cc <= rr) ? L[(int64_t)(32 * I + rr) * kp + 32 * I + cc] : (rr == cc ? 1.0 : 0.0);
        }
        __syncthreads();
        if (e == 0 && row < 256 + 32) v[row] -= s;
        __syncthreads();
        if (t < 64) { // the 32 x 32 triangle by one wave: lane c holds y_c
            const int c = lane & 31;
            double rc = v[32 * I + c];
            for (int jj = 0; jj < 32; ++jj) {
                const double yj = __shfl(rc / dblk[jj][jj], jj, 64);
                if (c == jj) rc = yj;
                else if (c > jj) rc -= dblk[c][jj] * yj;
            }
            if (lane < 32) v[32 * I + c] = rc;
        }
        __syncthreads();
    }
    // ---- backward: L^T x = y.  Block I: x_i needs sum over rows j >= 32 (I + 1) of L[j][i] x_j: lane = column i (coalesced rows of L)
    for (int I = nb - 1; I >= 0; --I) {
        const int ci = t & 31, gph = t >> 5, col = 32 * I + ci;
        double s = 0.0;
        for (int j = 32 * (I + 1) + gph; j < n; j += 8) s += L[(int64_t)j * kp + col] * v[j];
        part[gph][ci] = s;
        for (int idx = t; idx < 32 * 32; idx += 256) {
            const int rr = idx >> 5, cc = idx & 31;
            dblk[rr][cc] = (32 * I + rr < n && 32 * I + cc < n && cc <= rr) ? L[(int64_t)(32 * I + rr) * kp + 32 * I + cc] : (rr == cc ? 1.0 : 0.0);
        }
        __syncthreads();
        if (t < 32) {
            double tot = 0.0;
            for (int u = 0; u < 8; ++u) tot += part[u][t];
            v[32 * I + t] -= tot;
        }
        __syncthreads();
        if (t < 64) {
            const int c = lane & 31;
            double rc = v[32 * I + c];
            for (int jj = 31; jj >= 0; --jj) {
                const double xj = __shfl(rc / dblk[jj][jj], jj, 64);
                if (c == jj) rc = xj;
                else if (c < jj) rc -= dblk[jj][c] * xj;
            }
            if (lane < 32) v[32 * I + c] = rc;
        }
        __syncthreads();
    }
    float *out = step + (int64_t)bad[brow] * kp;
    for (int i = t; i < kp; i += 256) out[i] = (i < n) ? (float)v[i] : 0.f;
}

// dst[q] = src[idx[q]] for kp x kp float64 images (compaction of the clamped rows' Hessians)
__global__ __launch_bounds__(256) void ref64_gather_kernel(const double *src, const int *idx, double *dst, int64_t kk) {
    const double *s = src + (int64_t)idx[blockIdx.y] * kk;
    double *d = dst + (int64_t)blockIdx.y * kk;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < kk; e += (int64_t)gridDim.x * 256) d[e] = s[e];
}

} // namespace cmfk
