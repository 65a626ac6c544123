// cmf_chol_mfma.hip.h -- batched per-row Newton solve  step_i = g_i H_i^-1  for n_components <= 256 (k_pad = 256) through a
// BLOCKED right-looking Cholesky whose O(n^3) work runs on the matrix pipe (v_mfma_f32_32x32x2_f32).
//
// Replaces, at k_pad = 256, the rank-1 register Cholesky of chol_solve_kernel<16> (cmf_eigen.hip.h): one barrier and a few hundred
// cycles of latency per COLUMN (256 of them, 34 ms per C3 iteration for 57 344 matrices) become two barriers per 32-column PANEL.
// Reference: NewtonSolver._safe_invert + _row_newton_update, pycmf/cmf_solvers.py:346-356, :321-326, for the rows whose Hessian has
// lambda_min >= hessian_pertubation (decided exactly by a first factorisation of H - pert I unless a certificate says so), where
// max(|lambda|, pert) is the identity and the step is a plain solve; the others are flagged for the spectral clamp.
//
// Layout.  The matrix is cut into 8 x 8 blocks of 32 x 32; the 36 blocks (I, K), I >= K, of the lower triangle live in the
// accumulator registers of the four waves, 2D-cyclic: wave (a, b) = (I & 1, K & 1) holds block (I, K) in slot (I >> 1, K >> 1) --
// ten slots of 16 registers.  A block is kept TRANSPOSED in the MFMA accumulator layout: register r of lane l holds
//     D'[c][i] = A[32 I + i][32 K + c],   c = (r & 3) + 8 (r >> 2) + 4 (l >> 5),  i = l & 31.
// With that orientation every product of the factorisation takes its operands straight from accumulator registers:
//   * the 16 registers of a block ARE the 16 B-operands (k-slot = lane half) of  X D'  and the 16 A-operands of  D'^T-type products;
//   * a rank-1 column step of the diagonal block is ONE MFMA with the scaled row j in k-slot (j >> 2) & 1 and zero in the other.
// Per panel J:
//   1. chain (one wave): 32 column steps on the 32 x 32 diagonal block and on E (starts as I): one readlane (pivot), one rsq, two
//      MFMAs per step; E ends as L_JJ^-1.  No LDS, no barrier inside.  L_JJ^-1 goes to LDS (8 x 4.2 KB stay for the solves).
//   2. panel blocks (I, J), I > J:  L'_IJ = L_JJ^-1 D'_IJ  -- 16 MFMAs per block, A-operands from LDS, B-operands = the block's
//      own registers.  The finished blocks are published to LDS as raw register images (4 KB each).
//   3. trailing blocks (I, K), I >= K > J:  D'_IK -= L'_KJ^T-contraction L'_IJ  -- 16 MFMAs per block, both operands raw images.
// Two barriers per panel.  The triangular solves walk the same blocks: forward with per-lane dot products over the 16 registers,
// backward with a 16-step transpose-reduce over the lanes; the diagonal blocks through the stored L_JJ^-1.
#pragma once

namespace cmfk {

typedef float cm_f32x16 __attribute__((ext_vector_type(16)));

struct CholMfma {
    static constexpr int NB = 8, LP = 33;                       // blocks per side, pitch of an L_JJ^-1 image
    static constexpr int LINV = NB * 32 * LP;                   // floats
    static constexpr int LX = NB * 16 * 64;                     // raw register images of one panel
    static constexpr int VEC = 256;
    static constexpr size_t LDS_BYTES = (size_t)(LINV + LX + 3 * VEC + 4 * VEC + 32 + 32) * sizeof(float);
};

__device__ __forceinline__ int cm_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ constexpr int cm_slot(int p, int q) { return p * (p + 1) / 2 + q; }

__device__ __forceinline__ float cm_readlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// one column step of the chain: diagonal block D (symmetric, both triangles) and the companion E
template <int J_>
__device__ __forceinline__ void cm_chain_step(cm_f32x16 &D, cm_f32x16 &E, int h, int n31, float floor_, float &pmin, bool &ok) {
    constexpr int rj = (J_ & 3) + 4 * (J_ >> 3), hj = (J_ >> 2) & 1;
    const float piv = cm_readlane(D[rj], 32 * hj + J_);
    const bool good = piv > floor_;
    ok = ok && good;
    pmin = fminf(pmin, piv);
    const float inv = good ? __builtin_amdgcn_rsqf(piv) : 0.f;
    const bool mine = (h == hj);
    const float l = (mine && n31 > J_) ? D[rj] * inv : 0.f;     // L[n][j], n > j, in k-slot hj; zero in the other slot
    const float e = mine ? E[rj] * inv : E[rj];                  // row j of E is final after the scaling
    E[rj] = e;
    const float eb = mine ? e : 0.f;
    const float nl = -l;
    D = __builtin_amdgcn_mfma_f32_32x32x2f32(nl, l, D, 0, 0, 0);
    E = __builtin_amdgcn_mfma_f32_32x32x2f32(nl, eb, E, 0, 0, 0);
}

template <int J0, int J1>
struct CmChain {
    static __device__ __forceinline__ void run(cm_f32x16 &D, cm_f32x16 &E, int h, int n31, float floor_, float &pmin, bool &ok) {
        cm_chain_step<J0>(D, E, h, n31, floor_, pmin, ok);
        CmChain<J0 + 1, J1>::run(D, E, h, n31, floor_, pmin, ok);
    }
};
template <int J1>
struct CmChain<J1, J1> {
    static __device__ __forceinline__ void run(cm_f32x16 &, cm_f32x16 &, int, int, float, float &, bool &) {}
};

// sums of each of the 16 values over the 32 lanes of a half: lane (bits b4 b3 b2 b1 b0 of l & 31) ends with the total of
// register 8 b4 + 4 b3 + 2 b2 + b1 (both lanes of a b0 pair hold it)
__device__ __forceinline__ float cm_transpose_reduce(const float (&v)[16], int n31) {
    float u8[8], u4[4], u2[2];
    const bool b4 = n31 & 16, b3 = n31 & 8, b2 = n31 & 4, b1 = n31 & 2;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float send = b4 ? v[q] : v[q + 8], keep = b4 ? v[q + 8] : v[q];
        u8[q] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float send = b3 ? u8[q] : u8[q + 4], keep = b3 ? u8[q + 4] : u8[q];
        u4[q] = keep + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float send = b2 ? u4[q] : u4[q + 2], keep = b2 ? u4[q + 2] : u4[q];
        u2[q] = keep + __shfl_xor(send, 4, 64);
    }
    const float send = b1 ? u2[0] : u2[1], keep = b1 ? u2[1] : u2[0];
    float s = keep + __shfl_xor(send, 2, 64);
    s += __shfl_xor(s, 1, 64);
    return s;
}

// One panel step J of the factorisation (J compile-time: every slot index and most predicates fold away; a, b stay run-time,
// wave-uniform).  Returns with `failed` set (uniformly) when a pivot of the diagonal block was not above `floor_`.
template <int J>
__device__ __forceinline__ void cm_panel_step(cm_f32x16 (&acc)[10], float *linv, float *lx, int *iflag, int a, int b, int h, int n31, int lane,
                                              int nblk, float floor_, float &pmin, bool &failed) {
    using C = CholMfma;
    constexpr int Jp = J >> 1, Jb = J & 1;
    if (J >= nblk || failed) return;
    // ---- 1. chain on the diagonal block (its owner wave alone), in place; E starts as the identity and ends as L_JJ^-1
    if (a == Jb && b == Jb) {
        cm_f32x16 E;
#pragma unroll
        for (int r = 0; r < 16; ++r) E[r] = (cm_row(r, h) == n31) ? 1.f : 0.f;
        bool ok = true;
        CmChain<0, 32>::run(acc[cm_slot(Jp, Jp)], E, h, n31, floor_, pmin, ok);
        float *dst = linv + J * (32 * C::LP) + n31;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[cm_row(r, h) * C::LP] = E[r];
        if (!ok && lane == 0) iflag[0] = 1;
    }
    __syncthreads();
    if (iflag[0]) { failed = true; return; }
    // ---- 2. panel blocks (I, J), I > J:  L' = L_JJ^-1 D'
    if (b == Jb) {
        float af[16];
        const float *src = linv + J * (32 * C::LP) + n31 * C::LP + 4 * h;
#pragma unroll
        for (int s = 0; s < 16; ++s) af[s] = src[(s & 3) + 8 * (s >> 2)];
#pragma unroll
        for (int p = Jp; p < 4; ++p) {
            const int I = 2 * p + a;
            const bool on = (p > Jp || (Jb == 0 && a == 1)) && I < nblk;
            if (on) {
                cm_f32x16 o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) o = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], acc[cm_slot(p, Jp)][s], o, 0, 0, 0);
                acc[cm_slot(p, Jp)] = o;
                float *dst = lx + I * (16 * 64) + lane;
#pragma unroll
                for (int s = 0; s < 16; ++s) dst[s * 64] = o[s];
            }
        }
    }
    __syncthreads();
    // ---- 3. trailing blocks (I, K), I >= K > J
#pragma unroll
    for (int q = Jp; q < 4; ++q) {
        const int K = 2 * q + b;
        const bool kon = (q > Jp || (Jb == 0 && b == 1)) && K < nblk;
        if (kon) {
            float af[16];
            const float *sa = lx + K * (16 * 64) + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) af[s] = -sa[s * 64];
#pragma unroll
            for (int p = q; p < 4; ++p) {
                const int I = 2 * p + a;
                const bool on = (p > q || a >= b) && I < nblk;
                if (on) {
                    const float *sb = lx + I * (16 * 64) + lane;
#pragma unroll
                    for (int s = 0; s < 16; ++s) acc[cm_slot(p, q)] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], sb[s * 64], acc[cm_slot(p, q)], 0, 0, 0);
                }
            }
        }
    }
}

// Arguments as chol_solve_kernel (cmf_eigen.hip.h); sub-block images (sub > 1) and the timing diagnostics are not offered here.
__global__ __launch_bounds__(256, 1) void chol_solve_mfma_kernel(const float *Hin, const float *grad, float *step, int *need_jacobi, int n, int kp,
                                                                 int64_t stride, float pert, int nmat, const int *rowidx, const int *cert,
                                                                 int cert_rows, int cert_split, float *condest) {
    using C = CholMfma;
    extern __shared__ __attribute__((aligned(16))) float cm_smem[];
    float *linv = cm_smem;                 // [8][32 * 33]
    float *lx = linv + C::LINV;            // [8][16][64]
    float *vg = lx + C::LX;                // right-hand side g, then x
    float *vy = vg + C::VEC;               // y
    float *rt = vy + C::VEC;               // [32] block right-hand side (+ padding to VEC)
    float *part = rt + C::VEC;             // [4][256] per-wave partial sums of the substitutions
    float *red = part + 4 * C::VEC;        // [32] reductions / flags
    int *iflag = reinterpret_cast<int *>(red + 32);

    const int mat = blockIdx.x;
    if (mat >= nmat) return;
    const float *H = Hin + (int64_t)mat * stride;
    const int ldh = kp;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int a = w & 1, b = w >> 1, h = lane >> 5, n31 = lane & 31;
    const int64_t orow = rowidx ? rowidx[mat] : mat;
    const int nblk = (n + 31) >> 5;

    float dmax = 0.f;
    for (int i = t; i < n; i += 256) dmax = fmaxf(dmax, fabsf(H[(int64_t)i * ldh + i]));
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
    if (lane == 0) red[w] = dmax;
    for (int i = t; i < C::VEC; i += 256) vg[i] = (i < n) ? grad[orow * kp + i] : 0.f;
    __syncthreads();
    dmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const bool certified = cert && cert[(mat / cert_rows) * 2 + ((mat % cert_rows) >= cert_split ? 1 : 0)] == 0;

    cm_f32x16 acc[10];
    float pmin_all = 3.0e38f;

    for (int pass = certified ? 1 : 0; pass < 2; ++pass) {
        const float shift = pass == 0 ? pert : 0.f;
        const float floor_ = pass == 0 ? 4.0e-6f * dmax : 0.f;
        // ---- load: block (I, K) transposed; identity outside the valid n x n part
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q <= p; ++q) {
                const int I = 2 * p + a, K = 2 * q + b;
                const int gc = 32 * I + n31, gr0 = 32 * K + 4 * h;
                const float *src = H + (int64_t)gr0 * ldh + gc;
                const bool live = I >= K && gc < n;
                cm_f32x16 v;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2), gr = gr0 + dr;
                    float x = (gr == gc) ? 1.f : 0.f;
                    if (live && gr < n) x = src[(int64_t)dr * ldh] - (gr == gc ? shift : 0.f);
                    v[r] = x;
                }
                acc[cm_slot(p, q)] = v;
            }
        if (t == 0) iflag[0] = 0;
        __syncthreads();
        float pmin = 3.0e38f;
        bool failed = false;
        cm_panel_step<0>(acc, linv, lx, iflag, a, b, h, n31, lane, nblk, floor_, pmin, failed);
        cm_panel_step<1>(acc, linv, lx, iflag, a, b, h, n31, lane, nblk, floor_, pmin, failed);
        cm_panel_step<2>(acc, linv, lx, iflag, a, b, h, n31, lane, nblk, floor_, pmin, failed);
        cm_panel_step<3>(acc, linv, lx, iflag, a, b, h, n31, lane, nblk, floor_, pmin, failed);
        cm_panel_step<4>(acc, linv, lx, iflag, a, b, h, n31, lane, nblk, floor_, pmin, failed);
        cm_panel_step<5>(acc, linv, lx, iflag, a, b, h, n31, lane, nblk, floor_, pmin, failed);
        cm_panel_step<6>(acc, linv, lx, iflag, a, b, h, n31, lane, nblk, floor_, pmin, failed);
        cm_panel_step<7>(acc, linv, lx, iflag, a, b, h, n31, lane, nblk, floor_, pmin, failed);
        if (failed) { // lambda_min < pert (pass 0), or a non-positive pivot of H itself: the clamp matters -> spectral route
            if (t == 0) need_jacobi[orow] = 1;
            return;
        }
        pmin_all = pmin;
        if (pass == 0) __syncthreads(); // lx / linv are rewritten by the second factorisation
    }
    if (t == 0) need_jacobi[orow] = 0;
    if (condest) { // max H_ii / min L_ii^2 <= cond(H) (cmf_newton.hip.h, clamp_stats)
        for (int off = 32; off > 0; off >>= 1) pmin_all = fminf(pmin_all, __shfl_xor(pmin_all, off, 64));
        if (lane == 0) red[8 + w] = pmin_all;
    }
    // ---- forward substitution  L y = g
    for (int i = t; i < 4 * C::VEC; i += 256) part[i] = 0.f;
    __syncthreads();
    if (condest && t == 0) condest[orow] = dmax / fmaxf(fminf(fminf(red[8], red[9]), fminf(red[10], red[11])), 1.0e-37f);
    for (int J = 0; J < nblk; ++J) {
        const int Jp = J >> 1, Jb = J & 1;
        if (w == 0) {
            const int i0 = 32 * J + n31;
            const float r = vg[i0] - (part[i0] + part[C::VEC + i0] + part[2 * C::VEC + i0] + part[3 * C::VEC + i0]);
            if (h == 0) rt[n31] = r;
            const float *Lr = linv + J * (32 * C::LP) + n31 * C::LP + 16 * h; // row c = n31, columns 16 h ..
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += Lr[i] * rt[16 * h + i];
            s += __shfl_xor(s, 32, 64);
            if (h == 0) vy[i0] = s;
        }
        __syncthreads();
        if (b == Jb) {
            float yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[r] = vy[32 * J + cm_row(r, h)];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q <= p; ++q) {
                    const int I = 2 * p + a;
                    if (q == Jp && I > J && I < nblk) {
                        const cm_f32x16 d = acc[cm_slot(p, q)];
                        float s = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) s += d[r] * yv[r];
                        s += __shfl_xor(s, 32, 64);
                        if (h == 0) part[w * C::VEC + 32 * I + n31] += s;
                    }
                }
        }
        __syncthreads();
    }
    // ---- back substitution  L^T x = y
    for (int i = t; i < 4 * C::VEC; i += 256) part[i] = 0.f;
    __syncthreads();
    for (int J = nblk - 1; J >= 0; --J) {
        const int Jp = J >> 1, Jb = J & 1;
        if (w == 0) {
            const int i0 = 32 * J + n31;
            const float r = vy[i0] - (part[i0] + part[C::VEC + i0] + part[2 * C::VEC + i0] + part[3 * C::VEC + i0]);
            if (h == 0) rt[n31] = r;
            const float *Lc = linv + J * (32 * C::LP) + (16 * h) * C::LP + n31; // column i = n31, rows 16 h ..
            float s = 0.f;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) s += Lc[cc * C::LP] * rt[16 * h + cc];
            s += __shfl_xor(s, 32, 64);
            if (h == 0) vg[i0] = s;
        }
        __syncthreads();
        if (a == Jb) { // blocks (J, K), K < J: the block row of J
            const float xl = vg[32 * J + n31];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q <= p; ++q) {
                    const int K = 2 * q + b;
                    if (p == Jp && K < J) {
                        const cm_f32x16 d = acc[cm_slot(p, q)];
                        float v[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = d[r] * xl;
                        const float s = cm_transpose_reduce(v, n31);
                        const int r = ((n31 >> 4) & 1) * 8 + ((n31 >> 3) & 1) * 4 + ((n31 >> 2) & 1) * 2 + ((n31 >> 1) & 1);
                        if (!(n31 & 1)) part[w * C::VEC + 32 * K + cm_row(r, h)] += s;
                    }
                }
        }
        __syncthreads();
    }
    for (int i = t; i < kp; i += 256) step[orow * kp + i] = (i < n) ? vg[i] : 0.f;
}

} // namespace cmfk
