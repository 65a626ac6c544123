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

#ifdef CMF_DIAG_BUILD // cycle stamps of one workgroup's phases (tools/chol_mfma_test.hip); never in the product build
__device__ unsigned long long cm_prof[128];
#define CM_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) cm_prof[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CM_STAMP(i) do { } while (0)
#endif

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

// TWO column steps of the chain (columns j, j + 1, j even) as ONE rank-2 MFMA per block: rows j and j + 1 of the diagonal block D
// (symmetric, both triangles) sit in registers rj, rj + 1 of the same lane half hj; column j is scaled, applied to row j + 1 by
// one FMA, column j + 1 is scaled, and a v_permlane32_swap puts the two columns side by side -- column j in lanes 0-31 (k-slot 0),
// column j + 1 in lanes 32-63 (k-slot 1) -- which is the operand layout of v_mfma_f32_32x32x2_f32.  The companion E (starts as
// the identity, ends as L_JJ^-1) takes the same two row operations.  No comparison, no branch: the smallest pivot is tracked
// and judged once per panel (a pivot <= floor, or a NaN after it, only ever produces garbage that the caller discards; the
// padding beyond the valid order carries max |H_ii| on its diagonal, so it can neither be the smallest pivot nor fall under the floor).
template <int J_>
__device__ __forceinline__ void cm_chain_pair(cm_f32x16 &D, cm_f32x16 &E, int h, int n31, int &pmin_bits) {
    static_assert((J_ & 1) == 0, "pairs start at even columns");
    constexpr int rj = (J_ & 3) + 4 * (J_ >> 3), hj = (J_ >> 2) & 1;
    const float piv0 = cm_readlane(D[rj], 32 * hj + J_);
    const float inv0 = __builtin_amdgcn_rsqf(piv0);
    const float l0 = D[rj] * inv0;                               // L[n][j] in lane n of half hj (n = j: the diagonal entry)
    const float c = cm_readlane(l0, 32 * hj + J_ + 1);           // L[j + 1][j]
    const float a1 = D[rj + 1] - c * l0;                         // row j + 1 after column j
    const float piv1 = cm_readlane(a1, 32 * hj + J_ + 1);
    const float inv1 = __builtin_amdgcn_rsqf(piv1);
    const float l1 = a1 * inv1;
    // smallest pivot on the SCALAR unit: pivots are wave-uniform (readlane results), and signed-integer order is float order for
    // positive floats while every negative float is a negative integer -- so min_i32 yields the exact minimum when all pivots
    // are positive (the condition estimate) and some negative value as soon as one is not (the failure test): no vector work
    pmin_bits = min(pmin_bits, min(__builtin_bit_cast(int, piv0), __builtin_bit_cast(int, piv1)));
    const float f0 = E[rj] * inv0;                               // final rows j, j + 1 of E
    const float f1 = (E[rj + 1] - c * f0) * inv1;
    const bool mine = (h == hj);
    E[rj] = mine ? f0 : E[rj];
    E[rj + 1] = mine ? f1 : E[rj + 1];
    const bool below = n31 > J_ + 1;
    const unsigned x = __float_as_uint(below ? l0 : 0.f), y = __float_as_uint(below ? l1 : 0.f);
    const auto sl = __builtin_amdgcn_permlane32_swap(x, y, false, false);   // [0] = {x.lo, y.lo}, [1] = {x.hi, y.hi}
    const auto se = __builtin_amdgcn_permlane32_swap(__float_as_uint(f0), __float_as_uint(f1), false, false);
    const float ol = __uint_as_float(hj == 0 ? sl[0] : sl[1]);
    const float oe = __uint_as_float(hj == 0 ? se[0] : se[1]);
    const float nl = -ol;
    D = __builtin_amdgcn_mfma_f32_32x32x2f32(nl, ol, D, 0, 0, 0);
    E = __builtin_amdgcn_mfma_f32_32x32x2f32(nl, oe, E, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}

template <int J0, int J1>
struct CmChain {
    static __device__ __forceinline__ void run(cm_f32x16 &D, cm_f32x16 &E, int h, int n31, int &pmin_bits) {
        cm_chain_pair<J0>(D, E, h, n31, pmin_bits);
        CmChain<J0 + 2, J1>::run(D, E, h, n31, pmin_bits);
    }
};
template <int J1>
struct CmChain<J1, J1> {
    static __device__ __forceinline__ void run(cm_f32x16 &, cm_f32x16 &, int, int, int &) {}
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

// The chain of panel J: the owner wave of block (J, J) turns it into L_JJ (discarded) and publishes L_JJ^-1 to LDS.
template <int J>
__device__ __forceinline__ void cm_chain_panel(cm_f32x16 (&acc)[10], float *linv, int *iflag, int h, int n31, int lane, float floor_, float &pmin) {
    using C = CholMfma;
    // (lane coordinates through an opaque asm: the lane masks of a chain are the same in all eight panels and would otherwise be
    // computed once and pinned in SGPRs for the whole kernel)
    int ho = h, no = n31;
    asm volatile("" : "+v"(ho), "+v"(no));
    cm_f32x16 E;
#pragma unroll
    for (int r = 0; r < 16; ++r) E[r] = (cm_row(r, ho) == no) ? 1.f : 0.f;
    int pbits = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pmin));
    CmChain<0, 32>::run(acc[cm_slot(J >> 1, J >> 1)], E, ho, no, pbits);
    pmin = __builtin_bit_cast(float, pbits);
    float *dst = linv + J * (32 * C::LP) + no;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[cm_row(r, ho) * C::LP] = E[r];
    if (!(pmin > floor_) && lane == 0) iflag[0] = 1;
}

// acc(p, q) -= L'_KJ (x) L'_IJ for one trailing block; af = the negated raw image of L'_KJ
template <int P, int Q>
__device__ __forceinline__ void cm_trail_block(cm_f32x16 (&acc)[10], const float (&af)[16], const float *lx, int I, int lane) {
    const float *sb = lx + I * (16 * 64) + lane;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[cm_slot(P, Q)] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], sb[s * 64], acc[cm_slot(P, Q)], 0, 0, 0);
}

// The trailing blocks of one panel step as compile-time recursions over the slot grid (q = block column pair, p = block row pair):
// slot (P, Q) of wave (a, b) is block (2 P + a, 2 Q + b); it takes part when its column lies behind the panel (K > J), it exists
// (I >= K, inside the valid order) and it is not the block the look-ahead has already updated.
template <int J, int Q, int P>
struct CmTrailRows {
    static __device__ __forceinline__ void run(cm_f32x16 (&acc)[10], const float (&af)[16], const float *lx, int a, int b, int lane, int nblk, bool next_owner) {
        constexpr int Np = (J + 1) >> 1;
        const int I = 2 * P + a;
        const bool on = (P > Q || a >= b) && I < nblk && !(next_owner && P == Np && Q == Np);
        if (on) cm_trail_block<P, Q>(acc, af, lx, I, lane);
        __builtin_amdgcn_sched_barrier(0);
        CmTrailRows<J, Q, P + 1>::run(acc, af, lx, a, b, lane, nblk, next_owner);
    }
};
template <int J, int Q>
struct CmTrailRows<J, Q, 4> {
    static __device__ __forceinline__ void run(cm_f32x16 (&)[10], const float (&)[16], const float *, int, int, int, int, bool) {}
};
template <int J, int Q>
struct CmTrailCols {
    static __device__ __forceinline__ void run(cm_f32x16 (&acc)[10], const float *lx, int a, int b, int lane, int nblk, bool next_owner) {
        constexpr int Jp = J >> 1, Jb = J & 1;
        const int K = 2 * Q + b;
        const bool kon = (Q > Jp || (Jb == 0 && b == 1)) && K < nblk;
        if (kon) {
            float af[16];
            const float *sa = lx + K * (16 * 64) + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) af[s] = -sa[s * 64];
            CmTrailRows<J, Q, Q>::run(acc, af, lx, a, b, lane, nblk, next_owner);
        }
        __builtin_amdgcn_sched_barrier(0);
        CmTrailCols<J, Q + 1>::run(acc, lx, a, b, lane, nblk, next_owner);
    }
};
template <int J>
struct CmTrailCols<J, 4> {
    static __device__ __forceinline__ void run(cm_f32x16 (&)[10], const float *, int, int, int, int, bool) {}
};

// One panel step J of the factorisation (J compile-time: every slot index and most predicates fold away; a, b stay run-time,
// wave-uniform).  On entry the chain of panel J has been run by its owner (cm_chain_panel: before the first step, or as the
// LOOK-AHEAD of step J - 1).  Look-ahead: the owner of block (J + 1, J + 1) updates that block first, runs the chain of panel
// J + 1 and only then its other trailing blocks -- the chain, one wave's 16 dependent rank-2 steps, runs beside the trailing
// products of the other three waves instead of in front of a barrier they all wait at.
// Returns with `failed` set (uniformly) when a pivot of the diagonal block was not above `floor_`.
template <int J>
__device__ __forceinline__ void cm_panel_step(cm_f32x16 (&acc)[10], float *linv, float *lx, int *iflag, int a, int b, int h, int n31, int lane,
                                              int nblk, float floor_, float &pmin, bool &failed) {
    using C = CholMfma;
    constexpr int Jp = J >> 1, Jb = J & 1;
    if (J >= nblk || failed) return;
    // wave coordinates and order through an opaque asm per step: the ~80 block predicates below are loop invariants of the whole
    // kernel otherwise, each pinned in SGPRs from the first instruction on
    asm volatile("" : "+s"(a), "+s"(b), "+s"(nblk));
    __syncthreads();                       // L_JJ^-1 and the verdict on its pivots are published
    CM_STAMP(8 + 3 * J);
    if (iflag[0]) { failed = true; return; }
    // ---- panel blocks (I, J), I > J:  L' = L_JJ^-1 D'
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
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    CM_STAMP(9 + 3 * J);
    // ---- trailing blocks (I, K), I >= K > J; block (J + 1, J + 1) first, then (its owner) the chain of the next panel
    constexpr int N = J + 1, Np = N >> 1, Nb = N & 1;
    const bool next_owner = (N < 8) && a == Nb && b == Nb && N < nblk;
    if constexpr (N < 8) {
        if (next_owner) {
            float af[16];
            const float *sa = lx + N * (16 * 64) + lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) af[s] = -sa[s * 64];
            cm_trail_block<Np, Np>(acc, af, lx, N, lane);
            __builtin_amdgcn_sched_barrier(0);
            cm_chain_panel<N>(acc, linv, iflag, h, n31, lane, floor_, pmin);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    CmTrailCols<J, Jp>::run(acc, lx, a, b, lane, nblk, next_owner);
    CM_STAMP(10 + 3 * J);
}

// Arguments as chol_solve_kernel (cmf_eigen.hip.h); sub-block images (sub > 1) and the timing diagnostics are not offered here.
__global__ __launch_bounds__(256, 2) void chol_solve_mfma_kernel(const float *Hin, const float *grad, float *step, int *need_jacobi, int n, int kp,
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
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int a = w & 1, b = w >> 1, h = lane >> 5, n31 = lane & 31;
    const int64_t orow = rowidx ? rowidx[mat] : mat;
    const int nblk = (n + 31) >> 5;

    float dmax = 0.f;
    for (int i = t; i < n; i += 256) dmax = fmaxf(dmax, fabsf(H[(int64_t)i * ldh + i]));
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
    if (lane == 0) red[w] = dmax;
    const bool test_only = grad == nullptr;   // threshold test alone (group certificates): the flag is the whole result
    for (int i = t; i < C::VEC; i += 256) vg[i] = (!test_only && i < n) ? grad[orow * kp + i] : 0.f;
    __syncthreads();
    dmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const bool certified = cert && cert[(mat / cert_rows) * 2 + ((mat % cert_rows) >= cert_split ? 1 : 0)] == 0;

    cm_f32x16 acc[10];
    float pmin_all = 3.0e38f;
    CM_STAMP(0);

    // pert == 0 (the images of the spectral clamp, already >= pert by construction): ONE factorisation, with the positivity floor
    const bool one_pass = certified || pert == 0.f;
    for (int pass = one_pass ? 1 : 0; pass < 2; ++pass) {
        const float shift = pass == 0 ? pert : 0.f;
        const float floor_ = (pass == 0 || !certified) && (pass == 0 || pert == 0.f) ? 4.0e-6f * dmax : 0.f;
        // ---- load: block (I, K) transposed; max |H_ii| I outside the valid n x n part
        // (the pointer and the order pass through an opaque asm: otherwise the 160 load addresses and their masks are hoisted out
        // of the pass loop as loop invariants and held in registers across everything -- 800 spilled registers)
        const float *Hp = H;
        int nn = n;
        asm volatile("" : "+s"(Hp), "+s"(nn));
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q <= p; ++q) {
                const int I = 2 * p + a, K = 2 * q + b;
                const int gc = 32 * I + n31, gr0 = 32 * K + 4 * h;
                const float *src = Hp + (int64_t)gr0 * ldh + gc;
                cm_f32x16 v;
                if (I >= K && 32 * I + 32 <= nn) { // the whole block lies inside the valid part: sixteen plain loads
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = src[(int64_t)((r & 3) + 8 * (r >> 2)) * ldh];
                    if (p == q && a == b) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] -= (gr0 + (r & 3) + 8 * (r >> 2) == gc) ? shift : 0.f;
                    }
                } else {
                    const bool live = I >= K && gc < nn;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2), gr = gr0 + dr;
                        float x = (gr == gc) ? dmax : 0.f;
                        if (live && gr < nn) x = src[(int64_t)dr * ldh] - (gr == gc ? shift : 0.f);
                        v[r] = x;
                    }
                }
                acc[cm_slot(p, q)] = v;
                if ((cm_slot(p, q) & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two blocks' loads in flight, not all ten
            }
        if (t == 0) iflag[0] = 0;
        __syncthreads();
        CM_STAMP(1);
        float pmin = 3.0e38f;
        bool failed = false;
        if (w == 0) cm_chain_panel<0>(acc, linv, iflag, h, n31, lane, floor_, pmin);   // panel 0; the later chains run as look-ahead
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
        if (test_only) {
            if (t == 0) need_jacobi[orow] = 0;
            return;
        }
        if (pass == 0) __syncthreads(); // lx / linv are rewritten by the second factorisation
    }
    CM_STAMP(2);
    if (t == 0) need_jacobi[orow] = 0;
    if (condest) { // max H_ii / min L_ii^2 <= cond(H) (cmf_newton.hip.h, clamp_stats)
        for (int off = 32; off > 0; off >>= 1) pmin_all = fminf(pmin_all, __shfl_xor(pmin_all, off, 64));
        if (lane == 0) red[8 + w] = pmin_all;
    }
    // ---- forward substitution  L y = g.  ONE barrier per block row: every wave forms the block's right-hand side and solves the
    // 32 x 32 diagonal system itself (redundantly, through L_JJ^-1 in LDS), then adds its own blocks' products to its own
    // partial sums -- nobody waits for a designated solver wave
    for (int i = t; i < 4 * C::VEC; i += 256) part[i] = 0.f;
    float *rw = rt + 64 * w;               // wave-private: [0, 32) block right-hand side, [32, 64) block solution
    for (int J = 0; J < nblk; ++J) {
        const int Jp = J >> 1, Jb = J & 1;
        __syncthreads();
        if (J == 0 && condest && t == 0) condest[orow] = dmax / fmaxf(fminf(fminf(red[8], red[9]), fminf(red[10], red[11])), 1.0e-37f);
        const int i0 = 32 * J + n31;
        const float r = vg[i0] - (part[i0] + part[C::VEC + i0] + part[2 * C::VEC + i0] + part[3 * C::VEC + i0]);
        if (h == 0) rw[n31] = r;
        const float *Lr = linv + J * (32 * C::LP) + n31 * C::LP + 16 * h; // row c = n31, columns 16 h ..
        float s4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent chains: a 16-term dependent FMA chain is 130 cycles of latency
#pragma unroll
        for (int i = 0; i < 16; ++i) s4[i & 3] += Lr[i] * rw[16 * h + i];
        float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        s += __shfl_xor(s, 32, 64);
        if (h == 0) rw[32 + n31] = s;
        if (w == 0 && h == 0) vy[i0] = s;
        if (b == Jb) {
            float yv[16];
#pragma unroll
            for (int r4 = 0; r4 < 16; ++r4) yv[r4] = rw[32 + cm_row(r4, h)];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q <= p; ++q) {
                    const int I = 2 * p + a;
                    if (q == Jp && I > J && I < nblk) {
                        const cm_f32x16 d = acc[cm_slot(p, q)];
                        float q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int r4 = 0; r4 < 16; ++r4) q4[r4 & 3] += d[r4] * yv[r4];
                        float sum = (q4[0] + q4[1]) + (q4[2] + q4[3]);
                        sum += __shfl_xor(sum, 32, 64);
                        if (h == 0) part[w * C::VEC + 32 * I + n31] += sum;
                    }
                }
        }
    }
    CM_STAMP(3);
    // ---- back substitution  L^T x = y, the same way (the block solution stays in registers: lane i holds x_i)
    __syncthreads();
    for (int i = t; i < 4 * C::VEC; i += 256) part[i] = 0.f;
    for (int J = nblk - 1; J >= 0; --J) {
        const int Jp = J >> 1, Jb = J & 1;
        __syncthreads();
        const int i0 = 32 * J + n31;
        const float r = vy[i0] - (part[i0] + part[C::VEC + i0] + part[2 * C::VEC + i0] + part[3 * C::VEC + i0]);
        if (h == 0) rw[n31] = r;
        const float *Lc = linv + J * (32 * C::LP) + (16 * h) * C::LP + n31; // column i = n31, rows 16 h ..
        float x4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) x4[cc & 3] += Lc[cc * C::LP] * rw[16 * h + cc];
        float xs = (x4[0] + x4[1]) + (x4[2] + x4[3]);
        xs += __shfl_xor(xs, 32, 64);
        if (w == 0 && h == 0) vg[i0] = xs;
        if (a == Jb) { // blocks (J, K), K < J: the block row of J, all in ONE basic block (a block that does not exist -- K >= J --
                       // runs on zeros) so that the five exchange stages of the up to four transpose-reduces overlap
            const int r4 = ((n31 >> 4) & 1) * 8 + ((n31 >> 3) & 1) * 4 + ((n31 >> 2) & 1) * 2 + ((n31 >> 1) & 1);
            float *pw = part + w * C::VEC + cm_row(r4, h);
#define CM_BACK_ROW(P)                                                                                        \
    do {                                                                                                      \
        float sums[P + 1];                                                                                    \
        _Pragma("unroll") for (int q = 0; q <= P; ++q) {                                                      \
            const float xm = (2 * q + b < J) ? xs : 0.f;                                                      \
            float v[16];                                                                                      \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) v[k] = acc[cm_slot(P, q)][k] * xm;                 \
            sums[q] = cm_transpose_reduce(v, n31);                                                            \
        }                                                                                                     \
        if (!(n31 & 1)) {                                                                                     \
            _Pragma("unroll") for (int q = 0; q <= P; ++q) pw[32 * (2 * q + b)] += sums[q];                   \
        }                                                                                                     \
    } while (0)
            switch (Jp) {
            case 0: CM_BACK_ROW(0); break;
            case 1: CM_BACK_ROW(1); break;
            case 2: CM_BACK_ROW(2); break;
            default: CM_BACK_ROW(3); break;
            }
#undef CM_BACK_ROW
        }
    }
    __syncthreads();
    CM_STAMP(4);
    for (int i = t; i < kp; i += 256) step[orow * kp + i] = (i < n) ? vg[i] : 0.f;
}

} // namespace cmfk
