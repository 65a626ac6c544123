// cmf_eigclamp.hip.h -- _safe_invert's clamp as a batched symmetric eigen-solve on the vector units (round 6).
//
// Reference: pycmf/cmf_solvers.py:346-356 (_safe_invert: eigh, |lambda|, clamp at pert, Q diag(1/lambda) Q^T) as used by the
// per-row sweeps, :321-326 (_row_newton_update: M[i] -= g H^-1).  A per-row sweep never needs the inverse, only the step
//   y = g * safe_inverse(H) = Q f(Lambda) Q^T g,   f(x) = 1 / max(|x|, pert),
// ONE vector per matrix.  Round 2-5 formed M = max(H, pert I) by Newton-Schulz polynomials of H - pert I: 37-49 products of
// 256^3 per row on the matrix pipe (1.3-1.6 GFLOP per row; 0.66 s of a C3 iteration at the reference's default l2 = 0, where the
// clamp acts on every row of U and Z from iteration 7 on -- 255 of 256 eigenvalues below pert there, ~140 of 256 at C3X:
// tools/r06_spectrum_probe.py).  Here, per matrix (n <= NP = k_pad in {128, 256}):
//   1. eig_tridiag_kernel   Householder tridiagonalisation H = Q T Q^T (4/3 n^3 flop = 22 MFLOP at n = 256, sixty times fewer than
//                           the polynomials), the matrix held in the REGISTERS of one workgroup (2 NP threads: thread (t, h) owns
//                           A[t][2 cl + h], cl < NP/2), symv and rank-2 update thread-local on packed FMAs, v / w broadcast from
//                           LDS; the reflectors go to a scratch image row by row (coalesced; H itself is left alone for the caller's
//                           fallback), g is carried along: gt = Q^T g.
//   2. eig_ql_kernel        implicit-shift QL on T, one LANE per matrix, the 64 matrices of a wave in lockstep (streamed d, e: no
//                           LDS), every plane rotation logged ((c, s): ~n^2 of them); the log is replayed forward on gt (= Z^T gt),
//                           the result scaled by f(lambda) and the log replayed backward (= Z ...): O(n^2) per matrix, never an
//                           n x n eigenvector matrix.
//   3. eig_backtransform_kernel  y = Q (.), reflectors applied in reverse, one wave per matrix; writes the step row and clears
//                           the matrix's flag.
// No assumption on the spectrum (|lambda| as the reference: an indefinite H is served too), no resolution parameter: eigenvalues
// are those of the float32 matrix to a few eps32 ||H||, the clamp is exact on them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfk {

typedef float eig_v2f __attribute__((ext_vector_type(2)));

// workspace of a batch of nb matrices (NB = nb rounded up to 64), floats:  d[NP][NB] | e[NP][NB] | gt[NP][NB] | tau[NB][NP] | sens[NB]
// (d, e, gt transposed: the QL kernel's lanes are matrices)
__host__ __device__ inline size_t eig_ws_floats(int np, int64_t nb_pad) { return (size_t)4 * np * (size_t)nb_pad + (size_t)nb_pad; }

// ---- 1. Householder tridiagonalisation -------------------------------------------------------------------------------------------
// 512 threads = 32 row classes x 16 column classes, 2D-cyclic: thread (r, q) owns A[r + 32 i][q + 16 k], i < NP/32, k < NP/16 (128
// registers at NP = 256), lane = q + 16 (r % 4), wave = r / 4 -- a wave's 16-lane DPP rows are the 16 column classes of one row
// class, so the symv's sum over columns is a DPP row reduction, and a thread needs only ITS rows' and columns' entries of v and w
// (40 floats per step from LDS; the first version, one matrix row per thread with v and w broadcast to every thread, moved 1.5 KB
// per thread and step through LDS and was bound by exactly that: 0.76 ms per matrix against 0.2 for the arithmetic).
// LDS vectors live in two permuted layouts: "row layout" [r * RI + i] = element r + 32 i, "column layout" [q * CK + k] = element
// q + 16 k, so that a thread's entries are contiguous (ds_read_b128).
template <int NP>
struct EigTri {
    static constexpr int RI = NP / 32, CK = NP / 16, NW = NP / 64;
    static __device__ __forceinline__ int row_slot(int t) { return (t & 31) * RI + (t >> 5); }
    static __device__ __forceinline__ int col_slot(int t) { return (t & 15) * CK + (t >> 4); }
};

template <int NP, int K>
__device__ __forceinline__ void eig_take_column(const eig_v2f (&a)[NP / 32][NP / 32], float (&val)[NP / 32]) {
#pragma unroll
    for (int i = 0; i < NP / 32; ++i) val[i] = (K & 1) ? a[i][K >> 1].y : a[i][K >> 1].x;
}

template <int NP>
__global__ __launch_bounds__(512) void eig_tridiag_kernel(const float *H, const int *idx, const float *grad, int n, int64_t stride,
                                                          float *ws, int64_t NB, float *R) {
    using L = EigTri<NP>;
    constexpr int RI = L::RI, CK = L::CK, NW = L::NW;
    // x: the current column below the diagonal (element t > j; zero elsewhere), dsc: its diagonal element
    __shared__ __attribute__((aligned(16))) float xr[NP], xq[NP], wq[NP], pr[NP], gr[NP];
    __shared__ float dsc;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 15, r = wave * 4 + (lane >> 4);
    const int m_idx = idx ? idx[b] : b;
    const float *mat = H + (int64_t)m_idx * stride;
    float *refl = R + (int64_t)b * NP * NP; // reflector j in row j
    float *dT = ws, *eT = ws + (size_t)NP * NB, *gT = ws + (size_t)2 * NP * NB, *tau = ws + (size_t)3 * NP * NB + (size_t)b * NP;
    eig_v2f a[RI][CK / 2]; // a[i][kk] = A[r + 32 i][q + 32 kk], A[r + 32 i][q + 32 kk + 16]
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int kk = 0; kk < CK / 2; ++kk) {
            const int t = r + 32 * i, c0 = q + 32 * kk, c1 = c0 + 16;
            const float v0 = mat[(int64_t)t * NP + c0], v1 = mat[(int64_t)t * NP + c1]; // (the image is NP x NP)
            a[i][kk].x = (t < n && c0 < n) ? v0 : 0.f;
            a[i][kk].y = (t < n && c1 < n) ? v1 : 0.f;
        }
    if (tid < NP) {
        const float x0 = (tid < n) ? mat[tid] : 0.f; // column 0
        xr[L::row_slot(tid)] = tid > 0 ? x0 : 0.f;
        xq[L::col_slot(tid)] = tid > 0 ? x0 : 0.f;
        if (tid == 0) dsc = x0;
        gr[L::row_slot(tid)] = (tid < n) ? grad[(int64_t)m_idx * NP + tid] : 0.f;
    }
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int j = 0; j + 1 < n; ++j) {
        const int rs1 = L::row_slot(j + 1);
        // ---- reflector of column j: every wave forms the same scalars from the whole vector
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int f = lane + 64 * u;
            const float x = xr[f];
            part += (f == rs1) ? 0.f : x * x;
        }
        const float sigma = wave_sum(part), alpha = xr[rs1], djj = dsc;
        float beta = alpha, tauj = 0.f, scale = 0.f;
        if (sigma > 0.f) {
            // (correctly rounded square root and divisions: H_j = I - tau v v^T is orthogonal only as far as tau ||v||^2 = 2 holds,
            // and 255 reflectors act on the right-hand side twice; the 1-ulp hardware reciprocals left 7e-5 on the step)
            beta = -copysignf(sqrtf(alpha * alpha + sigma), alpha);
            tauj = (beta - alpha) / beta;
            scale = 1.0f / (alpha - beta);
        }
        // this thread's entries of v (rows r + 32 i, columns q + 16 k): x * scale, and 1 at element j + 1 (x is zero at and above j)
        float vr[RI], vq[CK];
#pragma unroll
        for (int i = 0; i < RI; i += 4) {
            const float4 x4 = *reinterpret_cast<const float4 *>(&xr[r * RI + i]);
            vr[i] = x4.x * scale; vr[i + 1] = x4.y * scale; vr[i + 2] = x4.z * scale; vr[i + 3] = x4.w * scale;
        }
#pragma unroll
        for (int k = 0; k < CK; k += 4) {
            const float4 x4 = *reinterpret_cast<const float4 *>(&xq[q * CK + k]);
            vq[k] = x4.x * scale; vq[k + 1] = x4.y * scale; vq[k + 2] = x4.z * scale; vq[k + 3] = x4.w * scale;
        }
        {
            const bool myrow = ((j + 1) & 31) == r, mycol = ((j + 1) & 15) == q;
            switch ((j + 1) >> 5) {
#define EIG_CASE(I_) case I_: if constexpr (I_ < RI) vr[I_ < RI ? I_ : 0] = myrow ? 1.0f : vr[I_ < RI ? I_ : 0]; break;
                EIG_CASE(0) EIG_CASE(1) EIG_CASE(2) EIG_CASE(3) EIG_CASE(4) EIG_CASE(5) EIG_CASE(6) EIG_CASE(7)
#undef EIG_CASE
            default: break;
            }
            switch ((j + 1) >> 4) {
#define EIG_CASE(K_) case K_: if constexpr (K_ < CK) vq[K_ < CK ? K_ : 0] = mycol ? 1.0f : vq[K_ < CK ? K_ : 0]; break;
                EIG_CASE(0) EIG_CASE(1) EIG_CASE(2) EIG_CASE(3) EIG_CASE(4) EIG_CASE(5) EIG_CASE(6) EIG_CASE(7)
                EIG_CASE(8) EIG_CASE(9) EIG_CASE(10) EIG_CASE(11) EIG_CASE(12) EIG_CASE(13) EIG_CASE(14) EIG_CASE(15)
#undef EIG_CASE
            default: break;
            }
        }
        if (q == 2) {
#pragma unroll
            for (int i = 0; i < RI; ++i) refl[(int64_t)j * NP + r + 32 * i] = vr[i];
        }
        if (tid == 0) {
            dT[(size_t)j * NB + b] = djj;
            eT[(size_t)j * NB + b] = beta;
            tau[j] = tauj;
        }
        // finished parts of the trailing block (v = w = 0 there: skipping is an optimisation, never a correctness condition):
        // row class i of this wave when 4 wave + 3 + 32 i <= j, column pair kk when 31 + 32 kk <= j.  (The empty asm statements keep
        // the compiler from turning the uniform branches into per-element selects.)
        const int i_fin = (j - 4 * wave - 3 >= 0) ? (j - 4 * wave - 3) / 32 + 1 : 0;
        const int kk_fin = (j >= 31) ? (j - 31) / 32 + 1 : 0;
        // ---- p = A v: partial over this thread's columns, summed over the 16 column classes by DPP
        float p[RI];
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            eig_v2f acc = {0.f, 0.f}, acc2 = {0.f, 0.f};
            p[i] = 0.f;
            if (i >= i_fin) {
                asm volatile("");
#pragma unroll
                for (int kk = 0; kk < CK / 2; kk += 2) {
                    acc = __builtin_elementwise_fma(a[i][kk], eig_v2f{vq[2 * kk], vq[2 * kk + 1]}, acc);
                    acc2 = __builtin_elementwise_fma(a[i][kk + 1], eig_v2f{vq[2 * kk + 2], vq[2 * kk + 3]}, acc2);
                }
                acc += acc2;
                p[i] = group_sum<16>(acc.x + acc.y);
                if (r + 32 * i <= j) p[i] = 0.f; // (a finished row keeps its old entries in the registers)
            }
        }
        if (q == 0) {
#pragma unroll
            for (int i = 0; i < RI; i += 4) *reinterpret_cast<float4 *>(&pr[r * RI + i]) = make_float4(p[i], p[i + 1], p[i + 2], p[i + 3]);
        }
        __syncthreads();
        // ---- gamma = p . v, zeta = g . v (every wave, whole vectors), w = tau p - (tau^2 gamma / 2) v
        float pg = 0.f, pz = 0.f;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int f = lane + 64 * u;
            const float v = (f == rs1) ? 1.0f : xr[f] * scale;
            pg += pr[f] * v;
            pz += gr[f] * v;
        }
        const float gamma = wave_sum(pg), zeta = wave_sum(pz);
        const float hg = 0.5f * tauj * tauj * gamma;
        float wr[RI];
#pragma unroll
        for (int i = 0; i < RI; ++i) wr[i] = tauj * p[i] - hg * vr[i];
        if (q == 0) {
#pragma unroll
            for (int i = 0; i < RI; ++i) wq[L::col_slot(r + 32 * i)] = wr[i];
        }
        __syncthreads();
        if (q == 1) {
            const float tz = tauj * zeta;
#pragma unroll
            for (int i = 0; i < RI; ++i) gr[r * RI + i] -= tz * vr[i];
        }
        // ---- A -= v w^T + w v^T on the trailing block
#pragma unroll
        for (int kk = 0; kk < CK / 2; kk += 2) {
            if (kk + 1 >= kk_fin) {
                asm volatile("");
                const float4 w4 = *reinterpret_cast<const float4 *>(&wq[q * CK + 2 * kk]);
#pragma unroll
                for (int i = 0; i < RI; ++i) {
                    const eig_v2f nv = {-vr[i], -vr[i]}, nw = {-wr[i], -wr[i]};
                    a[i][kk] = __builtin_elementwise_fma(nw, eig_v2f{vq[2 * kk], vq[2 * kk + 1]}, __builtin_elementwise_fma(nv, eig_v2f{w4.x, w4.y}, a[i][kk]));
                    a[i][kk + 1] = __builtin_elementwise_fma(nw, eig_v2f{vq[2 * kk + 2], vq[2 * kk + 3]}, __builtin_elementwise_fma(nv, eig_v2f{w4.z, w4.w}, a[i][kk + 1]));
                }
            }
        }
        // ---- column j + 1 of the result is the next x: held by the threads of column class (j + 1) % 16 at k = (j + 1) / 16
        {
            const int kstar = (j + 1) >> 4, qstar = (j + 1) & 15;
            float val[RI];
#pragma unroll
            for (int i = 0; i < RI; ++i) val[i] = 0.f;
            switch (kstar) {
#define EIG_CASE(K_) case K_: if constexpr (K_ < CK) eig_take_column<NP, (K_ < CK ? K_ : 0)>(a, val); break;
                EIG_CASE(0) EIG_CASE(1) EIG_CASE(2) EIG_CASE(3) EIG_CASE(4) EIG_CASE(5) EIG_CASE(6) EIG_CASE(7)
                EIG_CASE(8) EIG_CASE(9) EIG_CASE(10) EIG_CASE(11) EIG_CASE(12) EIG_CASE(13) EIG_CASE(14) EIG_CASE(15)
#undef EIG_CASE
            default: break;
            }
            if (q == qstar) {
#pragma unroll
                for (int i = 0; i < RI; ++i) {
                    const int t = r + 32 * i;
                    if (t == j + 1) dsc = val[i];
                    const float xv = (t > j + 1) ? val[i] : 0.f;
                    xr[r * RI + i] = xv;
                    xq[L::col_slot(t)] = xv;
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        dT[(size_t)(n - 1) * NB + b] = dsc;
        eT[(size_t)(n - 1) * NB + b] = 0.f;
    }
    if (tid < n) gT[(size_t)tid * NB + b] = gr[L::row_slot(tid)];
}

// ---- 2. implicit-shift QL with a rotation log --------------------------------------------------------------------------------------
// (the classical tqli recurrences) on nb tridiagonal matrices, one LANE each, the 64 matrices of a wave in LOCKSTEP: every sweep of
// the wave runs i from max(m) - 1 down to min(l) over the lanes' own blocks [l, m]; a lane outside its block passes (rotation
// (1, 0)).  d and e of the 64 matrices sit in LDS as [i][lane] (128 KB at n = 256: one wave per CU -- a chunk of 8192 matrices is
// 128 waves, the LDS is not what limits the launch); with the index wave-uniform the log is [trip][lane] with no index stream (two
// trips per 16-byte store) and the replays need no per-lane bookkeeping.  What bounds the kernel is the dependent chain of one
// rotation (~ 100 cycles) times the ~ 1.2 n^2 trips of a wave: everything else is kept out of the way -- LDS reads CH elements
// ahead, log loads of the replays RC trips ahead, reciprocal / rsqrt instructions (1 ulp; a rotation needs c^2 + s^2 = 1 to
// round-off, no more).  History (profiles/HISTORY.md, round 6): every lane at its own index, d / e in LDS: 39 ms per 8192
// matrices (instruction issue of the divergent loop); lockstep with d / e streamed through global memory: 54 ms (four
// outstanding loads per lane), 26 ms with 16-element chunks (the 6-bit vmcnt: five memory operations per trip).
// Lanes drift apart by a few sweeps over a whole matrix (an eigenvalue takes one to three sweeps): the wave runs max-over-lanes
// sweeps, a few per cent more than the average.  Between sweeps a lane advances l over the eigenvalues that converged and picks the
// end of its next block WITHOUT scanning: the sweep itself notes the lowest off-diagonal element that became negligible (mlow),
// which together with the block's own end m (e[m] = 0) is all a scan would find; only a new block (l > m) scans.
// lg: (c, s) of two trips per float4 and lane, cap pairs per wave; sw: per wave and sweep (ihi, ilo, first pair); fail[b] = 1: no
// convergence / log overflow (the caller's fallback serves the matrix).  Output: gT[.][b] <- Z f(Lambda) Z^T gt.
struct EigSweep { int ihi, ilo; int64_t pos; };

template <int NP>
__global__ __launch_bounds__(64) void eig_ql_kernel(float *ws, int64_t NB, int nb, int n, float pert, float4 *lg, int64_t cap, EigSweep *sw_all,
                                                    int sw_cap, int *fail, float *lam_out, long long *stats, int early_exit) {
    constexpr int CH = 16, RC = 32; // elements read ahead of the sweep (LDS), trips of log read ahead of a replay (global)
    extern __shared__ float eig_lds[];
    const int lane = threadIdx.x, b = blockIdx.x * 64 + lane;
    const bool live = b < nb;
    float *D = eig_lds + lane, *E = eig_lds + NP * 64 + lane; // element i of this lane's matrix: D[i * 64]
    float *dT = ws + b, *eT = ws + (size_t)NP * NB + b, *gT = ws + (size_t)2 * NP * NB + b;
    float4 *wl = lg + (size_t)blockIdx.x * cap * 64 + lane;    // pair t: wl[t * 64]
    EigSweep *sw = sw_all + (size_t)blockIdx.x * sw_cap;
    float anorm = 0.f, eprev = 0.f;
    for (int i = 0; i < n; ++i) {
        const float di = dT[(size_t)i * NB], ei = eT[(size_t)i * NB];
        D[i * 64] = di;
        E[i * 64] = ei;
        anorm = fmaxf(anorm, fabsf(di) + fabsf(ei) + eprev);
        eprev = fabsf(ei);
    }
    // an off-diagonal element is negligible relative to its neighbours (the classical test) OR to the matrix: eps32 ||T|| / 2.  The
    // second test is what ends the iteration on rank-deficient Hessians (fewer samples than components) -- their null space is
    // round-off noise of the large part, which the relative test would chase for ever -- and it costs nothing here: the clamp
    // max(|lambda|, pert) needs eigenvalues to an ABSOLUTE accuracy, and a float32 matrix defines them to eps32 ||T|| anyway.
    const float tol_abs = 5.9604645e-8f * anorm;
    auto negl = [&](float e_, float d0, float d1) -> bool {
        const float em = fabsf(e_), dd = fabsf(d0) + fabsf(d1);
        return (em + dd) == dd || em <= tol_abs;
    };
    auto negl_mem = [&](int q_) -> bool {
        if (q_ >= n - 1) return true;
        return negl(E[q_ * 64], D[q_ * 64], D[(q_ + 1) * 64]);
    };
    // ---- how much of the spectrum does the solve need?  f(x) = 1 / max(|x|, pert) is CONSTANT on (-pert, pert) and 1 / x above: once
    // the eigenvalues still to be found (those of the trailing block [l, n): QL deflates at the top) all lie inside (-pert, pert),
    // f of that block is I / pert, and once they all lie above pert it is the block's inverse, a positive definite tridiagonal solve
    // -- either way the iteration can stop.  Sturm counts at +-pert say how many eigenvalues are inside / above / below; every
    // eigenvalue the iteration finds is taken off its count.  (A count and the iteration can disagree only about an eigenvalue within
    // round-off of the threshold, where the two treatments agree to round-off: f is continuous.)  While one side is a small
    // minority its eigenvalues are looked for FIRST: the first sweep for a new eigenvalue takes a bound of the spectrum (or zero) as
    // its shift instead of Wilkinson's.  C3 at the reference's default l2 = 0 (one eigenvalue of 2e3, 255 under the threshold):
    // three sweeps instead of 320.
    int rem_in = 0, rem_pos = 0, rem_neg = 0;
    auto sturm = [&](int l0, float sg, int &below_pos, int &below_neg) { // eigenvalues of T[l0:n, l0:n] below +sg / below -sg
        float qp = D[l0 * 64] - sg, qm = D[l0 * 64] + sg;
        int cp = qp < 0.f, cm = qm < 0.f;
        for (int i = l0 + 1; i < n; ++i) {
            const float di = D[i * 64], e2 = E[(i - 1) * 64] * E[(i - 1) * 64];
            if (fabsf(qp) < 1e-30f) qp = -1e-30f;
            if (fabsf(qm) < 1e-30f) qm = -1e-30f;
            qp = di - sg - e2 / qp;
            qm = di + sg - e2 / qm;
            cp += qp < 0.f;
            cm += qm < 0.f;
        }
        below_pos = cp; below_neg = cm;
    };
    {
        int cp, cm;
        sturm(0, pert, cp, cm);
        rem_neg = cm; rem_in = cp - cm; rem_pos = n - cp;
    }
    bool seed_ok_top = true;
    float top = anorm; // upper bound of the eigenvalues still to be found above pert: the last one found there
    auto take = [&](float lam) { // an eigenvalue found: off its count
        if (lam > pert && seed_ok_top) top = fminf(top, lam);
        if (lam > pert) --rem_pos;
        else if (lam < -pert) --rem_neg;
        else --rem_in;
    };
    int mode = 0, lstop = n;     // mode 1: the block [lstop, n) is left tridiagonal, all its eigenvalues inside (-pert, pert); 2: all above pert
    bool seed_ok = true;
    int l = 0, m = -1, mlow = -1, iter = 0, nsw = 0;
    int64_t pos = 0;
    long long lane_rot = 0, wave_trips = 0;
    const long long t_start = stats ? (long long)__builtin_readcyclecounter() : 0;
    bool done = !live, bad = false, conv = false;
    for (;;) {
        // ---- per lane: advance over converged eigenvalues, settle the block [l, m] of the next sweep
        if (!done) {
            if (conv) { take(D[l * 64]); ++l; iter = 0; conv = false; }
            for (;;) {
                if (l >= n) { done = true; break; }
                if (early_exit && rem_neg <= 0 && (rem_pos <= 0 || rem_in <= 0)) {
                    // by the bookkeeping the rest of the spectrum is one-sided.  COUNT AGAIN on the block itself before stopping: an
                    // eigenvalue within round-off of the threshold may have been booked on the other side than the first count had
                    // it -- harmless for that eigenvalue, but it leaves the wrong side one short, and the eigenvalue that then
                    // remains (any value at all) would get the wrong treatment
                    int cp, cm;
                    sturm(l, pert, cp, cm);
                    rem_neg = cm; rem_in = cp - cm; rem_pos = (n - l) - cp;
                    if (rem_neg == 0 && (rem_pos == 0 || rem_in == 0)) {
                        mode = rem_pos == 0 ? 1 : 2; lstop = l; done = true;
                        break;
                    }
                }
                if (l > m) { // a new block: first negligible off-diagonal element at or after l
                    int q_ = l;
                    while (!negl_mem(q_)) ++q_;
                    m = q_; mlow = -1;
                }
                if (l == m) { take(D[l * 64]); ++l; iter = 0; continue; } // e[l] negligible: d[l] is an eigenvalue
                if (mlow >= 0 && mlow < l) mlow = -1;
                if (iter == 0 && (mlow == l || negl_mem(l))) { take(D[l * 64]); ++l; continue; }
                if (mlow > l) { m = mlow; mlow = -1; }
                break;
            }
            if (!done && ++iter > 60) { bad = true; done = true; }
            if (iter > 10) seed_ok = false; // a bound as shift did not single out an eigenvalue (no gap): Wilkinson shifts from now on
        }
        const bool active = !done;
        int ihi = active ? m - 1 : -1, ilo = active ? l : n;
        for (int off = 32; off > 0; off >>= 1) {
            ihi = max(ihi, __shfl_xor(ihi, off, 64));
            ilo = min(ilo, __shfl_xor(ilo, off, 64));
        }
        ihi = __builtin_amdgcn_readfirstlane(ihi); // (wave-uniform: keep the loop bounds and the range tests on the scalar unit)
        ilo = __builtin_amdgcn_readfirstlane(ilo);
        if (ihi < 0) break; // every lane is done
        if (active) lane_rot += m - l;
        wave_trips += ihi - ilo + 1;
        const int npairs = (ihi - ilo + 2) / 2;
        if (pos + npairs > cap || nsw >= sw_cap) { // (not reached with cap = n^2 pairs: measured 0.3 .. 0.6 n^2)
            if (active) bad = true;
            break;
        }
        if (lane == 0) { sw[nsw].ihi = ihi; sw[nsw].ilo = ilo; sw[nsw].pos = pos; }
        ++nsw;
        // ---- shift (per lane)
        float g = 0.f, s = 1.f, c = 1.f, p = 0.f, dnext = 0.f;
        bool run = active, uflow = false; // run: this lane still takes part in the sweep
        if (active) {
            const float dl = D[l * 64], el = E[l * 64];
            g = (D[(l + 1) * 64] - dl) * __builtin_amdgcn_rcpf(2.0f * el);
            const float r = __builtin_amdgcn_sqrtf(g * g + 1.0f);
            const float gw = g, dm = D[m * 64];
            g = dm - dl + el * __builtin_amdgcn_rcpf(gw + copysignf(r, gw)); // Wilkinson: the eigenvalue of the top 2 x 2 block nearer d[l]
            const int outside = rem_pos + rem_neg, minority = min(rem_in, outside);
            if (early_exit && seed_ok && minority > 0 && minority * 8 <= n - l) { // the minority side first (see above)
                // first sweep: a bound of the spectrum (or zero) as shift; then the eigenvalue of the top 2 x 2 block on the WANTED
                // side (Wilkinson's choice, the one nearer d[l], wanders off to whatever the block happens to be near)
                const float target = outside <= rem_in ? (rem_pos > 0 ? top : -anorm) : 0.f; // (found in descending order: the last one bounds the rest)
                const float k1 = dl - el * __builtin_amdgcn_rcpf(gw + r), k2 = dl - el * __builtin_amdgcn_rcpf(gw - r);
                const float ks = fabsf(k1 - target) <= fabsf(k2 - target) ? k1 : k2;
                g = dm - (iter == 1 ? target : ks);
            }
        }
        mlow = -1;
        // ---- the sweep: i = ihi .. ilo for the whole wave.  One trip is straight-line code: every lane computes the rotation,
        // selects decide what it keeps (exec-mask branches around the pieces cost more than the pieces, and every join made the
        // compiler wait for ALL outstanding LDS operations).  A lane leaves the sweep at i == l (its last rotation: d[l], e[l] get
        // their final values) or on underflow (f = g = 0: the block is split there and re-scanned).
        float dcur = D[(ihi + 1) * 64], e1o = E[(ihi + 1) * 64]; // current d[i + 1], e[i + 1]
        float4 *wp = wl + (size_t)pos * 64;
        auto trip = [&](int i, float di, float ei) -> float2 {
            const bool act = run && i < m; // (i >= l holds while run is set)
            const float f = s * ei, bb = c * ei;
            const float x2 = f * f + g * g;
            const bool uf = act && x2 == 0.f, ok = act && x2 != 0.f, fin = ok && i == l;
            // 1 / sqrt(x2): the hardware estimate (1 ulp, not unbiased) and one Newton step -- every element meets some 600
            // rotations, whose c^2 + s^2 - 1 must stay at the level of the rounding of c and s themselves
            float ri = __builtin_amdgcn_rsqf(x2);
            ri = fmaf(0.5f * ri, fmaf(-(x2 * ri), ri, 1.0f), ri);
            const float rr = x2 * ri;
            const float s1 = f * ri, c1 = g * ri;
            const float g1 = dcur - p;
            const float r2 = (di - g1) * s1 + 2.0f * c1 * bb;
            const float p1 = s1 * r2;
            const float dn = g1 + p1, g2 = c1 * r2 - bb, dl = di - p1;
            E[(i + 1) * 64] = ok ? rr : (uf ? 0.f : e1o);
            D[(i + 1) * 64] = ok ? dn : (uf ? g1 : dcur);
            D[i * 64] = fin ? dl : di;
            E[i * 64] = fin ? g2 : ei;
            if (ok && i + 1 < m && negl(rr, dn, dnext)) mlow = i + 1; // e[i + 1] is final for this sweep
            if (fin) conv = negl(g2, dl, dn);
            dnext = ok ? dn : dnext;
            s = ok ? s1 : s; c = ok ? c1 : c; g = ok ? g2 : g; p = ok ? p1 : p;
            dcur = fin ? dl : di;
            e1o = fin ? g2 : ei;
            uflow = uflow || uf;
            run = run && !uf && !fin;
            return ok ? make_float2(c1, s1) : make_float2(1.f, 0.f);
        };
        float dA[CH], eA[CH], dB[CH], eB[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int iu = max(ihi - u, 0);
            dA[u] = D[iu * 64];
            eA[u] = E[iu * 64];
        }
        for (int itop = ihi; itop >= ilo; itop -= CH) {
            // (the elements of the next chunk lie below everything this chunk's trips write: i + 1 >= itop - CH + 2)
            if (itop - CH >= ilo) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int iu = max(itop - CH - u, 0);
                    dB[u] = D[iu * 64];
                    eB[u] = E[iu * 64];
                }
            }
#pragma unroll
            for (int u = 0; u < CH; u += 2) {
                if (itop - u >= ilo) {
                    const float2 c0 = trip(itop - u, dA[u], eA[u]);
                    float2 c1 = make_float2(1.f, 0.f);
                    if (itop - u - 1 >= ilo) c1 = trip(itop - u - 1, dA[u + 1], eA[u + 1]);
                    wp[(size_t)((ihi - itop + u) >> 1) * 64] = make_float4(c0.x, c0.y, c1.x, c1.y);
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) { dA[u] = dB[u]; eA[u] = eB[u]; }
        }
        pos += npairs;
        if (active) E[m * 64] = 0.f;                      // the end of the block this lane swept (written over by its first rotation)
        if (uflow) { m = l - 1; --iter; }                 // underflow: the next settle re-scans from l
    }
    const long long t_ql = stats ? (long long)__builtin_readcyclecounter() : 0;
    // ---- a block left tridiagonal with its spectrum above pert: factor it now (T = L diag(q) L^T, positive definite: no pivoting),
    // pivots over d, multipliers to global memory (e's place in LDS is about to hold r)
    const float delta = 4.0f * 1.1920929e-7f * anorm;
    float rest_low = 3.0e38f; // smallest |eigenvalue| the block [lstop, n) may hold that is not safely clamped
    if (mode == 2) {
        float q = D[lstop * 64];
        for (int i = lstop + 1; i < n; ++i) {
            const float ei = E[(i - 1) * 64], mi = ei / q;
            eT[(size_t)i * NB] = mi;
            q = D[i * 64] - mi * ei;
            D[i * 64] = q;
        }
        rest_low = pert;
    } else if (mode == 1) { // inside (-pert, pert): all of it safely (by delta) inside?
        int cp, cm;
        sturm(lstop, pert - delta, cp, cm);
        if (cp - cm != n - lstop) rest_low = pert - delta;
    }
    // ---- r = Z^T gt (forward replay), scale by f(lambda), y = Z r (backward replay); r takes e's place in LDS
    float *Rv = E;
    for (int q = 0; q < n; ++q) Rv[q * 64] = gT[(size_t)q * NB];
    for (int t = 0; t < nsw; ++t) {
        const int ihi = __builtin_amdgcn_readfirstlane(sw[t].ihi), ilo = __builtin_amdgcn_readfirstlane(sw[t].ilo), np_ = (ihi - ilo + 2) / 2;
        const float4 *wp = wl + (size_t)sw[t].pos * 64;
        float4 cA[RC / 2], cB[RC / 2];
#pragma unroll
        for (int u = 0; u < RC / 2; ++u) cA[u] = wp[(size_t)min(u, np_ - 1) * 64];
        float rcur = Rv[(ihi + 1) * 64];
        for (int itop = ihi; itop >= ilo; itop -= RC) {
            if (itop - RC >= ilo) {
#pragma unroll
                for (int u = 0; u < RC / 2; ++u) cB[u] = wp[(size_t)min(((ihi - itop + RC) >> 1) + u, np_ - 1) * 64];
            }
#pragma unroll
            for (int u = 0; u < RC; ++u)
                if (itop - u >= ilo) {
                    const float cc = (u & 1) ? cA[u >> 1].z : cA[u >> 1].x, ss = (u & 1) ? cA[u >> 1].w : cA[u >> 1].y;
                    const float ri = Rv[(itop - u) * 64];
                    Rv[(itop - u + 1) * 64] = ss * ri + cc * rcur;
                    rcur = cc * ri - ss * rcur;
                }
#pragma unroll
            for (int u = 0; u < RC / 2; ++u) cA[u] = cB[u];
        }
        Rv[ilo * 64] = rcur;
    }
    const long long t_fwd = stats ? (long long)__builtin_readcyclecounter() : 0;
    // Sensitivity of this solve to the float32 arithmetic behind T (the refinement's test, clamp_stats_kernel): the eigenvalues are
    // those of H to delta ~ 4 eps32 ||T||; f(x) = 1 / max(|x|, pert) is constant below the threshold and has slope 1 / x^2 above,
    // so the relative error of the step is bounded by delta / (smallest |lambda| that is NOT safely clamped) -- and by zero when
    // every eigenvalue is: then max(H, pert I) = pert I whatever the rounding.  (C3 at the reference's default l2 = 0, steady state:
    // one eigenvalue of 2e3 and 255 below pert: 1e-6, although ||H|| / pert = 1e4.)
    // ... PLUS the rotation of the invariant subspaces against each other: a direction j that is not clamped leaks into the clamped
    // subspace by an angle delta / (lambda_j - lambda_c), and f differs by up to 1 / pert - 1 / lambda_j between the two, so the
    // step picks up an error of (delta / (pert lambda_j)) |g_j| = (delta / pert) |step_j| there -- small against ||step|| while the
    // clamped part of the step (g / pert) dominates it (C3's steady state: 1e-4 of it), NOT small where the gradient lies along
    // the large eigenvalues (second campaign of profiles/fuzz_r06.md: 6e-3 on V with the first term alone).  su2 / st2: squared
    // norms of the step's part in the directions not safely clamped / of the whole step (in the eigenbasis: same norms).
    float minabove = rest_low, su2 = 0.f, st2 = 0.f;
    for (int q = 0; q < lstop; ++q) {
        const float lam = D[q * 64], al = fabsf(lam);
        const float y = Rv[q * 64] / fmaxf(al, pert);
        if (al >= pert - delta) { minabove = fminf(minabove, al); su2 = fmaf(y, y, su2); }
        st2 = fmaf(y, y, st2);
        Rv[q * 64] = y;
        if (lam_out && live) lam_out[(size_t)b * NP + q] = lam;
    }
    if (mode == 1) {
        const float ip = 1.0f / pert;
        for (int q = lstop; q < n; ++q) {
            const float y = Rv[q * 64] * ip;
            if (rest_low < 3.0e38f) su2 = fmaf(y, y, su2); // (not all of the rest is safely inside: priced as if none were)
            st2 = fmaf(y, y, st2);
            Rv[q * 64] = y;
        }
    } else if (mode == 2) { // L z = r, y = L^-T diag(1 / q) z
        float z = Rv[lstop * 64];
        for (int i = lstop + 1; i < n; ++i) {
            z = Rv[i * 64] - eT[(size_t)i * NB] * z;
            Rv[i * 64] = z;
        }
        float y = Rv[(n - 1) * 64] / D[(n - 1) * 64];
        Rv[(n - 1) * 64] = y;
        for (int i = n - 2; i >= lstop; --i) {
            y = Rv[i * 64] / D[i * 64] - eT[(size_t)(i + 1) * NB] * y;
            Rv[i * 64] = y;
        }
        for (int i = lstop; i < n; ++i) { // the whole block lies above the threshold
            const float yi = Rv[i * 64];
            su2 = fmaf(yi, yi, su2);
            st2 = fmaf(yi, yi, st2);
        }
    }
    if (lam_out && live)
        for (int q = lstop; q < n; ++q) lam_out[(size_t)b * NP + q] = __int_as_float(0x7fc00000); // (not computed: the iteration stopped)
    if (live) ws[(size_t)4 * NP * NB + b] = minabove < 3.0e38f ? delta / fmaxf(minabove, 1e-30f) + (delta / pert) * sqrtf(su2 / fmaxf(st2, 1e-37f)) : 0.f;
    for (int t = nsw - 1; t >= 0; --t) {
        // ascending i = ilo .. ihi: trip index ihi - i descends; chunks of RC trips aligned to the pairs of the forward order
        const int ihi = __builtin_amdgcn_readfirstlane(sw[t].ihi), ilo = __builtin_amdgcn_readfirstlane(sw[t].ilo), len = ihi - ilo + 1, np_ = (len + 1) / 2;
        const float4 *wp = wl + (size_t)sw[t].pos * 64;
        float bcur = Rv[ilo * 64];
        float4 cA[RC / 2], cB[RC / 2];
        // chunk k holds the pairs [k * RC / 2, (k + 1) * RC / 2): the last chunk first
        int k = (np_ - 1) / (RC / 2);
#pragma unroll
        for (int u = 0; u < RC / 2; ++u) cA[u] = wp[(size_t)min(k * (RC / 2) + u, np_ - 1) * 64];
        for (; k >= 0; --k) {
            if (k > 0) {
#pragma unroll
                for (int u = 0; u < RC / 2; ++u) cB[u] = wp[(size_t)((k - 1) * (RC / 2) + u) * 64];
            }
#pragma unroll
            for (int u = RC - 1; u >= 0; --u) {
                const int tr = k * RC + u; // trip index of the forward order
                if (tr < len) {
                    const int i = ihi - tr;
                    const float cc = (u & 1) ? cA[u >> 1].z : cA[u >> 1].x, ss = (u & 1) ? cA[u >> 1].w : cA[u >> 1].y;
                    const float bj = Rv[(i + 1) * 64];
                    Rv[i * 64] = cc * bcur + ss * bj;
                    bcur = cc * bj - ss * bcur;
                }
            }
#pragma unroll
            for (int u = 0; u < RC / 2; ++u) cA[u] = cB[u];
        }
        Rv[(ihi + 1) * 64] = bcur;
    }
    for (int q = 0; q < n; ++q) gT[(size_t)q * NB] = Rv[q * 64];
    if (live) fail[b] = bad ? 1 : 0;
    if (stats) { // (measurement: sweeps and trips of the wave, rotations of the lane's own blocks)
        for (int off = 32; off > 0; off >>= 1) lane_rot += __shfl_xor(lane_rot, off, 64);
        const long long t_end = (long long)__builtin_readcyclecounter();
        long long n1 = live && mode == 1, n2 = live && mode == 2, ls = live ? lstop : 0;
        for (int off = 32; off > 0; off >>= 1) { n1 += __shfl_xor(n1, off, 64); n2 += __shfl_xor(n2, off, 64); ls += __shfl_xor(ls, off, 64); }
        if (lane == 0) { stats[9 * blockIdx.x] = nsw; stats[9 * blockIdx.x + 1] = wave_trips; stats[9 * blockIdx.x + 2] = lane_rot;
                         stats[9 * blockIdx.x + 3] = t_ql - t_start; stats[9 * blockIdx.x + 4] = t_fwd - t_ql; stats[9 * blockIdx.x + 5] = t_end - t_fwd;
                         stats[9 * blockIdx.x + 6] = n1; stats[9 * blockIdx.x + 7] = n2; stats[9 * blockIdx.x + 8] = ls; }
    }
}

// y = H_0 H_1 ... H_{n-2} yt (reflector j in row j of the matrix's scratch image, tau in the workspace), one wave per matrix; step row
// written, flag cleared -- unless the QL kernel gave the matrix up (fail[b]: its flag stays for the caller's fallback).
template <int NP>
__global__ __launch_bounds__(256) void eig_backtransform_kernel(const float *R, const int *idx, int nb, int n, const float *ws,
                                                                int64_t NB, const int *fail, float *step, int *flags, float *sens_out) {
    constexpr int NW = NP / 64;
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nb) return;
    const int m_idx = idx ? idx[b] : b;
    if (sens_out && lane == 0) sens_out[m_idx] = fail[b] ? 1.0f : ws[(size_t)4 * NP * NB + b]; // (a matrix the iteration gave up on: refine it)
    if (fail[b]) return;
    const float *mat = R + (int64_t)b * NP * NP;
    const float *gT = ws + (size_t)2 * NP * NB, *tau = ws + (size_t)3 * NP * NB + (size_t)b * NP;
    float y[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const int r = q * 64 + lane;
        y[q] = (r < n) ? gT[(size_t)r * NB + b] : 0.f;
    }
    constexpr int PF = 8; // reflector rows in flight
    float v[PF][NW];
    int j = n - 2;
    for (; j >= 0; j -= PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int q = 0; q < NW; ++q) v[u][q] = (j - u >= 0) ? mat[(int64_t)(j - u) * NP + q * 64 + lane] : 0.f;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (j - u < 0) break;
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < NW; ++q) dot += v[u][q] * y[q];
            dot = wave_sum(dot) * tau[j - u];
#pragma unroll
            for (int q = 0; q < NW; ++q) y[q] -= dot * v[u][q];
        }
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const int r = q * 64 + lane;
        if (r < NP) step[(int64_t)m_idx * NP + r] = (r < n) ? y[q] : 0.f;
    }
    if (lane == 0 && flags) flags[m_idx] = 0;
}

} // namespace cmfk
