// cmf_rank1clamp.hip.h -- _safe_invert's clamp (pycmf/cmf_solvers.py:346-356) where ONE eigenvalue of the per-row Hessian stands
// above the threshold and all the others below it (round 6; test infrastructure is elsewhere: this is product code).
//
// That is the steady state of BASELINE configs[2] at the reference's default l2 = 0 (tools/r06_spectrum_probe.py: one eigenvalue of
// ~2e3 / 5e2, 255 below pert = 0.2 on every row of U and Z).  There
//     safe_inverse(H) = Q diag(1 / max(|lambda|, pert)) Q^T = (I - q q^T) / pert + q q^T / max(lambda_1, pert)
// needs the top eigenpair (lambda_1, q) only -- a few matrix-vector products -- and a PROOF that nothing else reaches the threshold:
// H is positive semi-definite by construction (weights >= 0; the caller checks), so H' = H - lambda_1 q q^T has no eigenvalue below
// -eps, and theta I - H' positive definite (one Cholesky factorisation, the blocked one of cmf_chol_mfma.hip.h in its test-only
// mode) says every other eigenvalue lies below theta = pert - delta, delta = 4 * 2^-23 lambda_1 (the margin the tridiagonal solver's
// error bound uses, cmf_eigclamp.hip.h: a row whose spectrum comes closer to the threshold than float32 resolves is not served).
// A row that fails either test (the iteration has not converged: a second large eigenvalue; the certificate does not hold) keeps
// its flag and goes through the tridiagonal eigen-solve as before.  Every decision is the row's own: the route a row takes does not
// depend on the batch it arrives in (row-sharded runs stay bit-identical to single-device ones).  Per 8192 rows at n = 256: ~ 3.5 ms against 19.5.
//
//   rank1_power_kernel   one workgroup per flagged matrix: power iteration from the row of the largest diagonal entry (thread t owns column t; H is
//                        symmetric, so the product is a coalesced sweep down the rows), Rayleigh quotient and residual of the
//                        last product, then the image A = theta I - H + lambda q q^T for the certificate
//   rank1_compose_kernel rows whose certificate holds: step = (g - (q.g) q) / pert + (q.g) q / max(lambda, pert), flag cleared
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfk {

__device__ __forceinline__ float r1_block_sum(float v, float *red, int t) { // 256 threads; red: 8 floats; two barriers
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// H: [.. x kp x kp] (stride floats apart), idx: the matrices served (null: 0 .. nf - 1); A: [nf][kp * kp] images; q: [nf][kp];
// lam: [nf]; ok: [nf] (1: converged; the certificate is still to come)
__global__ __launch_bounds__(256) void rank1_power_kernel(const float *H, const int *idx, int n, int kp, int64_t stride, float pert, int passes,
                                                          float res_tol, float *A, float *qout, float *lam, int *ok, unsigned long long *nok) {
    __shared__ float x[256];
    __shared__ float red[8];
    const int b = blockIdx.x, t = threadIdx.x;
    const float *Hm = H + (int64_t)(idx ? idx[b] : b) * stride;
    const bool live = t < n;
    // start: the row of the largest diagonal entry, normalised -- one power step from e_j taken for free (H e_j is a contiguous row),
    // and q_j^2 is largest where H_jj ~ lambda_1 q_j^2 is
    __shared__ int jstar;
    {
        float dv = live ? Hm[(int64_t)t * kp + t] : -1.0f;
        int di = t;
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(dv, off, 64);
            const int oi = __shfl_xor(di, off, 64);
            if (ov > dv || (ov == dv && oi < di)) { dv = ov; di = oi; }
        }
        if ((t & 63) == 0) { red[t >> 6] = dv; red[4 + (t >> 6)] = __int_as_float(di); }
        __syncthreads();
        if (t == 0) {
            float bv = red[0];
            int bi = __float_as_int(red[4]);
            for (int w = 1; w < 4; ++w)
                if (red[w] > bv || (red[w] == bv && __float_as_int(red[4 + w]) < bi)) { bv = red[w]; bi = __float_as_int(red[4 + w]); }
            jstar = bi;
        }
        __syncthreads();
        const float h = live ? Hm[(int64_t)jstar * kp + t] : 0.f;
        const float n2 = r1_block_sum(h * h, red, t);
        x[t] = n2 > 0.f ? h / sqrtf(n2) : 0.f;
    }
    float p = 0.f, lambda = 0.f, res2 = 0.f, xt = 0.f;
    for (int it = 0; it < passes; ++it) {
        __syncthreads();
        xt = x[t];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (live) {
            const float *col = Hm + t;
            int r = 0;
            for (; r + 8 <= n; r += 8) { // eight loads in flight; four accumulators
                const float h0 = col[(int64_t)(r + 0) * kp], h1 = col[(int64_t)(r + 1) * kp], h2 = col[(int64_t)(r + 2) * kp], h3 = col[(int64_t)(r + 3) * kp];
                const float h4 = col[(int64_t)(r + 4) * kp], h5 = col[(int64_t)(r + 5) * kp], h6 = col[(int64_t)(r + 6) * kp], h7 = col[(int64_t)(r + 7) * kp];
                a0 = fmaf(h0, x[r + 0], a0); a1 = fmaf(h1, x[r + 1], a1); a2 = fmaf(h2, x[r + 2], a2); a3 = fmaf(h3, x[r + 3], a3);
                a0 = fmaf(h4, x[r + 4], a0); a1 = fmaf(h5, x[r + 5], a1); a2 = fmaf(h6, x[r + 6], a2); a3 = fmaf(h7, x[r + 7], a3);
            }
            for (; r < n; ++r) a0 = fmaf(col[(int64_t)r * kp], x[r], a0);
        }
        p = (a0 + a1) + (a2 + a3);
        const float nrm2 = r1_block_sum(p * p, red, t);
        if (it == 0 && passes > 1) {
            // screening, per matrix (so that WHICH route serves a row never depends on what else is in the batch): after the one step
            // the start already took (tan of e_j's angle to q: up to sqrt(n)) an eigenvalue dominant enough for the final test
            // (lambda_2 / lambda_1 < 9e-3: three steps to 1e-5) leaves a relative residual ~ tan(angle) < 0.15; a spectrum without one
            // (C3X: lambda_2 / lambda_1 > 0.1) leaves more than 1 -- that matrix stops here, one sweep in instead of five
            const float l1 = r1_block_sum(xt * p, red, t);
            const float d1 = p - l1 * xt;
            const float r1 = r1_block_sum(d1 * d1, red, t);
            if (!(l1 > 0.f) || r1 > 0.09f * l1 * l1) {
                if (t == 0) { lam[b] = l1; ok[b] = 0; }
                return;
            }
        }
        if (it + 1 < passes) {
            __syncthreads();
            x[t] = nrm2 > 0.f ? p / sqrtf(nrm2) : 0.f;
        } else { // the pair of the last product: (x^T H x, x), residual ||H x - lambda x||
            lambda = r1_block_sum(xt * p, red, t);
            const float d = p - lambda * xt;
            res2 = r1_block_sum(d * d, red, t);
        }
    }
    const bool good = lambda > 0.f && res2 <= res_tol * res_tol * lambda * lambda;
    if (t == 0) {
        lam[b] = lambda;
        ok[b] = good ? 1 : 0;
        if (good) atomicAdd(nok, 1ull);
    }
    if (t < kp) qout[(int64_t)b * kp + t] = live ? xt : 0.f;
    if (!good) return; // (wave-uniform: every thread holds the same sums) -- no image, the certificate is skipped by ok[b] = 0
    // A = theta I - H + lambda x x^T on the n x n part (the factorisation pads the rest itself)
    const float theta = pert - 4.0f * 1.1920929e-7f * lambda; // (delta as cmf_eigclamp.hip.h takes it)
    float *Ab = A + (int64_t)b * kp * kp;
    if (live) {
        const float lx = lambda * xt;
        for (int r = 0; r < n; ++r) {
            const float h = Hm[(int64_t)r * kp + t];
            Ab[(int64_t)r * kp + t] = fmaf(lx, x[r], (r == t ? theta : 0.f) - h);
        }
    }
}

// cert[b] == 0: the certificate of matrix b holds (chol_solve_mfma_kernel, test only).  One wave per matrix.
__global__ __launch_bounds__(256) void rank1_compose_kernel(const int *idx, int nf, int n, int kp, float pert, const float *qin, const float *lam,
                                                            const int *ok, const int *cert, const float *grad, float *step, int *flags, float *sens,
                                                            unsigned long long *served) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= nf) return;
    if (!ok[b] || cert[b] != 0) return;
    const int64_t row = idx ? idx[b] : b;
    const float *q = qin + (int64_t)b * kp, *g = grad + row * kp;
    float dot = 0.f;
    for (int i = lane; i < n; i += 64) dot = fmaf(q[i], g[i], dot);
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    const float l1 = lam[b];
    const float ip = 1.0f / pert, il = 1.0f / fmaxf(l1, pert);
    float st2 = 0.f;
    for (int i = lane; i < kp; i += 64) {
        float s = 0.f;
        if (i < n) s = (g[i] - dot * q[i]) * ip + (dot * il) * q[i];
        st2 = fmaf(s, s, st2);
        step[row * kp + i] = s;
    }
    for (int off = 32; off > 0; off >>= 1) st2 += __shfl_xor(st2, off, 64);
    if (lane == 0) {
        flags[row] = 0;
        // the bound the tridiagonal solver reports (cmf_eigclamp.hip.h): delta / lambda_1 for the one eigenvalue that is not clamped,
        // plus (delta / pert) ||step along q|| / ||step|| for the rotation of q against the clamped subspace
        const float delta = 4.0f * 1.1920929e-7f * l1;
        if (sens) sens[row] = l1 >= pert - delta ? delta / fmaxf(l1, 1e-30f) + (delta / pert) * (fabsf(dot) * il) * __builtin_amdgcn_rsqf(fmaxf(st2, 1e-37f)) : 0.f;
        atomicAdd(served, 1ull);
    }
}

} // namespace cmfk
