// cmf_sparse.hip.h -- CSR kernels for sparse X / Y (gfx950), HBM-bound.
//
// The reference accepts scipy CSR/CSC (pycmf/cmf.py:679) and multiplies through
// sklearn's safe_sparse_dot (pycmf/cmf_solvers.py:232,238,244).  Here a sparse data
// matrix is kept twice on the device, as CSR of A and as CSR of A^T, so that both
// A*F and A^T*F are row-gather SpMMs:  out[r,:] = sum_q val[q] * F[idx[q],:].
// A group of k_pad/4 lanes owns one output row (float4 per lane = one coalesced
// 16-B-per-lane read of a factor row per nonzero); a 64-lane wave carries
// 64/(k_pad/4) rows.  No atomics, deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfk {

struct CsrView {
    const int64_t *indptr; // rows + 1
    const int32_t *idx;    // column of each nonzero
    const float *val;
    int64_t rows;
};

// GL = lanes per row group = min(64, kp/4); CH = float4 chunks per lane (kp/4/GL)
template <int GL, int CH>
__global__ __launch_bounds__(256) void spmm_csr_kernel(CsrView A, const float *F, int kp, float *out, int accumulate) {
    constexpr int RPW = 64 / GL; // rows per wave
    const int lane = threadIdx.x & 63;
    const int gl = lane % GL, gsub = lane / GL;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t row = wave * RPW + gsub;
    if (row >= A.rows) return;
    const int64_t beg = A.indptr[row], end = A.indptr[row + 1];
    f32x4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int64_t q = beg;
    for (; q + 4 <= end; q += 4) { // 4 independent gathers in flight per lane
        int32_t j[4];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { j[u] = A.idx[q + u]; v[u] = A.val[q + u]; }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            f32x4 f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) f[u] = *reinterpret_cast<const f32x4 *>(F + (int64_t)j[u] * kp + 4 * (gl + GL * c));
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[c] += v[u] * f[u];
        }
    }
    for (; q < end; ++q) {
        const int32_t j = A.idx[q];
        const float v = A.val[q];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] += v * *reinterpret_cast<const f32x4 *>(F + (int64_t)j * kp + 4 * (gl + GL * c));
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(out + row * kp + 4 * (gl + GL * c));
        *dst = accumulate ? (*dst + acc[c]) : acc[c];
    }
}

// Target term of a per-row Newton gradient on a NATIVE sparse side (cmf_newton.hip.h): the row kernel ran with zero targets, i.e.
// it accumulated  s sum_{j in S_i} f(f_i . o_j) o_j;  the reference's residual is f(.) - t_ij (pycmf/cmf_solvers.py:419-420,
// :459-461, :495-497), so what is missing is  - s sum_{j in S_i, t_ij != 0} t_ij o_j  -- a row-gather SpMM over the stored values
// of data row i that lie in its sample S_i (all of them when the sweep is not sampled).  `lists`: the rows' ASCENDING index
// lists (per entries each) or null; membership by binary search.  grad rows are relative to row0.
template <int GL, int CH>
__global__ __launch_bounds__(256) void csr_sampled_sub_kernel(CsrView A, const float *F, int kp, float *grad, float scale, int64_t row0,
                                                              int64_t nrows, const int32_t *lists, int64_t per) {
    constexpr int RPW = 64 / GL;
    const int lane = threadIdx.x & 63;
    const int gl = lane % GL, gsub = lane / GL;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t lr = wave * RPW + gsub;
    if (lr >= nrows) return;
    const int64_t row = row0 + lr;
    const int64_t beg = A.indptr[row], end = A.indptr[row + 1];
    const int32_t *L = lists ? lists + row * per : nullptr;
    f32x4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t q = beg; q < end; ++q) {
        const int32_t j = A.idx[q];
        if (L) { // j in S_i ?
            int64_t lo = 0, hi = per;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (L[mid] < j) lo = mid + 1;
                else hi = mid;
            }
            if (lo >= per || L[lo] != j) continue;
        }
        const float v = A.val[q];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] += v * *reinterpret_cast<const f32x4 *>(F + (int64_t)j * kp + 4 * (gl + GL * c));
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(grad + lr * kp + 4 * (gl + GL * c));
        *dst = *dst - scale * acc[c];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Output-stationary, column-blocked SpMM for matrices whose gathered operand does not fit the XCD's 4 MB L2
// (C5: X V gathers 1 KB rows of a 102 MB V, X^T U of a 1 GB U; one gather per non-zero served by the Infinity Cache /
// HBM ran at 7.4 TB/s, 52x the compulsory bytes).  Here the non-zeros are regrouped on the host into
//   row GROUPS  (<= G consecutive rows, G k_pad floats = 128 KB of LDS accumulators, closed early when a group gets
//                more than its share of non-zeros: nnz-balanced work items), and
//   column BLOCKS (B gathered rows = 3 MB of the factor),
// entries sorted by (group, owner wave, block, row).  A persistent 512-thread workgroup per CU takes one group at a time,
// keeps its G output rows in LDS for the whole pass and sweeps the column blocks IN ORDER; the workgroups that share an
// XCD (blockIdx % 8, the dispatcher's observed round-robin: a speed assumption only) start every group together
// (bounded, timing-only counter barrier), so at any moment they gather from the same 3 MB slab, which stays in their
// L2: every factor row is fetched from the fabric once per XCD and round instead of once per non-zero
// (reuse = rows in flight per XCD x density = 4096 x 1e-3 = 4.1 at C5: the FIRST touch of a factor row in a round always misses,
// so the L2 hit rate of the gathers is bounded by 1 - 1 / 4.1 = 76 % on uniformly scattered non-zeros -- measured 66 % -- and
// only more output rows in flight per XCD could raise it: they are bounded by the 128 KB of LDS accumulators per CU).
// Wave w owns the rows with row % 8 == w: no two waves touch one accumulator, the sum order inside a row is the order
// of the entry list -- deterministic, no atomics.  Entry metadata is wave-uniform and comes through the scalar cache.
struct BcsrEntry {
    int32_t col;   // gathered row of the factor
    int32_t rowl;  // output row inside the group
    float val;
    int32_t pad;
};
struct BcsrView {
    const BcsrEntry *ent;
    const int64_t *seg;     // [(g * 8 + w) * nsync + i] -> first entry of stretch i of wave w's list for group g; one extra at the end
    const int32_t *grow;    // first accumulator row of group g; ngroups + 1 entries
    int ngroups;
    int nsync;              // stretches per list: the class re-aligns between stretches (every few column blocks)
    const int32_t *vmap;    // accumulator row -> output row (>= 0), or -(1 + slot): a piece of a split row, written to part[slot]
    float *part;            // partial output rows of the pieces (summed by spmm_partial_sum_kernel)
};
// Round 6: accumulator rows are no longer output rows one to one.  A row whose non-zeros exceed a wave's fair share of a group (the hot
// words of a bag-of-words matrix: Zipf(1.1) columns put 13 % of the documents on the first word, and X^T U has the words as rows) is cut
// into pieces of at most that share; every piece is an accumulator row of its own, possibly in another group (another workgroup),
// and the waves of a group take its accumulator rows longest-first (the host's greedy assignment: the wave of a row is that of its
// entries, no longer row % 8).  Pieces write to a partial buffer; spmm_partial_sum_kernel adds the pieces of a row in order:
// deterministic, no atomics.  Measured on the Zipf(1.1) variant of C5 (c5z): 72 -> see DESIGN.md section 8 ms per launch; uniform
// columns (c5) have nothing to split and keep their layout.

constexpr int BCSR_NW = 8;  // waves per workgroup (host layout and kernel agree on it)
template <int VEC> // floats per lane; k_pad = 64 * VEC
// `accumulate`: bit 0 out += ; bit 1: the output IS a Newton factor update (re-associated sweep, cmf_newton.hip.h): negatives
// clamp to 0; kvalid > 0: columns >= kvalid are written as 0.
__global__ __launch_bounds__(64 * BCSR_NW) void spmm_blocked_kernel(BcsrView A, const float *F, float *out, int accumulate, unsigned *bar, int kvalid) {
    typedef float vec __attribute__((ext_vector_type(VEC)));
    constexpr int KP = 64 * VEC;
    constexpr int CH = BCSR_NW == 8 ? 16 : 8; // gathers issued together; two chunks are in flight (32 KB per wave at k_pad = 256)
    extern __shared__ __attribute__((aligned(16))) float lacc[]; // [rows of the group][KP]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int groups_q = (A.ngroups - q + 7) / 8; // groups of this XCD class: q, q + 8, ...
    unsigned target = 0;
    for (int round = 0;; ++round) {
        const int gi = j + per * round; // index inside the class
        if (gi >= groups_q) break;
        const int g = q + 8 * gi;
        const int r0 = A.grow[g], nrows = A.grow[g + 1] - r0;
        for (int r = w; r < nrows; r += BCSR_NW) {
            vec z;
#pragma unroll
            for (int e = 0; e < VEC; ++e) z[e] = 0.f;
            *reinterpret_cast<vec *>(lacc + r * KP + lane * VEC) = z;
        }
        const int left = groups_q - per * round;
        const unsigned active = (unsigned)(left < per ? left : per); // workgroups of the class that work in this round
        // this wave's entry list for the group: sorted by (column block, row, column), cut into nsync stretches of a few
        // column blocks.  The metadata of 64 entries arrives by ONE coalesced vector load (lane u holds entry u) and is
        // broadcast with v_readlane; gathers go out 16 at a time, the next chunk's before the current chunk is consumed.
        const int64_t *segw = A.seg + ((int64_t)g * BCSR_NW + w) * A.nsync;
        int cur = -1;
        vec racc;
#pragma unroll
        for (int e = 0; e < VEC; ++e) racc[e] = 0.f;
        auto flush = [&]() {
            if (cur >= 0) {
                vec *p = reinterpret_cast<vec *>(lacc + cur * KP + lane * VEC);
                *p = *p + racc;
            }
        };
        for (int st = 0; st < A.nsync; ++st) {
        // timing-only rendezvous of the class in front of every stretch: the workgroups that share an L2 gather from the
        // same few column blocks at the same time (a long sweep drifts apart by many blocks otherwise: X^T U at C5 walks
        // 489 blocks in 4.5 ms).  No data passes through it; on a timeout (GPU shared with another process, placement
        // not as assumed) only L2 locality is lost.
        target += active;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(bar + 4 * q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int spin = 0; spin < 4000; ++spin) {
                if (__hip_atomic_load(bar + 4 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __syncthreads();
        const int64_t s0 = segw[st], s1 = segw[st + 1];
        for (int64_t e0 = s0; e0 < s1; e0 += 64) {
            const int n = (int)(s1 - e0 < 64 ? s1 - e0 : 64);
            BcsrEntry me;
            me.col = 0; me.rowl = 0; me.val = 0.f; me.pad = 0;
            if (lane < n) me = A.ent[e0 + lane];
            vec x[2][CH];
            auto issue = [&](int c0, vec *dst) {
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (c0 + u < n) {
                        const int col = __builtin_amdgcn_readlane(me.col, c0 + u);
                        dst[u] = *reinterpret_cast<const vec *>(F + (int64_t)col * KP + lane * VEC);
                    }
            };
            issue(0, x[0]);
#pragma unroll
            for (int c = 0; c < 64 / CH; ++c) {
                const int c0 = c * CH;
                if (c0 >= n) break;
                if (c + 1 < 64 / CH) issue(c0 + CH, x[(c + 1) & 1]);
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (c0 + u < n) {
                        const int rl = __builtin_amdgcn_readlane(me.rowl, c0 + u);
                        const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.val), c0 + u));
                        if (rl != cur) {
                            flush();
                            cur = rl;
#pragma unroll
                            for (int e = 0; e < VEC; ++e) racc[e] = 0.f;
                        }
                        racc += v * x[c & 1][u];
                    }
            }
        }
        }
        flush();
        __syncthreads(); // (an accumulator row is summed by the wave the host gave it to, written back by wave r % 8)
        for (int r = w; r < nrows; r += BCSR_NW) {
            vec v = *reinterpret_cast<const vec *>(lacc + r * KP + lane * VEC);
            const int orow = A.vmap[r0 + r];
            if (orow < 0) { // a piece of a split row: the partial sum as it is
                *reinterpret_cast<vec *>(A.part + (int64_t)(-orow - 1) * KP + lane * VEC) = v;
                continue;
            }
            vec *dst = reinterpret_cast<vec *>(out + (int64_t)orow * KP + lane * VEC);
            if (accumulate & 1) v += *dst;
            if (kvalid > 0 || (accumulate & 2)) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if (kvalid > 0 && lane * VEC + e >= kvalid) v[e] = 0.f;
                    else if ((accumulate & 2) && v[e] < 0.f) v[e] = 0.f;
                }
            }
            *dst = v;
        }
    }
}

// out[prow[i]] (+)= sum of the partial rows first[i] .. first[i] + cnt[i] (in that order), with the epilogue of spmm_blocked_kernel
template <int VEC>
__global__ __launch_bounds__(64) void spmm_partial_sum_kernel(const float *part, const int32_t *prow, const int32_t *first, const int32_t *cnt, float *out,
                                                              int accumulate, int kvalid) {
    typedef float vec __attribute__((ext_vector_type(VEC)));
    constexpr int KP = 64 * VEC;
    const int i = blockIdx.x, lane = threadIdx.x;
    vec v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = 0.f;
    for (int s = 0; s < cnt[i]; ++s) v += *reinterpret_cast<const vec *>(part + (int64_t)(first[i] + s) * KP + lane * VEC);
    vec *dst = reinterpret_cast<vec *>(out + (int64_t)prow[i] * KP + lane * VEC);
    if (accumulate & 1) v += *dst;
    if (kvalid > 0 || (accumulate & 2)) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (kvalid > 0 && lane * VEC + e >= kvalid) v[e] = 0.f;
            else if ((accumulate & 2) && v[e] < 0.f) v[e] = 0.f;
        }
    }
    *dst = v;
}

// cross[wg] = sum over the nonzeros of this workgroup's rows of  a_ij * (L_i . R_j)
// (the 2 tr((A R)^T L) term of the expanded Frobenius error, sklearn _beta_divergence
// sparse branch used at pycmf/cmf_solvers.py:40)
// logit != 0: a_ij * sigmoid(L_i . R_j) instead -- the cross term of ||A - sigmoid(L R^T)||^2 = sum_all sigmoid^2 + sum_nnz
// (a^2 - 2 a sigmoid), whose first part is a dense pass with zero targets (pycmf/cmf_solvers.py:42 on a sparse target)
template <int GL, int CH>
__global__ __launch_bounds__(256) void sddmm_cross_kernel(CsrView A, const float *L, const float *R, int kp, double *partials, int logit) {
    constexpr int RPW = 64 / GL;
    const int lane = threadIdx.x & 63;
    const int gl = lane % GL, gsub = lane / GL;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t row = wave * RPW + gsub;
    float acc = 0.f;
    if (row < A.rows) {
        f32x4 l[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) l[c] = *reinterpret_cast<const f32x4 *>(L + row * kp + 4 * (gl + GL * c));
        const int64_t beg = A.indptr[row], end = A.indptr[row + 1];
        for (int64_t q = beg; q < end; ++q) {
            const int32_t j = A.idx[q];
            const float v = A.val[q];
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const f32x4 r = *reinterpret_cast<const f32x4 *>(R + (int64_t)j * kp + 4 * (gl + GL * c));
                dot += l[c][0] * r[0] + l[c][1] * r[1] + l[c][2] * r[2] + l[c][3] * r[3];
            }
            if (logit) { // the lanes of a row group hold partial dot products: finish the sum before the sigmoid
                float full = dot;
                for (int off = GL / 2; off > 0; off >>= 1) full += __shfl_xor(full, off, 64);
                acc += (gl == 0) ? v * (1.0f / (1.0f + __expf(-full))) : 0.f;
            } else
            acc += v * dot;
        }
    }
    double v = (double)acc;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __shared__ double red[4];
    if (lane == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// dense[r][idx] += val : expand CSR into the padded dense layout (per-row Newton images)
__global__ void csr_to_dense_kernel(CsrView A, float *dense, int64_t ld) {
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= A.rows) return;
    const int lane = threadIdx.x & 63;
    for (int64_t q = A.indptr[row] + lane; q < A.indptr[row + 1]; q += 64) dense[row * ld + A.idx[q]] += A.val[q];
}

// sum_ab A[a][b] * B[a][b] over kp x kp (fp64 accumulate, single block)
__global__ void frob_inner_kernel(const float *A, const float *B, int n, double *out) {
    __shared__ double red[256];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += (double)A[i] * (double)B[i];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

} // namespace cmfk
