// cmf_rowhess6.hip.h -- row_hess_kernel<256> on the bf16 matrix pipe (part of the OPTIONAL arithmetic, gemm_arith = 1).
//
// Same contract as row_hess_kernel<256, ., ., 3> (cmf_rowhess.hip.h): for one factor row f_i and its sampled rows o_j
//     g_i = sum_j r_j o_j          H_i = sum_j (sqrt(w_j) o_j)(sqrt(w_j) o_j)^T        (weights >= 0)
// but the staged row sqrt(w_j) o_j is split exactly into three bf16 planes (cmf_bf16x6.hip.h) while it is written to
// LDS, and the 36 upper 32 x 32 blocks of H_i accumulate six cross products each on v_mfma_f32_32x32x16_bf16.  The
// contraction index j is the SLOW axis of the row-major tile, so the k-contiguous operands are fetched with
// ds_read_b64_tr_b16 (a 16-lane group reads 4 rows x 16 columns and every lane receives one column: probe and
// standalone study in tools/spike/).  Gradient, dot product, link and weights stay in fp32 registers as before.
// Every lambda is force-inlined: an out-of-line generic lambda would keep all by-reference captures in scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfk {

constexpr int R6_PITCH = 256 + 32;              // bf16 elements per LDS row: 576 bytes = 16 banks mod 64, so the four rows a
                                                // 16-lane group addresses in one transposing read fall on disjoint bank quarters
constexpr int R6_PLANE = 32 * R6_PITCH;         // elements per plane of a 32-row tile
constexpr int R6_STAGE = 3 * R6_PLANE;
constexpr int R6_LDS_BYTES = 2 * R6_STAGE * 2;  // 110592

__device__ __forceinline__ uint64_t r6_tr_read(unsigned addr) {
    uint64_t v;
    // volatile keeps it in place relative to the barriers; no "memory" clobber (it would pin every by-reference
    // captured local, the kernel argument struct included, in scratch)
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

__global__ __launch_bounds__(512, 2) void row_hess6_kernel(RowHessArgs g) {
    constexpr int KP = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned short r6l[];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, l31 = lane & 31, lh = lane >> 5;
    const int uw = __builtin_amdgcn_readfirstlane(wid);
    const int wty = uw < 4 ? 0 : ((uw & 1) ? 2 : 1), sbase = uw >= 6 ? 4 : 0;
    const bool late = uw >= 4;
    int fblk[5]; // block columns whose fragments this wave reads (s3_* maps of cmf_rowhess.hip.h)
    if (wty == 0) { fblk[0] = uw; fblk[1] = 4; fblk[2] = 5; fblk[3] = 6; fblk[4] = 7; }
    else if (wty == 1) { fblk[0] = sbase; fblk[1] = sbase + 1; fblk[2] = sbase + 2; fblk[3] = sbase + 3; fblk[4] = sbase + 3; }
    else { fblk[0] = sbase + 1; fblk[1] = sbase + 2; fblk[2] = sbase + 3; fblk[3] = sbase + 3; fblk[4] = sbase + 3; }

    const bool cls = g.cls_cnt != nullptr; // class mode, see cmf_rowhess.hip.h
    const int64_t i = cls ? 0 : g.row0 + blockIdx.x;
    const int ns = cls ? g.cls_cnt[blockIdx.x] : g.s;
    const int trow = t >> 4, tl16 = t & 15; // staging: one gathered row per thread, 4 chunks of 4 floats at columns 4 (16 q + tl16)
    f32x4 u4[4], gacc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u4[q] = *reinterpret_cast<const f32x4 *>(g.F + i * KP + 4 * (16 * q + tl16));
        gacc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int32_t *list = cls ? g.idx + g.cls_off[blockIdx.x] : (g.idx ? g.idx + i * g.idx_stride : nullptr);
    const float *Ti = g.T + i * g.t_row;
    const int nt = (ns + 31) / 32;
    f32x16 hs[5];
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) hs[n][r] = 0.f;

    int jn = 0;
    f32x4 rr[4];
    float tt = 0.f;
    bool vv = false;
    auto load_idx = [&](int tl) __attribute__((always_inline)) {
        const int q = 32 * tl + trow;
        const int qc = q < ns ? q : ns - 1;
        jn = list ? list[qc] : qc;
    };
    auto gather = [&](int tl) __attribute__((always_inline)) {
        vv = 32 * tl + trow < ns;
        const float *src = g.O + (int64_t)jn * KP + 4 * tl16;
#pragma unroll
        for (int q = 0; q < 4; ++q) rr[q] = *reinterpret_cast<const f32x4 *>(src + 64 * q);
        tt = Ti[(int64_t)jn * g.t_col];
    };
    const float lk = g.link ? 1.0f : 0.0f, nlk = 1.0f - lk;
    auto stage = [&](int buf) __attribute__((always_inline)) {
        float z = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) z += u4[q][0] * rr[q][0] + u4[q][1] * rr[q][1] + u4[q][2] * rr[q][2] + u4[q][3] * rr[q][3];
        z = group_sum<16>(z);
        float slope;
        const float sg = sigmoid_slope_(z, slope);   // (the slope without the cancellation of s (1 - s): cmf_kernels.hip.h)
        const float f = lk * sg + nlk * z;
        const float valid = vv ? g.scale : 0.0f;
        const float res = valid * (f - tt);
        const float wgt = valid * (lk * slope + nlk);
        const float sqw = __builtin_amdgcn_sqrtf(fmaxf(wgt, 0.0f));
        unsigned short *base = r6l + buf * R6_STAGE + trow * R6_PITCH + 4 * tl16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gacc[q] += res * rr[q];
            uint64_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const Bf16x3 sp = split3(sqw * rr[q][e]);
                w0 |= (uint64_t)sp.a << (16 * e); w1 |= (uint64_t)sp.b << (16 * e); w2 |= (uint64_t)sp.c << (16 * e);
            }
            *reinterpret_cast<uint64_t *>(base + 64 * q) = w0;
            *reinterpret_cast<uint64_t *>(base + R6_PLANE + 64 * q) = w1;
            *reinterpret_cast<uint64_t *>(base + 2 * R6_PLANE + 64 * q) = w2;
        }
    };
    // transposing fragment read: plane p, 32-column block blk, k16 sub-step ks (tile rows 16 ks .. 16 ks + 15)
    const int g16 = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    auto frag = [&](int buf, int p, int blk, int ks) __attribute__((always_inline)) -> bf16x8 {
        const unsigned short *rowp = r6l + buf * R6_STAGE + p * R6_PLANE + (16 * ks + 8 * lh + tq) * R6_PITCH + 32 * blk + 16 * (g16 & 1) + 4 * tp;
        union { uint64_t u[2]; bf16x8 v; } x;
        x.u[0] = r6_tr_read((unsigned)(uintptr_t)rowp);
        x.u[1] = r6_tr_read((unsigned)(uintptr_t)(rowp + 4 * R6_PITCH));
        return x.v;
    };
    auto tile = [&](auto typ, int buf, bool do_stage, bool do_gather, int tl_gather, bool do_idx, int tl_idx) __attribute__((always_inline)) {
        constexpr int TY = decltype(typ)::value;
        constexpr int NF = s3_nf(TY), NP = sym_np(TY);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fr[3][5];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < NF; ++q) fr[p][q] = frag(buf, p, fblk[q], ks);
            // waves 4-7 run their VALU-heavy staging one k16 sub-step after waves 0-3, so that on every SIMD one wave's
            // splitting sits beside its partner's MFMAs
            if (ks == (late ? 1 : 0) && do_stage) stage(buf ^ 1);
            if (ks == 1) {
                if (do_gather) gather(tl_gather);
                if (do_idx) load_idx(tl_idx);
            }
            // The transposing reads are invisible to the compiler's counters: wait by hand, plane by plane (the reads were
            // issued plane 0, 1, 2 and LDS operations complete in order; at most 2 NF <= 10 reads per plane, so
            // "<= 2 NF (2 - p) outstanding" means plane p has landed -- later staging stores only make that conservative).
            // Largest terms first here, so that the first MFMAs need plane 0 only.
            if constexpr (NF == 5) asm volatile("s_waitcnt lgkmcnt(15)"); // 4-bit counter: 20 is not expressible
            else if constexpr (NF == 4) asm volatile("s_waitcnt lgkmcnt(15)");
            else asm volatile("s_waitcnt lgkmcnt(12)");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NP; ++n)
                hs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0][s3_ai(TY, n)], fr[0][s3_bi(TY, n)], hs[n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NF == 5) asm volatile("s_waitcnt lgkmcnt(10)");
            else if constexpr (NF == 4) asm volatile("s_waitcnt lgkmcnt(8)");
            else asm volatile("s_waitcnt lgkmcnt(6)");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const int ai = s3_ai(TY, n), bi = s3_bi(TY, n);
                hs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0][ai], fr[1][bi], hs[n], 0, 0, 0);
                hs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[1][ai], fr[0][bi], hs[n], 0, 0, 0);
                hs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[1][ai], fr[1][bi], hs[n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const int ai = s3_ai(TY, n), bi = s3_bi(TY, n);
                hs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0][ai], fr[2][bi], hs[n], 0, 0, 0);
                hs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2][ai], fr[0][bi], hs[n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (nt > 0) {
        load_idx(0);
        gather(0);
        stage(0);
        if (nt > 1) {
            load_idx(1);
            gather(1);
        }
        if (nt > 2) load_idx(2);
        __syncthreads();
        auto run = [&](auto typ) __attribute__((always_inline)) {
            for (int tl = 0; tl < nt; ++tl) {
                tile(typ, tl & 1, tl + 1 < nt, tl + 2 < nt, tl + 2, tl + 3 < nt, tl + 3);
                __syncthreads();
            }
        };
        if (wty == 0) run(IntC<0>{});
        else if (wty == 1) run(IntC<1>{});
        else run(IntC<2>{});
    }
    // ---- H_i: this wave's blocks and their mirror images
    float *Hi = g.H + (int64_t)blockIdx.x * KP * KP;
    auto emit = [&](auto typ) __attribute__((always_inline)) {
        constexpr int TY = decltype(typ)::value;
#pragma unroll
        for (int n = 0; n < sym_np(TY); ++n) {
            const int ba = fblk[s3_ai(TY, n)], bb = fblk[s3_bi(TY, n)];
            float *blk = Hi + (32 * ba + 4 * lh) * KP + 32 * bb + l31;
            const int col = 32 * bb + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * ba + 4 * lh + (r & 3) + 8 * (r >> 2);
                float *dst = blk + ((r & 3) + 8 * (r >> 2)) * KP;
                float v = hs[n][r];
                if (g.accumulate & 1) v += *dst;
                else {
                    if (g.S) v += g.S[row * KP + col];
                    if (row == col && row < g.kvalid) v += g.diag;
                }
                hs[n][r] = v;
                *dst = v;
            }
            if (ba != bb) {
                float *tb = Hi + (32 * bb + l31) * KP + 32 * ba + 4 * lh;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4 *>(tb + 8 * q) = f32x4{hs[n][4 * q], hs[n][4 * q + 1], hs[n][4 * q + 2], hs[n][4 * q + 3]};
            }
        }
    };
    if (wty == 0) emit(IntC<0>{});
    else if (wty == 1) emit(IntC<1>{});
    else emit(IntC<2>{});
    // ---- gradient part: one partial per tile row, summed through LDS (the tiles are dead now)
    if (cls) return;
    __syncthreads();
    float *gr = reinterpret_cast<float *>(r6l);
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(gr + trow * KP + 4 * (16 * q + tl16)) = gacc[q];
    __syncthreads();
    if (t < KP) {
        float sacc = 0.f;
        for (int rep = 0; rep < 32; ++rep) sacc += gr[rep * KP + t];
        float *dst = g.G + i * KP + t;
        *dst = sacc + ((g.accumulate & 2) ? *dst : 0.f);
    }
}

} // namespace cmfk
