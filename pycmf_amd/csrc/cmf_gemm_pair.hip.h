// Two data-pass products of one MU half-iteration as ONE balanced launch (k_pad = 128).
//
// At mid sizes (C2: 16384 x 8192 X, 8192 x 4096 Y, k = 128) a data pass has 16-64 output tiles of 256 x 128 against 256 CUs, so
// gemm_kernel splits its reduction 4-16 ways -- and the two products of a half-iteration come out unequal: X^T U gives every
// workgroup 64 K-steps, Y Z 16 (a quarter of the flops behind the same fill, drain, launch and slab traffic: 0.73 of the MFMA
// peak against 0.82 for the long pass).  Here the K-steps of BOTH products (cmf_solvers.py:244: P = X^T U + Y Z is one sum;
// :232 / :238: X V and Y^T V share their right operand) are laid end to end, tile by tile, and cut into equal quotas, one per
// CU ("stream-K" over a pair of problems): workgroup w owns units [w Q, (w + 1) Q) of
//     product 0: tile 0 steps 0..S0-1, tile 1 ..., then product 1: tile 0 steps 0..S1-1, ...
// and runs one SEGMENT per (product, tile) its range touches -- the main loop of gemm_kernel<MODE, 128, 0, 4> over those
// K-steps, then the partial tile to slot (w - first workgroup of that tile) of the product's slab workspace.  The consumer
// (factor_update_kernel) sums the slots of a tile in slot order = ascending K: deterministic for a given device.
// Every workgroup does the same number of K-steps (C2: 80), one launch replaces two, and a tile has ceil(S / Q) + 1 slots at
// most instead of the 8 / 16 slabs of the split form (the V update reads 36 MB of partials instead of 64).
#pragma once

namespace cmfk {

struct PairProb {
    const float *A;       // NN: [rows_pad x K] row-major, pitch lda; TN: [K x >= rows] row-major, pitch lda
    const float *B;       // [K x 128] factor, pitch ldb
    float *C;             // slab workspace of this product: slot s at C + s * slab_stride, row-major pitch 128
    int64_t lda, ldb, slab_stride;
    int64_t unit0;        // first work unit of this product in the launch's unit space (product 0: 0)
    int mode;             // MODE_NN | MODE_TN
    int tiles;            // 256-row output tiles
    int ksteps;           // K / 32
};
struct PairArgs {
    PairProb p[2];
    int64_t quota, total; // units per workgroup, units in all
};

template <int MODE>
__device__ __forceinline__ void pair_segment(const PairProb &pb, const int tile, const int ks, const int len, const int slot, float *smem) {
    using C = GemmCfg<MODE, 128, 0>;
    constexpr int BN = 128;
    constexpr int F4K = C::BK / 4, F4M = C::BM / 4, F4R = BN / 4;
    static_assert(C::TM == 2 && C::TN == 2 && C::A_LD == 4 && C::B_LD == 2, "256 x 128 x 32 tile, 64 x 64 wave tiles");
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wid / C::WN, wn = wid % C::WN;
    const int wrow0 = wm * C::WTM, wcol0 = wn * C::WTN;
    const int64_t row0 = (int64_t)tile * C::BM;
    const int64_t kbeg = (int64_t)ks * C::BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ra[4], rb[2];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = t + 512 * p;
            if constexpr (C::A_KC) {
                const int r = idx >> ilog2(F4K), c4 = idx & (F4K - 1);
                ra[p] = *reinterpret_cast<const f32x4 *>(pb.A + (row0 + r) * pb.lda + k0 + 4 * c4);
            } else {
                const int r = idx >> ilog2(F4M), c4 = idx & (F4M - 1);
                ra[p] = *reinterpret_cast<const f32x4 *>(pb.A + (k0 + r) * pb.lda + row0 + 4 * c4);
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = t + 512 * p;
            const int r = idx >> ilog2(F4R), c4 = idx & (F4R - 1);
            rb[p] = *reinterpret_cast<const f32x4 *>(pb.B + (k0 + r) * pb.ldb + 4 * c4);
        }
    };
    auto lstore = [&](float *As, float *Bs) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = t + 512 * p;
            if constexpr (C::A_KC) {
                const int r = idx >> ilog2(F4K), c4 = idx & (F4K - 1);
                *reinterpret_cast<f32x4 *>(As + r * C::PADK + 4 * c4) = ra[p];
            } else {
                const int r = idx >> ilog2(F4M), c4 = idx & (F4M - 1);
                *reinterpret_cast<f32x4 *>(As + r * C::BM + 4 * c4) = ra[p];
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = t + 512 * p;
            const int r = idx >> ilog2(F4R), c4 = idx & (F4R - 1);
            *reinterpret_cast<f32x4 *>(Bs + r * BN + 4 * c4) = rb[p];
        }
    };
    // one K-step: staging schedule 4 of gemm_kernel (all LDS writes of tile t+1 in group 0, loads of tile t+2 in group 1),
    // fragment reads of group s+1 in front of the MFMAs of group s
    auto compute = [&](const float *As, const float *Bs, float *nAs, float *nBs, bool do_write, bool do_load, int64_t next_k0) {
        auto side = [&](int sidx) {
            if (sidx == 0) {
                if (do_write) lstore(nAs, nBs);
            } else if (sidx == 1) {
                if (do_load) gload(next_k0);
            }
        };
        if constexpr (C::A_KC) {
            f32x4 a[2][2];
            float b[2][2];
            auto lda_frag = [&](int q, f32x4 *dst) {
#pragma unroll
                for (int i = 0; i < 2; ++i) dst[i] = *reinterpret_cast<const f32x4 *>(As + (wrow0 + 32 * i + l31) * C::PADK + 4 * (2 * q + lh));
            };
            auto ldb_frag = [&](int sidx, float *dst) { // sidx = 4 q + e
                const int kk = 8 * (sidx >> 2) + 4 * lh + (sidx & 3);
                VecLoad<2>::ld(Bs + kk * BN + wcol0 + 2 * l31, dst);
            };
            lda_frag(0, a[0]);
            ldb_frag(0, b[0]);
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) {
                const int q = sidx >> 2, e = sidx & 3;
                if (sidx + 1 < 16) {
                    ldb_frag(sidx + 1, b[(sidx + 1) & 1]);
                    if (e == 0 && q + 1 < 4) lda_frag(q + 1, a[(q + 1) & 1]);
                }
                side(sidx);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][i][e], b[sidx & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // this lane's fragment bases as LDS pointers the optimiser cannot take apart (see gemm_kernel): one register + immediates
            typedef __attribute__((address_space(3))) const float lds_cf;
            typedef __attribute__((address_space(3))) const f32x2 lds_cf2;
            lds_cf *Ap = (lds_cf *)(As + lh * C::BM + wrow0 + 2 * l31);
            lds_cf *Bp = (lds_cf *)(Bs + lh * BN + wcol0 + 2 * l31);
            asm volatile("" : "+v"(Ap), "+v"(Bp));
            f32x2 a[2], b[2];
            a[0] = *reinterpret_cast<lds_cf2 *>(Ap);
            b[0] = *reinterpret_cast<lds_cf2 *>(Bp);
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) {
                if (sidx + 1 < 16) {
                    a[(sidx + 1) & 1] = *reinterpret_cast<lds_cf2 *>(Ap + 2 * (sidx + 1) * C::BM);
                    b[(sidx + 1) & 1] = *reinterpret_cast<lds_cf2 *>(Bp + 2 * (sidx + 1) * BN);
                }
                side(sidx);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sidx & 1][i], b[sidx & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    gload(kbeg);
    lstore(smem, smem + C::A_ELEMS);
    if (len > 1) gload(kbeg + C::BK); // tile 1 waits in registers
    __syncthreads();
    for (int kt = 0; kt < len; ++kt) {
        float *cur = smem + (kt & 1) * C::STAGE;
        float *nxt = smem + ((kt + 1) & 1) * C::STAGE;
        compute(cur, cur + C::A_ELEMS, nxt, nxt + C::A_ELEMS, kt + 1 < len, kt + 2 < len, kbeg + (int64_t)(kt + 2) * C::BK);
        __syncthreads(); // (the last one also frees both LDS stages for the workgroup's next segment)
    }

    // partial tile -> slot `slot` of this product's slabs.  Element (row, col): NN rows 32 i + rr, TN rows 2 rr + i (fragment maps)
    float *Cs = pb.C + (int64_t)slot * pb.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int64_t row = row0 + wrow0 + (C::A_KC ? (32 * i + rr) : (2 * rr + i));
            *reinterpret_cast<f32x2 *>(Cs + row * BN + wcol0 + 2 * l31) = f32x2{acc[i][0][r], acc[i][1][r]};
        }
}

__global__ __launch_bounds__(512, 2) void gemm_pair_kernel(PairArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int64_t u = (int64_t)blockIdx.x * g.quota;
    const int64_t end = (u + g.quota < g.total) ? u + g.quota : g.total;
    while (u < end) {
        const int pi = (u >= g.p[1].unit0) ? 1 : 0;
        const PairProb &pb = g.p[pi];
        const int64_t local = u - pb.unit0;
        const int tile = (int)(local / pb.ksteps), ks = (int)(local - (int64_t)tile * pb.ksteps);
        const int64_t left = end - u;
        const int len = (int)(left < pb.ksteps - ks ? left : pb.ksteps - ks);
        const int slot = (int)((int64_t)blockIdx.x - (pb.unit0 + (int64_t)tile * pb.ksteps) / g.quota);
        if (pb.mode == MODE_TN) pair_segment<MODE_TN>(pb, tile, ks, len, slot, smem);
        else pair_segment<MODE_NN>(pb, tile, ks, len, slot, smem);
        u += len;
    }
}

} // namespace cmfk
