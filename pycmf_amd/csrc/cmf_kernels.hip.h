// cmf_kernels.hip.h -- gfx950 (MI355X / CDNA4) device kernels of the CMF update engine.
//
// Everything heavy is phrased as one of three fp32-MFMA GEMM forms built on
// v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 64 FLOP/clk/SIMD):
//   NN : C[M x N]  = A[M x K]   * B[K x N]      X V, Y Z, F (V^T V), W KR(V)
//   TN : C[Kc x N] = A[K x Kc]^T * B[K x N]     X^T U, Y^T V, Grams, W^T KR(U)
//   NT : S[M x N]  = A[M x K]   * B[N x K]^T    U V^T / V Z^T with a fused
//        epilogue (residual f(S)-T, sigma'(S) weights, or sum of squares)
// One 512-thread workgroup (8 waves, 2 per SIMD) owns a 256 x BN output tile
// and walks the reduction in 32-deep steps through double-buffered LDS.
// Reference arithmetic being replaced: pycmf/cmf_solvers.py:232-245 (MU
// contractions), :399-400/:436-440 (Newton residual products), :36-42 (error).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { MODE_NN = 0, MODE_TN = 1, MODE_NT = 2 };

struct GemmArgs {
    const float *A;
    int64_t lda;
    const float *B;
    int64_t ldb;
    float *C;            // NN/TN: slab base; NT: unused
    int64_t ldc;
    int64_t slab_stride; // elements between split-K slabs
    int64_t Mout;        // valid output rows (TN: guards A columns / C rows)
    int64_t Kred;        // reduction length, multiple of 32
    int64_t klen;        // reduction length per split, multiple of 32
    // ---- NT epilogue ----
    const float *T;      // target (X or Y), row-major ldt, may be null (treated as 0)
    int64_t ldt;
    float *R;            // residual out  scale_r * (f(S) - T) * mask      (nullable)
    float *W;            // weight out    scale_w * w(S) * mask            (nullable)
    int64_t ldr;
    const uint8_t *mask; // 0/1 bytes, row-major ldm                       (nullable)
    int64_t ldm;
    double *sq_out;      // per-workgroup sum of squares of (T - f(S))     (nullable)
    int64_t Mvalid, Nvalid; // true (unpadded) extent for sq / logit masking
    float scale_r, scale_w;
    int link;            // 0 linear, 1 logit
    int w_is_slope;      // W = scale_w * sigma'(S) (1) or scale_w (0)
    // XCD-aware tile order of the NT passes (ras_rb > 0): see the remap at the top of gemm_kernel
    int ras_rb, ras_cb;
    unsigned long long *dbg; // diagnostic only: per-workgroup {memtime, memrealtime} at start and end (nullable)
    // ---- ROLE 2 (block-diagonal batch of k_pad = 256 products, NN form): row tile x multiplies its own B ----
    int64_t b_batch;     // elements between the B operands of consecutive row tiles
    const float *D;      // C = alpha * A B + beta * D + gamma * I  (D stacked like C, nullable)
    float alpha, beta, gamma;
    // ---- in-kernel split-K reduction (NN / TN): red_out != null ----
    // every split writes its partial tile to slab z of C (write-through stores) and takes a ticket of its output tile; the
    // workgroup that arrives last sums the slabs IN SLAB ORDER (deterministic) into red_out (+= when red_acc).
    // A single split writes (or adds) straight into red_out.
    float *red_out;
    unsigned *ticket;    // one zero-initialised word per output tile; reset by the last arriver
    int red_acc;
    // ---- fused factor update (NN, factor-side product with one N tile): epi != 0 ----
    //  EPI_MU     acc = F G is the MU denominator: F <- F * P / reg(acc)                    (cmf_solvers.py:212-228)
    //             epi_F = F (= A), epi_P = numerator, a = l1, b = l2, c = eps, written to epi_out (= F, in place)
    //  EPI_GRAD   acc = F G: grad = a acc - a P + b sign(F) + c F                            (:399-400, :436-440)
    //             epi_F = F (= A), epi_P = data term, a = scale, b = l1, c = l2, written to epi_out
    //  EPI_APPLY  acc = grad H^-1 is the Newton step: F <- clamp(F - acc), zero outside the valid block   (:321-326)
    //  EPI_DIRECT acc = (T O) H^-1 with the plain inverse: F <- clamp(a acc) -- the same point, F - (F H - a T O) H^-1 = a T O H^-1
    //             epi_F = F, written to epi_out (= F)
    //  EPI_COMBINE acc = F E, E = I - H Hinv (zero unless the safe inverse clamped): F <- clamp(acc + a P), P = T (O Hinv) --
    //             the re-associated Newton sweep (cmf_newton.hip.h, sweep_side_shared); A = F, written to epi_out (= F)
    int epi;
    const float *epi_F, *epi_P;
    float *epi_out;
    float epi_a, epi_b, epi_c;
    int64_t epi_rows;
    int epi_kvalid, epi_nn;
};
enum { EPI_NONE = 0, EPI_MU = 1, EPI_GRAD = 2, EPI_APPLY = 3, EPI_DIRECT = 4, EPI_COMBINE = 5 };

// write-through (sc1) stores: the bytes leave the XCD's L2 at once, so a workgroup on another XCD can read them after
// the storing wave's s_waitcnt vmcnt(0) and a ticket, with no release fence (MI355X_MICROARCH.md, publish-large)
__device__ __forceinline__ void store_wt(float *p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_wt(float *p, f32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
typedef float f32x1 __attribute__((ext_vector_type(1)));
__device__ __forceinline__ void store_wt(float *p, f32x1 v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v[0]) : "memory"); }
__device__ __forceinline__ void store_wt(float *p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// TILE 0: 256 x BN output tile, 32-deep K-step (everything).  TILE 1 (BN = 128 data passes, A/B option gemm_tile512): 512 x 128
// output tile, 16-deep K-step -- wave tile 128 x 64 like the 256 x 256 kernel's 64 x 128 (six fragment reads per eight MFMAs
// instead of four per four, half the B-tile fill per flop), the same MFMA work per barrier as the 256 x 128 x 32 step, and
// 98 KB of LDS double-buffered (a 32-deep step of this tile would need 180 KB).  TILE 2 (NT form, option nt_tile16): the 256 x 128
// output tile with a 16-deep K-step -- 61 KB of LDS instead of 110, so that TWO workgroups share a CU: the NT passes reduce over
// K = k_pad only (8 K-steps of 32 at k_pad = 256), a third of a workgroup's life is the fill in front of the first MFMA and the
// epilogue behind the last one (targets from HBM, link, squared residual), and with one workgroup per CU the matrix pipe idles
// through both; a second resident workgroup runs its K-steps meanwhile.
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

template <int MODE, int BN, int TILE = 0>
struct GemmCfg {
    static constexpr int BM = TILE == 1 ? 512 : 256, BK = TILE ? 16 : 32, PADK = BK + 4, NT = 512;
    static constexpr bool A_KC = (MODE != MODE_TN);
    static constexpr bool B_KC = (MODE == MODE_NT);
    static constexpr int A_ELEMS = A_KC ? BM * PADK : BK * BM;
    static constexpr int B_ELEMS = B_KC ? BN * PADK : BK * BN;
    static constexpr int STAGE = A_ELEMS + B_ELEMS;
    static constexpr int WN = (BN >= 64) ? 2 : 1;
    static constexpr int WM = 8 / WN;
    static constexpr int WTM = BM / WM, WTN = BN / WN;
    static constexpr int TM = WTM / 32, TN = WTN / 32;
    static constexpr int A_LD = BM * BK / 4 / NT;                   // float4 per thread (4; 2 for the 256 x 128 x 16 tile)
    static constexpr int B_F4 = B_KC ? BN * BK / 4 : BK * BN / 4;   // float4 in B tile
    static constexpr int B_LD = (B_F4 + NT - 1) / NT;
    static constexpr size_t LDS_BYTES = 2 * STAGE * sizeof(float);
};

// v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the IEEE division sequence (11 dependent VALU instructions)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// sigmoid and its slope WITHOUT cancellation: a = exp(-|x|) in (0, 1], s(|x|) = 1 / (1 + a), s(x) = x >= 0 ? s(|x|) : a s(|x|),
// s'(x) = s (1 - s) = a s(|x|)^2.  (The product s (1 - s) loses the slope altogether beyond |x| = 17 in float32: the Hessian weights of
// saturated rows came out as exact zeros where float64 has 1e-8 -- times |Z|^2 = 1e4 over 600 rows that is an eigenvalue above
// pert = 0.01: third campaign of profiles/fuzz_r06.md.)
__device__ __forceinline__ float sigmoid_slope_(float x, float &slope) {
    const float a = __expf(-fabsf(x));
    const float sp = __builtin_amdgcn_rcpf(1.0f + a);
    slope = a * sp * sp;
    return x >= 0.f ? sp : a * sp;
}

// Sum over aligned groups of GS lanes (8, 16, 32 or 64); every lane of a group receives the sum.
// Pure VALU: v_add_f32_dpp for the in-row steps, v_permlane16/32_swap for the row exchanges -- the
// __shfl_xor form goes through ds_bpermute (LDS crossbar, ~64-cycle dependent steps).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int GS>
__device__ __forceinline__ float group_sum(float v) {
    v += dpp_move<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v); // row_half_mirror: the other quad of the 8-lane half row
    if constexpr (GS >= 16) v += dpp_move<0x140>(v); // row_mirror: the other half of the 16-lane row
    // v_permlane16/32_swap exchange rows between TWO registers in place.  Written as inline asm: with equal
    // operand values hipcc (ROCm 7.2) folds the builtin's second result onto the first (emits v + v).  The two
    // v_nop cover the VALU-write -> permlane-read hazard, which hipcc does not pad inside an asm statement.
    if constexpr (GS >= 32) { // rows 2r and 2r+1
        float a = v, b = v;
        asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v = a + b;
    }
    if constexpr (GS >= 64) { // lower and upper 32 lanes
        float a = v, b = v;
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v = a + b;
    }
    return v;
}

template <int N>
struct VecLoad;
template <>
struct VecLoad<1> {
    static __device__ __forceinline__ void ld(const float *p, float *o) { o[0] = p[0]; }
};
template <>
struct VecLoad<2> {
    static __device__ __forceinline__ void ld(const float *p, float *o) {
        f32x2 v = *reinterpret_cast<const f32x2 *>(p);
        o[0] = v[0]; o[1] = v[1];
    }
};
template <>
struct VecLoad<4> {
    static __device__ __forceinline__ void ld(const float *p, float *o) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(p);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
};

// ROLE only separates the symbols: 0 = pass over a data-sized operand (X, Y, residual or
// weight image), 1 = factor-side product (Gram, F*G, step); profiles then report them apart.
// ROLE 2 (NN, BN = 256 only) is the batched 256 x 256 x 256 product of the Newton-Schulz spectral clamp: the
// matrices of a batch are stacked vertically, row tile x = matrix x reads ITS B operand (B + x * b_batch) and the
// epilogue forms alpha * A B + beta * D + gamma * I.
// PIPE selects the staging schedule inside a K-step (A/B-able in one process, cmf_set_option):
//   0: global loads of tile t+1 at group 0, all LDS writes as a burst after group 15
//   1: LDS writes of tile t+1 one piece per group in groups 0..7, loads of tile t+2 at group 8
//   2: LDS writes two pieces per group in groups 0..3, loads of tile t+2 at group 4
//   3: LDS writes four pieces per group in groups 0..1, loads of tile t+2 at group 2
//   4: all eight LDS writes in group 0, loads of tile t+2 at group 1
//   5: schedule 4 with waves 4-7 staging half a K-step later than waves 0-3 (stagger)
//  10: direct-to-LDS loads of tile t+1 at group 0 (no staging registers, no ds_write): TN +1 %, NN -5 %
// Diagnostic builds (not kept): no staging at all 153-154 TF/s (98 %), LDS writes only 146-147, no barrier +0.5 %:
// the 7 % between schedule 4 and the bare MFMA+fragment loop is the 64 KB/step of LDS fill traffic and the
// global loads themselves, not where in the step they are issued.
// (also tried and dropped: piece p written in group p and re-loaded at once: -1..-3 %; one piece in the shadow
//  of each MFMA: -25 %, hipcc's conservative waitcnts serialise it; wave specialisation, 12 waves with one
//  loader wave per SIMD doing all the staging and the 8 MFMA waves only reading fragments: 131/121 TF/s NN/TN
//  with direct-to-LDS loads, 131/136 with register staging, against 142.5/141.7 for schedule 4 - a third wave
//  per SIMD costs more than the staging instructions it takes off the MFMA waves)
template <int MODE, int BN, int ROLE = 0, int PIPE = 0, int TILE = 0>
__global__ __launch_bounds__(512, TILE == 2 ? 4 : 2) void gemm_kernel(GemmArgs g) {
    using C = GemmCfg<MODE, BN, TILE>;
    static_assert(TILE == 0 || (TILE == 1 && BN == 128 && MODE != MODE_NT && ROLE == 0 && PIPE == 4) || (TILE == 2 && BN == 128 && MODE == MODE_NT),
                  "the 512 x 128 x 16 tile: k_pad = 128 data passes only; the 256 x 128 x 16 tile: NT passes only");
    // (shifts and masks, not / and %: with the signed division hipcc stopped folding the four A-tile LDS addresses of a thread into
    // one base + immediate offsets, and that alone cost the TN pass 2.7 % and 0.5 GB of extra operand fetch per launch at C4 --
    // profiles/HISTORY.md, round 4, A/B of the library builds)
    constexpr int F4K = C::BK / 4;   // 16-byte chunks per row of a k-contiguous tile
    constexpr int F4M = C::BM / 4;   // ... per row of the [k][rows] tile of the TN form
    constexpr int NGRP = C::BK / 2;  // MFMA groups (k pairs) per K-step
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wid = t >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wid / C::WN, wn = wid % C::WN;
    const int wrow0 = wm * C::WTM, wcol0 = wn * C::WTN;

    // NT passes reduce over K = k_pad only: a 256 x 128 tile reads 384 KB of factor rows for 16.8 MFLOP, and in launch order (row
    // tile fastest, workgroups dealt round-robin over the XCDs) the 32 workgroups an XCD runs at a time touch 32 DIFFERENT row tiles
    // (8 MB against 4 MB of L2): every operand byte comes from beyond L2, 2.5 TB/s at C4.  With ras_rb > 0 the workgroups that share
    // an XCD (linear id mod 8, speed only -- nothing depends on the placement) walk their own eighth of the column tiles in blocks
    // of ras_rb row tiles x ras_cb column tiles: 1 MB + 1 MB of operands per 32 tiles, re-read from the XCD's L2.
    unsigned tile_x = blockIdx.x, tile_y = blockIdx.y;
    if constexpr (MODE == MODE_NT) {
        if (g.ras_rb > 0) {
            const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
            const unsigned xcd = lin & 7u, q = lin >> 3;
            const unsigned per_x = gridDim.y >> 3, blk = (unsigned)(g.ras_rb * g.ras_cb);
            const unsigned b = q / blk, w = q % blk;
            const unsigned nrb = gridDim.x / (unsigned)g.ras_rb;
            tile_x = (b % nrb) * (unsigned)g.ras_rb + w % (unsigned)g.ras_rb;
            tile_y = xcd * per_x + (b / nrb) * (unsigned)g.ras_cb + w / (unsigned)g.ras_rb;
        }
    }
    const int64_t row0 = (int64_t)tile_x * C::BM; // output-row tile origin
    const int64_t n0 = (int64_t)tile_y * BN;      // output-col tile origin
    const int64_t kbeg = (int64_t)blockIdx.z * g.klen;
    int64_t kend = kbeg + g.klen;
    if (kend > g.Kred) kend = g.Kred;
    const int nkt = (int)((kend - kbeg) / C::BK);

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int64_t wg_linear = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (g.dbg && t == 0) { // clock diagnostic (tools/probe_clock.py); never set in production launches
        g.dbg[4 * wg_linear + 0] = __builtin_amdgcn_s_memtime();
        g.dbg[4 * wg_linear + 1] = __builtin_amdgcn_s_memrealtime();
    }

    const float *Bbase = (ROLE == 2) ? g.B + (int64_t)blockIdx.x * g.b_batch : g.B;
    f32x4 ra[C::A_LD], rb[C::B_LD];
    // Row of a k-contiguous tile that staging slot `q` (= thread index / chunks per row) carries.  16-deep tiles (TILE 2: 80-byte
    // rows, four 16-byte chunks each): the sixteen lanes a ds_write_b128 serves together would write rows q .. q + 3, whose 64-byte
    // pieces overlap mod 256 bytes (80 n mod 256 = 0, 80, 160, 240: 1.6e9 bank-conflict cycles at C4, PMC); rows q, q + 4, q + 8,
    // q + 12 sit at 0, 64, 128, 192 mod 256.  So the slots of each group of 16 rows are dealt out transposed (4 x 4).
    auto krow = [](int q) -> int {
        if constexpr (TILE == 2) return (q & ~15) | ((q & 3) << 2) | ((q >> 2) & 3);
        else return q;
    };

    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int p = 0; p < C::A_LD; ++p) {
            const int idx = t + C::NT * p;
            if constexpr (C::A_KC) {
                const int r = krow(idx >> ilog2(F4K)), c4 = idx & (F4K - 1);
                ra[p] = *reinterpret_cast<const f32x4 *>(g.A + (row0 + r) * g.lda + k0 + 4 * c4);
            } else {
                // branch-free: a conditional load makes hipcc wait for the loads it has just
                // issued (it must assume the skipped path); out-of-range columns are clamped
                // here and zeroed when the tile is written to LDS
                const int r = idx >> ilog2(F4M), c4 = idx & (F4M - 1);
                int64_t col = row0 + 4 * c4;
                if (ROLE == 1 && col > g.Mout - 4) col = g.Mout - 4;
                ra[p] = *reinterpret_cast<const f32x4 *>(g.A + (k0 + r) * g.lda + col);
            }
        }
#pragma unroll
        for (int p = 0; p < C::B_LD; ++p) {
            const int idx = t + C::NT * p;
            if (C::B_F4 >= C::NT || idx < C::B_F4) {
                if constexpr (C::B_KC) {
                    const int r = krow(idx >> ilog2(F4K)), c4 = idx & (F4K - 1);
                    rb[p] = *reinterpret_cast<const f32x4 *>(Bbase + (n0 + r) * g.ldb + k0 + 4 * c4);
                } else {
                    constexpr int F4R = BN / 4;
                    const int r = idx / F4R, c4 = idx % F4R;
                    rb[p] = *reinterpret_cast<const f32x4 *>(Bbase + (k0 + r) * g.ldb + n0 + 4 * c4);
                }
            }
        }
    };
    // one 16-byte piece of the staged tile -> LDS (p < A_LD: A tile, else B tile)
    auto lstore_piece = [&](float *As, float *Bs, int p) {
        if (p < C::A_LD) {
            const int idx = t + C::NT * p;
            if constexpr (C::A_KC) {
                const int r = krow(idx >> ilog2(F4K)), c4 = idx & (F4K - 1);
                *reinterpret_cast<f32x4 *>(As + r * C::PADK + 4 * c4) = ra[p];
            } else {
                const int r = idx >> ilog2(F4M), c4 = idx & (F4M - 1);
                f32x4 v = ra[p];
                if (ROLE == 1 && row0 + 4 * c4 >= g.Mout) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(As + r * C::BM + 4 * c4) = v;
            }
        } else if (p - C::A_LD < C::B_LD) {
            const int pb = p - C::A_LD;
            const int idx = t + C::NT * pb;
            if (C::B_F4 >= C::NT || idx < C::B_F4) {
                if constexpr (C::B_KC) {
                    const int r = krow(idx >> ilog2(F4K)), c4 = idx & (F4K - 1);
                    *reinterpret_cast<f32x4 *>(Bs + r * C::PADK + 4 * c4) = rb[pb];
                } else {
                    constexpr int F4R = BN / 4;
                    const int r = idx / F4R, c4 = idx % F4R;
                    *reinterpret_cast<f32x4 *>(Bs + r * BN + 4 * c4) = rb[pb];
                }
            }
        }
    };
    auto lstore = [&](float *As, float *Bs) {
#pragma unroll
        for (int p = 0; p < C::A_LD + C::B_LD; ++p) lstore_piece(As, Bs, p);
    };

    // One K-step = 16 MFMA groups (one per k-pair).  Software pipeline inside the step:
    //  * fragment reads of group s+1 are issued before the MFMAs of group s (register double
    //    buffer), so LDS latency sits under 8-16 MFMAs of 64 cycles;
    //  * the staged registers of tile t+1 (loaded during step t-1) are written to the other LDS
    //    buffer one 16-byte piece per group in groups 0..7, i.e. under MFMAs instead of as a
    //    burst in front of the barrier;
    //  * the global loads of tile t+2 are issued at group 8 into the registers just drained.
    // sched_barrier(0) pins that order (hipcc otherwise re-serialises reads and MFMAs).
    const bool late = __builtin_amdgcn_readfirstlane(wid) >= 4; // wave-uniform by construction
    // PIPE 10: direct-to-LDS loads (global_load_lds_dwordx4): no staging registers, no ds_write.  The LDS
    // image must then be lane-linear, so the k-contiguous A tile of the NN form loses its row padding and is
    // XOR-swizzled instead (16-byte chunk c of row r lives at chunk c ^ (r & 7)); the permutation is applied to
    // the per-lane SOURCE address and again on the fragment read.
    constexpr bool GLDS = (PIPE == 10);
    static_assert(!GLDS || TILE == 0, "direct-to-LDS staging is laid out for the 256-row tile");
    constexpr int APK = GLDS ? C::BK : C::PADK; // floats per row of a k-contiguous A tile
    auto glds_tile = [&](float *As, float *Bs, int64_t k0) {
        typedef __attribute__((address_space(3))) float lds_f;
#pragma unroll
        for (int p = 0; p < C::A_LD; ++p) {
            const int idx = t + C::NT * p;
            const int wave_first = __builtin_amdgcn_readfirstlane(idx & ~63);
            const float *src;
            if constexpr (C::A_KC) {
                const int r = idx >> 3, cpos = idx & 7;
                src = g.A + (row0 + r) * g.lda + k0 + 4 * (cpos ^ (r & 7));
            } else {
                const int r = idx >> 6, c4 = idx & 63;
                int64_t col = row0 + 4 * c4;
                if (ROLE == 1 && col > g.Mout - 4) col = g.Mout - 4;
                src = g.A + (k0 + r) * g.lda + col;
            }
            __builtin_amdgcn_global_load_lds(src, (lds_f *)(As + 4 * wave_first), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < C::B_LD; ++p) {
            const int idx = t + C::NT * p;
            if (C::B_F4 >= C::NT || idx < C::B_F4) {
                const int wave_first = __builtin_amdgcn_readfirstlane(idx & ~63);
                constexpr int F4R = BN / 4;
                const int r = idx / F4R, c4 = idx % F4R;
                const float *src = Bbase + (k0 + r) * g.ldb + n0 + 4 * c4;
                __builtin_amdgcn_global_load_lds(src, (lds_f *)(Bs + 4 * wave_first), 16, 0, 0);
            }
        }
    };
    auto compute = [&](const float *As, const float *Bs, float *nAs, float *nBs, bool do_write, bool do_load,
                       int64_t next_k0) {
        auto side = [&](int sidx) {
            if constexpr (PIPE == 0) {
                if (sidx == 0 && do_load) gload(next_k0);
            } else if constexpr (PIPE == 10) {
                if (sidx == 0 && do_write) glds_tile(nAs, nBs, next_k0); // next_k0 = tile t+1 here
            } else if constexpr (PIPE == 5) {
                // schedule 4 with the two waves of every SIMD staggered by half a K-step: waves 0-3 stage in
                // groups 0/1, waves 4-7 in groups 8/9, so one wave's staging sits beside its partner's MFMAs
                const int g0 = late ? 8 : 0;
                if (sidx == g0) {
                    if (do_write) lstore(nAs, nBs);
                } else if (sidx == g0 + 1) {
                    if (do_load) gload(next_k0);
                }
            } else {
                constexpr int WPG = 1 << (PIPE - 1); // LDS-write pieces per group: 1, 2 or 4
                constexpr int NG = 8 / WPG;          // groups that carry writes; loads go at group NG
                if (sidx < NG) {
                    if (do_write) {
#pragma unroll
                        for (int w = 0; w < WPG; ++w) lstore_piece(nAs, nBs, WPG * sidx + w);
                    }
                } else if (sidx == NG) {
                    if (do_load) gload(next_k0);
                }
            }
        };
        if constexpr (C::A_KC) {
            f32x4 a[2][C::TM];
            auto lda_frag = [&](int q, f32x4 *dst) {
#pragma unroll
                for (int i = 0; i < C::TM; ++i) {
                    const int row = wrow0 + 32 * i + l31;
                    const int chunk = GLDS ? ((2 * q + lh) ^ (row & 7)) : (2 * q + lh);
                    dst[i] = *reinterpret_cast<const f32x4 *>(As + row * APK + 4 * chunk);
                }
            };
            if constexpr (C::B_KC) {
                f32x4 b[2][C::TN];
                auto ldb_frag = [&](int q, f32x4 *dst) {
#pragma unroll
                    for (int j = 0; j < C::TN; ++j)
                        dst[j] = *reinterpret_cast<const f32x4 *>(Bs + (wcol0 + 32 * j + l31) * C::PADK + 4 * (2 * q + lh));
                };
                lda_frag(0, a[0]);
                ldb_frag(0, b[0]);
#pragma unroll
                for (int sidx = 0; sidx < NGRP; ++sidx) {
                    const int q = sidx >> 2, e = sidx & 3;
                    if (e == 0 && q + 1 < NGRP / 4) {
                        lda_frag(q + 1, a[(q + 1) & 1]);
                        ldb_frag(q + 1, b[(q + 1) & 1]);
                    }
                    side(sidx);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < C::TM; ++i)
#pragma unroll
                        for (int j = 0; j < C::TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][i][e], b[q & 1][j][e], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                float b[2][C::TN];
                auto ldb_frag = [&](int sidx, float *dst) { // sidx = 4*q + e
                    const int kk = 8 * (sidx >> 2) + 4 * lh + (sidx & 3);
                    VecLoad<C::TN>::ld(Bs + kk * BN + wcol0 + C::TN * l31, dst);
                };
                lda_frag(0, a[0]);
                ldb_frag(0, b[0]);
#pragma unroll
                for (int sidx = 0; sidx < NGRP; ++sidx) {
                    const int q = sidx >> 2, e = sidx & 3;
                    if (sidx + 1 < NGRP) {
                        ldb_frag(sidx + 1, b[(sidx + 1) & 1]);
                        if (e == 0 && q + 1 < NGRP / 4) lda_frag(q + 1, a[(q + 1) & 1]);
                    }
                    side(sidx);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < C::TM; ++i)
#pragma unroll
                        for (int j = 0; j < C::TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q & 1][i][e], b[sidx & 1][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            float a[2][C::TM], b[2][C::TN];
            // k_pad = 128 tile: this lane's fragment bases of the K-step, as LDS pointers the optimiser cannot take apart.  Left
            // alone, hipcc hoists (lane term + 2 sidx * row pitch) out of the K loop as 27 loop-invariant registers and rebuilds
            // every fragment address with four vector adds per MFMA group; from a pinned base the 16 groups are one register
            // plus immediate offsets, as in the 256-wide build (which folds them by itself and is left exactly as it was).
            typedef __attribute__((address_space(3))) const float lds_cf;
            constexpr bool PIN = (BN == 128 && TILE == 0);
            lds_cf *Ap = (lds_cf *)(As + lh * C::BM + wrow0 + C::TM * l31);
            lds_cf *Bp = (lds_cf *)(Bs + lh * BN + wcol0 + C::TN * l31);
            if constexpr (PIN) asm volatile("" : "+v"(Ap), "+v"(Bp));
            auto ld_frag = [&](int sidx, float *da, float *db) {
                if constexpr (PIN) {
                    static_assert(!PIN || (C::TM == 2 && C::TN == 2), "pinned fragment reads: 64 x 64 wave tile");
                    typedef __attribute__((address_space(3))) const f32x2 lds_cf2;
                    const f32x2 va = *reinterpret_cast<lds_cf2 *>(Ap + 2 * sidx * C::BM);
                    const f32x2 vb = *reinterpret_cast<lds_cf2 *>(Bp + 2 * sidx * BN);
                    da[0] = va[0]; da[1] = va[1];
                    db[0] = vb[0]; db[1] = vb[1];
                    return;
                }
                const int kk = 2 * sidx + lh;
                VecLoad<C::TM>::ld(As + kk * C::BM + wrow0 + C::TM * l31, da);
                VecLoad<C::TN>::ld(Bs + kk * BN + wcol0 + C::TN * l31, db);
            };
            ld_frag(0, a[0], b[0]);
#pragma unroll
            for (int sidx = 0; sidx < NGRP; ++sidx) {
                if (sidx + 1 < NGRP) ld_frag(sidx + 1, a[(sidx + 1) & 1], b[(sidx + 1) & 1]);
                side(sidx);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < C::TM; ++i)
#pragma unroll
                    for (int j = 0; j < C::TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sidx & 1][i], b[sidx & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if constexpr (GLDS) {
        static_assert(MODE != MODE_NT, "direct-to-LDS staging is wired for the NN / TN forms");
        if (nkt > 0) {
            glds_tile(smem, smem + C::A_ELEMS, kbeg);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int kt = 0; kt < nkt; ++kt) {
                float *cur = smem + (kt & 1) * C::STAGE;
                float *nxt = smem + ((kt + 1) & 1) * C::STAGE;
                compute(cur, cur + C::A_ELEMS, nxt, nxt + C::A_ELEMS, kt + 1 < nkt, false, kbeg + (int64_t)(kt + 1) * C::BK);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    } else if (nkt > 0) {
        gload(kbeg);
        lstore(smem, smem + C::A_ELEMS);
        if (PIPE != 0 && nkt > 1) gload(kbeg + C::BK); // tile 1 waits in registers
        // (asking for tiles 0 and 1 together through a second set of staging registers: no change at C2's 16-64 K-steps per
        //  workgroup, 0.172 ms either way in the interleaved A/B)
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            float *cur = smem + (kt & 1) * C::STAGE;
            float *nxt = smem + ((kt + 1) & 1) * C::STAGE;
            if constexpr (PIPE == 0) {
                const bool more = kt + 1 < nkt;
                compute(cur, cur + C::A_ELEMS, nxt, nxt + C::A_ELEMS, false, more, kbeg + (int64_t)(kt + 1) * C::BK);
                if (more) lstore(nxt, nxt + C::A_ELEMS);
            } else {
                compute(cur, cur + C::A_ELEMS, nxt, nxt + C::A_ELEMS, kt + 1 < nkt, kt + 2 < nkt,
                        kbeg + (int64_t)(kt + 2) * C::BK);
            }
            __syncthreads();
        }
    }

    if (g.dbg && t == 0) {
        g.dbg[4 * wg_linear + 2] = __builtin_amdgcn_s_memtime();
        g.dbg[4 * wg_linear + 3] = __builtin_amdgcn_s_memrealtime();
    }
    // ---------------------------------------------------------------- epilogue
    if constexpr (MODE != MODE_NT) {
        typedef float vecN __attribute__((ext_vector_type(C::TN)));
        const int nsplit = (int)gridDim.z;
        const bool in_red = g.red_out != nullptr;
        if (ROLE == 1 && MODE == MODE_NN && g.epi != EPI_NONE) {
            // fused factor update.  F and the second operand of EIGHT rows are fetched before any arithmetic: one latency per
            // batch instead of one per row
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    vecN f[8], pv[8];
                    int64_t off[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = r0 + q;
                        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        off[q] = (row0 + wrow0 + 32 * i + rr) * g.ldc + n0 + wcol0 + C::TN * l31;
                        if (g.epi != EPI_DIRECT && g.epi != EPI_COMBINE) f[q] = *reinterpret_cast<const vecN *>(g.epi_F + off[q]);
                        else
#pragma unroll
                            for (int j = 0; j < C::TN; ++j) f[q][j] = 0.f;
                        if (g.epi != EPI_APPLY && g.epi != EPI_DIRECT) pv[q] = *reinterpret_cast<const vecN *>(g.epi_P + off[q]);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = r0 + q;
                        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int64_t row = row0 + wrow0 + 32 * i + rr;
                        vecN o;
#pragma unroll
                        for (int j = 0; j < C::TN; ++j) {
                            const float av = acc[i][j][r], fv = f[q][j];
                            float res;
                            if (g.epi == EPI_MU) {
                                float d = av;
                                if (g.epi_a > 0.f) d += g.epi_a;
                                if (g.epi_b > 0.f) d = d + g.epi_b * fv;
                                if (d == 0.f) d = g.epi_c;
                                res = fv * (pv[q][j] / d);
                            } else if (g.epi == EPI_GRAD) {
                                const float sg = (fv > 0.f) ? 1.f : ((fv < 0.f) ? -1.f : 0.f);
                                float gval = g.epi_a * av;
                                gval += -g.epi_a * pv[q][j];
                                res = gval + g.epi_b * sg + g.epi_c * fv;
                            } else {
                                res = 0.f;
                                if (row < g.epi_rows && (int)(n0 + wcol0 + C::TN * l31 + j) < g.epi_kvalid) {
                                    res = g.epi == EPI_DIRECT ? g.epi_a * av : (g.epi == EPI_COMBINE ? av + g.epi_a * pv[q][j] : fv - av);
                                    if (g.epi_nn && res < 0.f) res = 0.f;
                                }
                            }
                            o[j] = res;
                        }
                        *reinterpret_cast<vecN *>(g.epi_out + off[q]) = o;
                    }
                }
        } else {
            float *Cs = (in_red && nsplit == 1) ? g.red_out : g.C + (int64_t)blockIdx.z * g.slab_stride;
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int64_t row = row0 + wrow0 + (C::A_KC ? (32 * i + rr) : (C::TM * rr + i));
                    if (row < g.Mout) {
                        const int64_t off = row * g.ldc + n0 + wcol0 + C::TN * l31;
                        float *dst = Cs + off;
                        if constexpr (ROLE == 2) {
                            static_assert(ROLE != 2 || (MODE == MODE_NN && BN == 256), "batched form: NN, 256 x 256 blocks");
                            const int rin = wrow0 + 32 * i + rr, c0 = wcol0 + 4 * l31; // coordinates inside the 256 x 256 block
                            f32x4 v = {g.alpha * acc[i][0][r], g.alpha * acc[i][1][r], g.alpha * acc[i][2][r], g.alpha * acc[i][3][r]};
                            if (g.D) v += g.beta * *reinterpret_cast<const f32x4 *>(g.D + row * g.ldc + c0);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (rin == c0 + j) v[j] += g.gamma;
                            *reinterpret_cast<f32x4 *>(dst) = v;
                        } else {
                            vecN v;
#pragma unroll
                            for (int j = 0; j < C::TN; ++j) v[j] = acc[i][j][r];
                            if (in_red && nsplit == 1) {
                                if (g.red_acc) v += *reinterpret_cast<const vecN *>(dst);
                                *reinterpret_cast<vecN *>(dst) = v;
                            } else if (in_red) {
                                store_wt(dst, v);
                            } else {
                                *reinterpret_cast<vecN *>(dst) = v;
                            }
                        }
                    }
                }
        }
        if (ROLE != 2 && in_red && nsplit > 1) {
            // ticket: the last of the nsplit workgroups of this output tile reduces (Guideline 16: every storing wave drains
            // its write-through stores, workgroup barrier, ONE agent-scope atomic add; the last arriver acquires once)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned *tk = g.ticket + (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
            int *flag = reinterpret_cast<int *>(smem);
            if (t == 0) {
                const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = (old == (unsigned)(nsplit - 1)) ? 1 : 0;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
                }
                *flag = last;
            }
            __syncthreads();
            if (*flag) {
                // 16 independent row segments in flight per thread and slab (a dependent chain of single loads made
                // this reduction latency-bound: 100 us for 1 MB)
#pragma unroll
                for (int i = 0; i < C::TM; ++i) {
                    vecN sum[16];
                    int64_t offs[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        int64_t row = row0 + wrow0 + (C::A_KC ? (32 * i + rr) : (C::TM * rr + i));
                        if (row >= g.Mout) row = g.Mout - 1; // clamped: loaded, never stored
                        offs[r] = row * g.ldc + n0 + wcol0 + C::TN * l31;
                        if (g.red_acc) sum[r] = *reinterpret_cast<const vecN *>(g.red_out + offs[r]);
                        else {
#pragma unroll
                            for (int j = 0; j < C::TN; ++j) sum[r][j] = 0.f;
                        }
                    }
                    for (int sidx = 0; sidx < nsplit; ++sidx) {
                        const float *slab = g.C + (int64_t)sidx * g.slab_stride;
                        vecN part[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) part[r] = *reinterpret_cast<const vecN *>(slab + offs[r]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum[r] += part[r];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int64_t row = row0 + wrow0 + (C::A_KC ? (32 * i + rr) : (C::TM * rr + i));
                        if (row < g.Mout) *reinterpret_cast<vecN *>(g.red_out + offs[r]) = sum[r];
                    }
                }
            }
        }
    } else {
        float sq = 0.0f;
        // per-lane origin (row of register 0, this lane's column); every element is then a
        // wave-uniform offset away, so the address arithmetic stays on the scalar unit
        const int64_t rbase = row0 + wrow0 + 4 * lh;
        const int64_t cbase = n0 + wcol0 + l31;
        const float *Tp = g.T ? g.T + rbase * g.ldt + cbase : nullptr;
        float *Rp = g.R ? g.R + rbase * g.ldr + cbase : nullptr;
        float *Wp = g.W ? g.W + rbase * g.ldr + cbase : nullptr;
        const uint8_t *Mp = g.mask ? g.mask + rbase * g.ldm + cbase : nullptr;
        const int ldt = (int)g.ldt, ldr = (int)g.ldr, ldm = (int)g.ldm;
        const int rlim = (int)min((int64_t)1 << 20, g.Mvalid - rbase); // valid iff rr < rlim
        const int clim = (int)min((int64_t)1 << 20, g.Nvalid - cbase);
        if (!Mp) {
            // no byte mask (everything but the masked-dense Newton formulation): fetch ALL targets of the wave tile
            // first -- one latency for 64 loads in flight instead of 64 dependent load -> wait -> use rounds, which
            // made this epilogue longer than the 8 K-steps in front of it -- then the arithmetic
            // (TILE 2 -- two workgroups per CU, 128 registers per lane: one block row of the wave tile at a time, 32 loads in flight)
            constexpr int IB = (TILE == 2 || BN == 256) ? 1 : C::TM;   // block rows per batch (256-wide tile: 64 of its 128 targets per lane at a time)
            const float slope = g.w_is_slope ? 1.0f : 0.0f, nslope = 1.0f - slope;
#pragma unroll
            for (int i0 = 0; i0 < C::TM; i0 += IB) {
                float tv[IB][16][C::TN];
#pragma unroll
                for (int i = 0; i < IB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int j = 0; j < C::TN; ++j) tv[i][r][j] = 0.0f;
                if (Tp) {
#pragma unroll
                    for (int i = 0; i < IB; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
#pragma unroll
                            for (int j = 0; j < C::TN; ++j)   // read once: non-temporal, so that the targets do not push the operands out of L2
                                tv[i][r][j] = __builtin_nontemporal_load(Tp + (32 * (i0 + i) + (r & 3) + 8 * (r >> 2)) * ldt + 32 * j);
                }
#pragma unroll
                for (int i = 0; i < IB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = 32 * (i0 + i) + (r & 3) + 8 * (r >> 2);
                        // the link is uniform over the launch: a scalar branch around the sigmoids of this register row (the blend
                        // lkf * sigmoid(s) + nlk * s paid v_exp + v_rcp, quarter-rate instructions, for every element of a LINEAR side)
                        float fv[C::TN];
#pragma unroll
                        for (int j = 0; j < C::TN; ++j) fv[j] = acc[i0 + i][j][r];
                        float sl[C::TN]; // the link's slope at fv (cancellation-free for the sigmoid: sigmoid_slope_)
                        if (__builtin_amdgcn_readfirstlane(g.link)) {
#pragma unroll
                            for (int j = 0; j < C::TN; ++j) fv[j] = sigmoid_slope_(fv[j], sl[j]);
                            asm volatile("" ::: "memory"); // keep the branch: do not speculate the transcendentals into a select
                        } else {
#pragma unroll
                            for (int j = 0; j < C::TN; ++j) sl[j] = fv[j] * (1.0f - fv[j]);
                        }
#pragma unroll
                        for (int j = 0; j < C::TN; ++j) {
                            const float f = fv[j];
                            const float mk = (rr < rlim && 32 * j < clim) ? 1.0f : 0.0f;
                            const float res = f - tv[i][r][j];
                            sq += mk * res * res;
                            if (Rp) Rp[rr * ldr + 32 * j] = g.scale_r * mk * res;
                            if (Wp) Wp[rr * ldr + 32 * j] = g.scale_w * mk * (slope * sl[j] + nslope);
                        }
                    }
            }
        } else
#pragma unroll
        for (int i = 0; i < C::TM; ++i) {
            // byte mask (masked-dense Newton formulation, gradient of the shared-partial-sum sides): all targets and mask
            // bytes of this block row of the wave tile are fetched before the arithmetic -- one latency, not 64
            float tv[16][C::TN];
            uint8_t mv[16][C::TN];
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int j = 0; j < C::TN; ++j) {
                    const int rr = 32 * i + (r & 3) + 8 * (r >> 2);
                    tv[r][j] = Tp ? Tp[rr * ldt + 32 * j] : 0.0f;
                    mv[r][j] = Mp[rr * ldm + 32 * j];
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = 32 * i + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int j = 0; j < C::TN; ++j) {
                    const float s = acc[i][j][r];
                    float slope = 1.0f;
                    const float f = g.link ? sigmoid_slope_(s, slope) : s;
                    float mk = (rr < rlim && 32 * j < clim) ? 1.0f : 0.0f;
                    mk *= (float)mv[r][j];
                    const float res = f - tv[r][j];
                    if (g.sq_out) sq += mk * res * res;
                    if (Rp) Rp[rr * ldr + 32 * j] = g.scale_r * mk * res;
                    if (Wp) {
                        const float w = g.w_is_slope ? (g.link ? slope : f * (1.0f - f)) : 1.0f;
                        Wp[rr * ldr + 32 * j] = g.scale_w * mk * w;
                    }
                }
            }
        }
        if (g.sq_out) {
            double v = (double)sq;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            __syncthreads();
            double *red = reinterpret_cast<double *>(smem);
            if (lane == 0) red[wid] = v;
            __syncthreads();
            if (t == 0) {
                double tot = 0.0;
                for (int w = 0; w < 8; ++w) tot += red[w];
                g.sq_out[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
            }
        }
    }
}

// ------------------------------------------------------------------ factor update with a small row tile (k_pad 64 / 128)
// The factor-side products of an update step are rows x k_pad x k_pad: at mid sizes (C2: U has 16384 rows) the 256-row tile
// of gemm_kernel gives 64 workgroups that each spend their time waiting for four dependent K-steps -- 29 us for 1.7 us of
// MFMA work.  Here one 256-thread workgroup owns 64 rows: the whole right operand (k_pad x k_pad) and its 64 x k_pad left
// tile are fetched in ONE round, then 2 x k_pad / 2 MFMAs per wave run back to back, then the update is applied as in
// gemm_kernel's fused epilogues (EPI_MU / EPI_GRAD / EPI_APPLY).  The second operand P of the epilogue may arrive as the
// UNREDUCED split-K slabs of the data pass that produced it (P = sum of nslab slabs, summed here in slab order -- the same
// values sum_slabs_kernel would have written, without its launch and its round trip through HBM).
// slots of a 256-row output tile in a paired launch (cmf_gemm_pair.hip.h): the workgroups whose unit ranges meet
// [unit0 + tile S, unit0 + (tile + 1) S)
__host__ __device__ __forceinline__ int pair_slots(int64_t quota, int64_t unit0, int ksteps, int64_t tile) {
    const int64_t first = (unit0 + tile * ksteps) / quota, last = (unit0 + (tile + 1) * ksteps - 1) / quota;
    return (int)(last - first + 1);
}

struct FactorUpdArgs {
    const float *A;      // left operand rows x KP (F for EPI_MU / EPI_GRAD, the gradient for EPI_APPLY), pitch KP
    const float *B;      // KP x KP right operand (Gram or inverse Hessian), pitch KP
    int epi;             // EPI_MU | EPI_GRAD | EPI_APPLY
    const float *F;      // factor (epilogue operand), pitch KP
    const float *P;      // second epilogue operand (nullable when it arrives as slabs only)
    const float *S1, *S2; // ... plus the unreduced split-K slabs of up to two data passes: P_total = P + sum S1 + sum S2
    int n1, n2;
    int64_t stride1, stride2;
    int64_t q1, q2, u1, u2; // slabs of a paired launch (q > 0: quota, first unit): n = pair_slots(q, u, ks, row tile), not a constant
    int ks1, ks2;
    float *out;
    float a, b, c;
    int64_t rows_valid;
    int kvalid, nn;
};

template <int KP>
__device__ __forceinline__ void factor_update_body(const FactorUpdArgs &g, const int64_t blk) {
    static_assert(KP == 64 || KP == 128, "small-tile factor update: k_pad 64 or 128");
    constexpr int LA = KP + 4;             // A tile pitch: 16-byte slot (KP / 4 + 1) * row: ds_read_b128 of 16 rows is conflict-free
    constexpr int NB = KP / 64;            // 32-column blocks per wave (wave tile 32 x KP / 2)
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float *Bs = fsm;                       // [KP][KP]
    float *As = fsm + KP * KP;             // [64][LA]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int64_t row0 = blk * 64;
    const int wr = (w & 1) * 32, wc = (w >> 1) * (KP / 2);
    {   // every global load of the workgroup is issued before the first LDS store: ONE memory latency for both operands
        constexpr int NBL = KP * KP / 4 / 256, NAL = 64 * KP / 4 / 256;
        f32x4 gb[NBL], ab[NAL];
#pragma unroll
        for (int q = 0; q < NBL; ++q) gb[q] = reinterpret_cast<const f32x4 *>(g.B)[t + 256 * q];
#pragma unroll
        for (int q = 0; q < NAL; ++q) {
            const int idx = t + 256 * q;
            ab[q] = *reinterpret_cast<const f32x4 *>(g.A + (row0 + idx / (KP / 4)) * KP + 4 * (idx % (KP / 4)));
        }
#pragma unroll
        for (int q = 0; q < NBL; ++q) reinterpret_cast<f32x4 *>(Bs)[t + 256 * q] = gb[q];
#pragma unroll
        for (int q = 0; q < NAL; ++q) {
            const int idx = t + 256 * q;
            *reinterpret_cast<f32x4 *>(As + (idx / (KP / 4)) * LA + 4 * (idx % (KP / 4))) = ab[q];
        }
    }
    // the epilogue's own operand F is requested before the MFMA loop (its latency hides under the product), not after it
    float fpre[NB][16];
    {
        const bool needf = (g.epi != EPI_DIRECT && g.epi != EPI_COMBINE);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float *Fb = g.F + (row0 + wr + 4 * lh) * KP + wc + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) fpre[j][r] = needf ? Fb[((r & 3) + 8 * (r >> 2)) * KP] : 0.f;
        }
    }
    __syncthreads();
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 4
    for (int q = 0; q < KP / 4; ++q) { // four k per chunk: two MFMA steps
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(As + (wr + l31) * LA + 4 * q);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = 4 * q + 2 * e + lh;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2 * e + lh], Bs[k * KP + wc + 32 * j + l31], acc[j], 0, 0, 0);
        }
    }
    // epilogue: element (row, col) = (row0 + wr + (r & 3) + 8 (r >> 2) + 4 lh, wc + 32 j + l31): 32 lanes = 128 contiguous bytes
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int col = wc + 32 * j + l31;
        float fv[16], pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) fv[r] = fpre[j][r];
#pragma unroll
        for (int r = 0; r < 16; ++r) pv[r] = 0.f;
        if (g.epi != EPI_APPLY && g.epi != EPI_DIRECT) {
            const int64_t eoff = (row0 + wr + 4 * lh) * KP + col;
            auto add_slabs = [&](const float *base, int ns, int64_t stride) {
                const float *Pb = base + eoff;
                int sidx = 0;
                for (; sidx + 4 <= ns; sidx += 4) { // 64 independent loads in flight, summed in slab order
                    float part[4][16];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) part[u][r] = Pb[(int64_t)(sidx + u) * stride + ((r & 3) + 8 * (r >> 2)) * KP];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) pv[r] += part[u][r];
                }
                for (; sidx < ns; ++sidx) {
                    float part[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[r] = Pb[(int64_t)sidx * stride + ((r & 3) + 8 * (r >> 2)) * KP];
#pragma unroll
                    for (int r = 0; r < 16; ++r) pv[r] += part[r];
                }
            };
            if (g.P) add_slabs(g.P, 1, 0);
            if (g.n1 > 0) add_slabs(g.S1, g.q1 > 0 ? pair_slots(g.q1, g.u1, g.ks1, row0 >> 8) : g.n1, g.stride1);
            if (g.n2 > 0) add_slabs(g.S2, g.q2 > 0 ? pair_slots(g.q2, g.u2, g.ks2, row0 >> 8) : g.n2, g.stride2);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float av = acc[j][r], f = fv[r];
            float res;
            if (g.epi == EPI_MU) {
                float d = av;
                if (g.a > 0.f) d += g.a;
                if (g.b > 0.f) d = d + g.b * f;
                if (d == 0.f) d = g.c;
                res = f * (pv[r] / d);
            } else if (g.epi == EPI_GRAD) {
                const float sg = (f > 0.f) ? 1.f : ((f < 0.f) ? -1.f : 0.f);
                float gval = g.a * av;
                gval += -g.a * pv[r];
                res = gval + g.b * sg + g.c * f;
            } else {
                res = 0.f;
                if (row < g.rows_valid && col < g.kvalid) {
                    res = g.epi == EPI_DIRECT ? g.a * av : (g.epi == EPI_COMBINE ? av + g.a * pv[r] : f - av);
                    if (g.nn && res < 0.f) res = 0.f;
                }
            }
            g.out[row * KP + col] = res;
        }
    }
}
template <int KP>
__global__ __launch_bounds__(256) void factor_update_kernel(FactorUpdArgs g) { factor_update_body<KP>(g, blockIdx.x); }
// two independent updates in one grid (U and Z behind the paired launch of X V and Y^T V: Z's 64 workgroups alone leave three
// quarters of the chip idle for a whole latency-bound launch): workgroups [0, nblk0) run g0, the rest g1
template <int KP>
__global__ __launch_bounds__(256) void factor_update2_kernel(FactorUpdArgs g0, FactorUpdArgs g1, int nblk0) {
    if ((int)blockIdx.x < nblk0) factor_update_body<KP>(g0, blockIdx.x);
    else factor_update_body<KP>(g1, (int64_t)blockIdx.x - nblk0);
}

// ------------------------------------------------------------------ small Grams (k_pad 64 / 128)
// G = F^T F for a factor of a few thousand rows (MU at C2: V^T V over 8192 rows, U^T U + Z^T Z over the stacked 20480) costs
// next to nothing in flops but was 12 % of the C2 iteration as a one-tile TN GEMM split 160 ways plus the sum of its 10 MB of
// slabs.  Here every workgroup forms ONE upper 64 x 64 tile over its share of the rows (four waves, one 32 x 32 MFMA block each,
// both operands from the same 32-row LDS image: pitch 96 floats puts the two 32-lane halves of a fragment read 32 banks apart)
// and there are at most 64 row shares, so the partial sums are 64 slabs of the upper tiles (0.75 MB at k_pad = 128), added up in
// share order by gram32_reduce_kernel (deterministic).  F: rows_pad x kp, zero padding rows.
__global__ __launch_bounds__(256) void gram32_partial_kernel(const float *F, int kp, int64_t rows_pad, int64_t chunk, float *slab) {
    constexpr int TS = 64, LD = TS + 32;
    __shared__ __attribute__((aligned(16))) float sA[32 * LD];
    __shared__ __attribute__((aligned(16))) float sB[32 * LD];
    const int T = kp / TS;
    int b = blockIdx.x, ti = 0;
    while (b >= T - ti) { b -= T - ti; ++ti; }
    const int tj = ti + b;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wi = w >> 1, wj = w & 1, l31 = lane & 31, lh = lane >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * chunk;
    const int64_t r1 = r0 + chunk < rows_pad ? r0 + chunk : rows_pad;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // 32 rows x 16 float4 per operand = two float4 per thread and operand; the next 32 rows wait in registers under the MFMAs
    f32x4 na[2], nb[2];
    auto fetch = [&](int64_t r) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = t + 256 * q, row = idx / (TS / 4), c4 = idx % (TS / 4);
            const float *src = F + (r + row) * kp + 4 * c4;
            na[q] = *reinterpret_cast<const f32x4 *>(src + ti * TS);
            nb[q] = *reinterpret_cast<const f32x4 *>(src + tj * TS);
        }
    };
    if (r0 < r1) fetch(r0);
    for (int64_t r = r0; r < r1; r += 32) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = t + 256 * q, row = idx / (TS / 4), c4 = idx % (TS / 4);
            *reinterpret_cast<f32x4 *>(sA + row * LD + 4 * c4) = na[q];
            *reinterpret_cast<f32x4 *>(sB + row * LD + 4 * c4) = nb[q];
        }
        __syncthreads();
        if (r + 32 < r1) fetch(r + 32);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a = sA[(2 * s + lh) * LD + wi * 32 + l31];
            const float bb = sB[(2 * s + lh) * LD + wj * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
        }
    }
    float *out = slab + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (TS * TS);
#pragma unroll
    for (int r = 0; r < 16; ++r)
        out[(wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TS + wj * 32 + l31] = acc[r];
}
// G[r][c] (kp x kp, symmetric) = sum over the shares, in share order
__global__ __launch_bounds__(256) void gram32_reduce_kernel(const float *slab, int kp, int nsplit, float *G) {
    constexpr int ts = 64;
    const int T = kp / ts, ntile = T * (T + 1) / 2;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < kp * kp; idx += gridDim.x * 256) {
        const int r = idx / kp, c = idx % kp;
        const int rr = r < c ? r : c, cc = r < c ? c : r; // upper-triangle representative (tile row <= tile column)
        const int ti = rr / ts, tj = cc / ts;
        int ir = rr % ts, ic = cc % ts;
        if (ti == tj) { ir = r % ts; ic = c % ts; }       // diagonal tiles are stored whole
        const int tile = ti * T - ti * (ti - 1) / 2 + (tj - ti);
        const float *p = slab + (int64_t)tile * ts * ts + ir * ts + ic;
        const int64_t st = (int64_t)ntile * ts * ts;
        float s = 0.f;
        int q = 0;
        for (; q + 8 <= nsplit; q += 8) { // eight loads in flight, added in share order (a chain of single dependent loads was 12 us for 1.5 MB)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(q + u) * st];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; q < nsplit; ++q) s += p[q * st];
        G[idx] = s;
    }
}

// ------------------------------------------------------------------ elementwise
// dst[i] = (accumulate ? dst[i] : 0) + sum_s src[s*stride + i]      (float4 lanes)
__global__ void sum_slabs_kernel(float *dst, const float *src, int64_t n4, int nslab,
                                 int64_t stride, int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 a = accumulate ? reinterpret_cast<f32x4 *>(dst)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nslab; ++s) a += reinterpret_cast<const f32x4 *>(src + (int64_t)s * stride)[i];
        reinterpret_cast<f32x4 *>(dst)[i] = a;
    }
}

// F *= num / reg(den)   -- MUSolver._regularized_delta, cmf_solvers.py:212-228
__global__ void mu_apply_kernel(float *F, const float *num, const float *den, int64_t n4,
                                float l1, float l2, float eps) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 f = reinterpret_cast<f32x4 *>(F)[i];
        const f32x4 nu = reinterpret_cast<const f32x4 *>(num)[i];
        f32x4 de = reinterpret_cast<const f32x4 *>(den)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d = de[e];
            if (l1 > 0.f) d += l1;
            if (l2 > 0.f) d = d + l2 * f[e];
            if (d == 0.f) d = eps;
            f[e] = f[e] * (nu[e] / d);
        }
        reinterpret_cast<f32x4 *>(F)[i] = f;
    }
}

// Newton gradient assembly + shared-inverse step pieces
// grad = a*P + b*Q + l1*sign(F) + l2*F          (cmf_solvers.py:400, :438)
__global__ void newton_grad_kernel(float *grad, const float *P, float a, const float *Q, float b,
                                   const float *F, float l1, float l2, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float f = F[i];
        float gval = a * P[i];
        if (Q) gval += b * Q[i];
        const float sg = (f > 0.f) ? 1.f : ((f < 0.f) ? -1.f : 0.f);
        grad[i] = gval + l1 * sg + l2 * f;
    }
}

// F <- F - step ; clamp (cmf_solvers.py:321-326); rows >= rows_valid stay 0
__global__ void newton_apply_kernel(float *F, const float *step, int64_t rows_valid, int kp, int kvalid,
                                    int64_t n, int non_negative) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / kp;
        const int c = (int)(i % kp);
        float v = 0.f;
        if (r < rows_valid && c < kvalid) {
            v = F[i] - step[i];
            if (non_negative && v < 0.f) v = 0.f;
        }
        F[i] = v;
    }
}

// re-associated Newton sweep (cmf_newton.hip.h): F <- clamp(acc + a P) on the valid block, 0 outside; acc (= F E) nullable
__global__ void combine_clamp_kernel(float *F, const float *acc, const float *P, float a, int64_t rows_valid, int kp, int kvalid,
                                     int64_t n4, int non_negative) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = (4 * i) / kp;
        const int c0 = (int)((4 * i) % kp);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < rows_valid) {
            const f32x4 pv = reinterpret_cast<const f32x4 *>(P)[i];
            v = a * pv;
            if (acc) v += reinterpret_cast<const f32x4 *>(acc)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (c0 + e >= kvalid) v[e] = 0.f;
                else if (non_negative && v[e] < 0.f) v[e] = 0.f;
            }
        }
        reinterpret_cast<f32x4 *>(F)[i] = v;
    }
}
// out = sign(F)  (the l1 term of the gradient, cmf_solvers.py:400, as a GEMM operand)
__global__ void sign_kernel(float *out, const float *F, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 f = reinterpret_cast<const f32x4 *>(F)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f[e] > 0.f) ? 1.f : ((f[e] < 0.f) ? -1.f : 0.f);
        reinterpret_cast<f32x4 *>(out)[i] = o;
    }
}

__global__ void axpby_diag_kernel(float *H, const float *A, float a, const float *B, float b, float diag,
                                  int kp, int kvalid) {
    // H = a*A + b*B + diag*I on the valid k x k block; identity on the padding (grid-stride: any k_pad, any grid)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)kp * kp; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / kp), c = (int)(i % kp);
        float v = a * A[i] + (B ? b * B[i] : 0.f);
        if (r == c) v += diag;
        if (r >= kvalid || c >= kvalid) v = (r == c) ? 1.0f : 0.0f;
        H[i] = v;
    }
}

__global__ void sum_doubles_kernel(const double *in, int64_t n, double *out) {
    __shared__ double red[256];
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) v += in[i];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

__global__ void sumsq_kernel(const float *A, int64_t n4, double *partials) {
    double v = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4 *>(A)[i];
        v += (double)a[0] * a[0] + (double)a[1] * a[1] + (double)a[2] * a[2] + (double)a[3] * a[3];
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

// partial sums of <A, B> (float32 operands, float64 products and sums); n4 = elements / 4
__global__ void dot_kernel(const float *A, const float *B, int64_t n4, double *partials) {
    double v = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = reinterpret_cast<const f32x4 *>(A)[i], b = reinterpret_cast<const f32x4 *>(B)[i];
        v += (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2] + (double)a[3] * b[3];
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
// <A, B> of two n-element float64 arrays (single block)
__global__ void frob_inner64_kernel(const double *A, const double *B, int n, double *out) {
    __shared__ double red[256];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += A[i] * B[i];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

// ------------------------------------------------------------------ synthetic data
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// |N(0,1)| at global coordinate (gi, gj): depends on (seed, gi, gj) only
__device__ __forceinline__ float absnormal_at(uint64_t seed, uint64_t gi, uint64_t gj) {
    const uint64_t h = mix64(mix64(seed ^ (gi * 0xD1342543DE82EF95ull)) ^ gj);
    const float u1 = ((float)((h >> 40) + 1)) * (1.0f / 16777217.0f);          // (0,1]
    const float u2 = ((float)((h >> 8) & 0xFFFFFF)) * (1.0f / 16777216.0f);    // [0,1)
    const float r = sqrtf(-2.0f * __logf(u1));
    return fabsf(r * __cosf(6.28318530718f * u2));
}
// kind 0: scale |N(0,1)|   1: sigmoid(N(0,1)) (targets of a logit side, benchmarks/benchmark_cmf.py:78)   2: Bernoulli(scale) in {0, 1}
__global__ void fill_absnormal_kernel(float *A, int64_t ld, int64_t rows, int64_t cols, uint64_t seed,
                                      int64_t row0, int64_t col0, float scale, int kind) {
    const int64_t c4n = (cols + 3) / 4;
    const int64_t total = rows * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4n, c = (i % c4n) * 4;
        for (int e = 0; e < 4 && c + e < cols; ++e) {
            float v;
            if (kind == 0) v = scale * absnormal_at(seed, (uint64_t)(row0 + r), (uint64_t)(col0 + c + e));
            else {
                const uint64_t h = mix64(mix64(seed ^ ((uint64_t)(row0 + r) * 0xD1342543DE82EF95ull)) ^ (uint64_t)(col0 + c + e));
                const float u1 = ((float)((h >> 40) + 1)) * (1.0f / 16777217.0f);
                const float u2 = ((float)((h >> 8) & 0xFFFFFF)) * (1.0f / 16777216.0f);
                if (kind == 1) v = 1.0f / (1.0f + __expf(-sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2)));
                else v = (u2 < scale) ? 1.0f : 0.0f;
            }
            A[r * ld + c + e] = v;
        }
    }
}

// ---- Newton-Schulz spectral clamp (flagged per-row Hessians, k_pad = 256) -------------------------------------
// indices of the flagged matrices of a chunk (order irrelevant) and their count
__global__ __launch_bounds__(256) void compact_flags_kernel(const int *flags, int n, int *idx, int *count) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (flags[i]) idx[atomicAdd(count, 1)] = i;
}

// Record of the spectral clamp in float32 (cmf_newton_clamp_stats): for every matrix the clamp acts on (flags[b] != 0, or all when
// flags is null) count it and keep the largest ||H_b||_F / pert.  The clamp max(|lambda|, pert) of a float32 Hessian resolves
// eigenvalues only to about eps32 * ||H||, i.e. to a RELATIVE error of eps32 * ||H|| / pert in the clamped directions of the
// inverse: the ratio says when that leaves the stated tolerance.  One wave per matrix, H is not modified.
// step / Frows / tol (round 6; all or none): the float32 result of matrix b and the factor row it updates.  A matrix above its
// ratio threshold is LISTED only if the float32 error can matter: eps32 * ratio * ||step_b|| > tol * ||F_b|| -- eps32 * ratio bounds
// the relative error of the clamped (or ill-conditioned) directions of the inverse, so the left side bounds the error of this
// row's update and the right side is the tolerance on the factor row.  One that passes is certified by the bound: counted as
// clamped, not entered in the maximum ratio "left in float32".
__global__ __launch_bounds__(64) void clamp_stats_kernel(const float *H, const int *flags, int n, int kp, int64_t stride, float pert,
                                                         unsigned long long *count, unsigned *maxratio, float thr, int mode, int *bad,
                                                         const float *condest, float thr_plain, const float *step = nullptr,
                                                         const float *Frows = nullptr, float tol = 0.f, int clamp_sens = 0) {
    // mode 0: record every clamped matrix.  mode 1 (float64 refinement on): matrices with ratio > thr are LISTED in bad[1 ..]
    // (bad[0] = how many; they will be redone in float64) and only the others recorded.  mode 2: record only those above thr
    // (a chunk whose list the host declined to refine).
    // condest (nullable): for the matrices the clamp does NOT act on (flags[b] == 0: solved by plain Cholesky), the solve
    // kernel's condition estimate max H_ii / min L_ii^2 -- the same eps32 * ratio argument applies to a plain float32 solve;
    // they enter the list / the maximum by that ratio but not the count of clamped matrices.
    const int b = blockIdx.x;
    const bool clamped = !flags || flags[b];
    if (!clamped && !condest) return;
    float fro = 0.f;
    if (clamped) {
        const float *src = H + (int64_t)b * stride;
        for (int r = 0; r < n; ++r)
            for (int q = threadIdx.x; q < n; q += 64) {
                const float v = src[r * kp + q];
                fro += v * v;
            }
        for (int off = 32; off > 0; off >>= 1) fro += __shfl_xor(fro, off, 64);
    }
    float s2 = 0.f, f2 = 0.f;
    if (step && Frows && tol > 0.f) {
        for (int q = threadIdx.x; q < n; q += 64) {
            const float sv = step[(int64_t)b * kp + q], fv = Frows[(int64_t)b * kp + q];
            s2 += sv * sv;
            f2 += fv * fv;
        }
        for (int off = 32; off > 0; off >>= 1) { s2 += __shfl_xor(s2, off, 64); f2 += __shfl_xor(f2, off, 64); }
    }
    if (threadIdx.x == 0) {
        float ratio = clamped ? sqrtf(fro) / pert : condest[b];
        if (!(ratio == ratio)) ratio = 3.0e38f;
        ratio = fminf(ratio, 3.0e38f);
        if (!clamped && mode != 2) atomicMax(maxratio + 1, __float_as_uint(ratio)); // largest estimate over ALL plain solves (diagnostic)
        // an overflowed Hessian (inf / NaN: a diverged iteration) is recorded, never listed: float64 cannot repair it
        bool above = ratio > (clamped ? thr : thr_plain);
        // clamp_sens: for the matrices the clamp acted on, condest[b] holds the eigen-solve's own sensitivity estimate (relative error
        // of its step: cmf_eigclamp.hip.h) -- sharper than eps32 * ratio, which prices every clamped matrix as if an eigenvalue sat
        // right at the threshold
        const float relerr = (clamped && clamp_sens && condest) ? condest[b] : 1.1920929e-7f * ratio;
        if (above && step && Frows && tol > 0.f && ratio < 1.0e30f && !(relerr * sqrtf(s2) > tol * sqrtf(f2))) {
            if (clamped) atomicAdd(count, 1ull); // certified by the error bound: float32 is enough for this row
            return;
        }
        if (mode == 1 && above && ratio < 1.0e30f) {
            bad[1 + atomicAdd(bad, 1)] = b;
            return;
        }
        if (mode == 2 && !above) return;
        if (!clamped && !above) return; // a well-conditioned plain solve: nothing to record
        atomicMax(maxratio, __float_as_uint(ratio));
        if (clamped) atomicAdd(count, 1ull);
    }
}


// The matrices listed for the float64 refinement (list[0 .. nb)) leave the float32 clamp path: their flag is cleared -- the spectral
// clamp by Newton-Schulz products (49 float32 256^3 products per matrix) would be thrown away when refine_rows64 overwrites the step --
// and their step row is zeroed, so that a row the refinement could not finish (a failed factorisation: not reached for positive
// semi-definite Hessians) stays where it was instead of moving by a half-finished solve.
__global__ __launch_bounds__(64) void unflag_listed_kernel(int *flags, const int *list, int nb, float *step, int kp) {
    const int q = blockIdx.x;
    if (q >= nb) return;
    const int b = list[q];
    if (!flags[b]) return; // a plain solve listed by its condition estimate: its float32 step stands until the refinement replaces it
    for (int c = threadIdx.x; c < kp; c += 64) step[(int64_t)b * kp + c] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) flags[b] = 0;
}

// B = H - pert I for flagged matrix b (valid n x n block; padding: c on the diagonal, 0 elsewhere),
// c = min(||B||_F, ||B||_inf) >= rho(B), X0 = B / c.  One workgroup per matrix; thread t owns column t
// (H is symmetric, so column sums are row sums and the reads are coalesced).  cmax collects max c (float bits).
// sub = 256 / k_pad matrices share one 256 x 256 block-diagonal image (matrix b = diagonal block b % sub of image
// b / sub, zeros off the diagonal blocks): products of block-diagonal matrices stay block-diagonal, so the 256^3
// batched kernel serves k_pad = 128 too.  Workgroups b >= nf fill the unused diagonal blocks with the identity.
__global__ __launch_bounds__(256) void ns_prepare_kernel(const float *H, const int *idx, float *Bm, float *X, int n, int kp,
                                                         int64_t stride, float pert, unsigned *cmax, int sub, int nf) {
    __shared__ float red_f[4], red_m[4];
    const int b = blockIdx.x, t = threadIdx.x;
    const int ld = sub * kp;                       // = 256
    const int64_t img = (int64_t)(b / sub) * ld * ld + (int64_t)(b % sub) * kp * ld; // first row of this matrix in its image
    const int c0 = (b % sub) * kp;                 // first column of its diagonal block
    float *Bb = Bm + img, *Xb = X + img;
    if (b >= nf) { // filler block: identity (sign +1), never read back
        if (t < ld)
            for (int r = 0; r < kp; ++r) {
                const float v = (t == c0 + r) ? 1.0f : 0.f;
                Bb[r * ld + t] = v;
                Xb[r * ld + t] = v;
            }
        return;
    }
    const float *src = H + (int64_t)idx[b] * stride;
    float fro = 0.f, colsum = 0.f;
    if (t < n)
        for (int r = 0; r < n; ++r) {
            const float v = src[r * kp + t] - (r == t ? pert : 0.f);
            fro += v * v;
            colsum += fabsf(v);
        }
    for (int off = 32; off > 0; off >>= 1) {
        fro += __shfl_xor(fro, off, 64);
        colsum = fmaxf(colsum, __shfl_xor(colsum, off, 64));
    }
    if ((t & 63) == 0) { red_f[t >> 6] = fro; red_m[t >> 6] = colsum; }
    __syncthreads();
    fro = red_f[0] + red_f[1] + red_f[2] + red_f[3];
    colsum = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    float c = fminf(sqrtf(fro), colsum);
    if (!(c > 1e-30f)) c = 1.0f;
    const float ci = 1.0f / c;
    if (t == 0) atomicMax(cmax, __float_as_uint(c));
    if (t < ld)
        for (int r = 0; r < kp; ++r) {
            const int cc = t - c0; // column inside the diagonal block (outside: zero)
            float v = 0.f;
            if (cc >= 0 && cc < kp) {
                v = (r == cc) ? c : 0.f;
                if (r < n && cc < n) v = src[r * kp + cc] - (r == cc ? pert : 0.f);
            }
            Bb[r * ld + t] = v;
            Xb[r * ld + t] = v * ci;
        }
}

} // namespace cmfk
