// cmf_rowhess.hip.h -- fused per-row Newton accumulation (gfx950).
//
// For one row f_i of a factor and the (sampled) rows o_j, j in S_i, of the other factor it forms in
// ONE pass over the gathered rows (pycmf/cmf_solvers.py:414-428, :455-484, :494-506):
//     z_j = f_i . o_j            r_j = s (link(z_j) - t_ij)           w_j = s link'(z_j)   (1 if linear)
//     g_i = sum_j r_j o_j        H_i = sum_j w_j o_j o_j^T
// One 512-thread workgroup per row.  The 32 gathered rows of a K-step are staged through registers
// into an LDS tile; while they are still in registers the owning lanes take the dot product with
// f_i (each thread holds k_pad/64 16-byte chunks of ONE gathered row, the at most 16 lanes that share
// the row finish the sum with four DPP adds), so z, r, w and the gradient cost no LDS traffic and one
// sigmoid / target load per thread and K-step.  The rows are written to LDS twice, raw (B operand) and scaled by
// w_j (A operand) -- scaling the A fragments in registers from a 32-float weight array instead measured 12 % slower
// (436 vs 390 ms at C3) -- so that H_i is a rank-32 MFMA update per step with exactly the TN GEMM's inner loop -- only the sampled rows are ever touched,
// there is no residual / weight / mask image and no Khatri-Rao matrix.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfk {

struct RowHessArgs {
    const float *O;       // other factor, rows x KP
    const float *F;       // this factor, rows x KP (row i = f_i)
    const int32_t *idx;   // [nrows x s] sample lists, or null: j = 0..s-1
    int64_t idx_stride;   // elements between the lists of consecutive rows (= s)
    int s;                // samples per row
    const float *T;       // targets: t_ij = T[i*t_row + j*t_col]
    int64_t t_row, t_col;
    float scale;
    int link;             // 0 linear, 1 logit
    float *H;             // [nrows x KP x KP] out (accumulate ? += : =)
    float *G;             // [nrows x KP] gradient part out (accumulate ? += : =)
    int accumulate;       // bit 0: H += , bit 1: G +=
    const float *S;       // shared (symmetric) k_pad x k_pad part added to every H_i by the launch that does not accumulate, or null
    float diag;           // ... together with diag * I on the first kvalid diagonal entries
    int kvalid;
    int64_t row0;         // first row of this launch (blockIdx.x + row0 = i)
    int64_t nrows;        // rows in this launch; H, G are indexed by blockIdx.x (H) / i (G)
    // CLASS mode (cls_cnt != null; linear link, see row_classes below): "row" b of the launch is a partial sum shared by
    // several factor rows -- its list starts at idx + cls_off[b] and holds cls_cnt[b] samples; only H is produced
    const int64_t *cls_off;
    const int32_t *cls_cnt;
    int cls_upper;        // class launch of the k_pad = 256 symmetric kernel: store the 36 upper blocks only ...
    int cls_nc1;          // ... of image blockIdx.x = group * cls_nc1 + class, block-major and class-minor (see CLS_BLOCK)
    // SPLIT mode (nsplit > 1; few rows with long sample lists, e.g. the Z sweep of a 64-column Y over 1e5 rows of V): workgroup
    // b serves chunk b / nrows of row b % nrows -- samples [chunk * split_len, ...) of its list -- and writes PARTIAL sums:
    // H to slot b (so the partials of chunk c form slab c of nrows images), G to G + chunk * g_split_stride.  The caller launches
    // with accumulate = 0, S = null, diag = 0 and adds the slabs up afterwards (deterministic, in chunk order).
    int nsplit, split_len;
    int64_t g_split_stride;
};

template <int KP>
struct RowHessCfg {
    static constexpr int LPR = KP / 4 < 16 ? KP / 4 : 16;  // lanes per gathered row
    static constexpr int CPT = (KP / 4) / LPR;             // 16-byte chunks of that row per lane (chunk q*LPR + lane)
    static constexpr int WM = KP >= 128 ? 4 : (KP == 64 ? 2 : 1);
    static constexpr int WN = KP >= 64 ? 2 : 1;
    static constexpr int TM = KP == 256 ? 2 : 1;
    static constexpr int TN = KP == 256 ? 4 : (KP == 128 ? 2 : 1);
    static constexpr int TILE = 32 * KP;
    static constexpr size_t LDS_BYTES = (4 * TILE) * sizeof(float); // 2 stages x (raw rows, rows scaled by w_j)
};

// SYM (k_pad = 256 only): H_i is symmetric, so only the 36 blocks (32 x 32) on or above the block diagonal are
// accumulated -- 9 MFMAs per k-pair and SIMD instead of 16 -- and every off-diagonal block is stored twice.
// Block columns are contiguous here (block b = columns 32b..32b+31).  Wave -> blocks, balanced 4 + 5 per SIMD:
//   waves 0-3 (type 0): (w, 4) (w, 5) (w, 6) (w, 7)
//   waves 4, 6 (type 1, base 0 / 4): (b, b) (b, b+1) (b, b+2) (b, b+3) (b+3, b+3)
//   waves 5, 7 (type 2, base 0 / 4): (b+1, b+1) (b+1, b+2) (b+1, b+3) (b+2, b+2) (b+2, b+3)
__host__ __device__ constexpr int sym_np(int ty) { return ty == 0 ? 4 : 5; }   // blocks of the wave
__host__ __device__ constexpr int sym_na(int ty) { return ty == 0 ? 1 : 2; }   // distinct A fragments
__host__ __device__ constexpr int sym_nb(int ty) { return ty == 2 ? 3 : 4; }   // distinct B fragments
__host__ __device__ constexpr int sym_ai(int ty, int n) { return ty == 0 ? 0 : (ty == 1 ? (n == 4 ? 1 : 0) : (n >= 3 ? 1 : 0)); }
__host__ __device__ constexpr int sym_bi(int ty, int n) { return ty == 0 ? n : (ty == 1 ? (n == 4 ? 3 : n) : (n < 3 ? n : n - 2)); }
// SYM = 3 (weights >= 0): ONE staged image sqrt(w_j) o_j feeds both MFMA operands, H_i = sum_j (sqrt(w_j) o_j)(sqrt(w_j) o_j)^T:
// half the LDS fill of the raw + weighted pair, and a wave reads each block fragment once (5 / 4 / 3 per k-pair for the
// three wave types instead of 5 / 6 / 5).  Measured at C3: 385.6 ms against 388.2 ms -- the LDS fill was not the cost.  Fragment f of a wave: type 0 -> block row w, then block columns 4..7;
// types 1, 2 -> its block columns.
__host__ __device__ constexpr int s3_nf(int ty) { return ty == 0 ? 5 : (ty == 1 ? 4 : 3); }
__host__ __device__ constexpr int s3_ai(int ty, int n) { return ty == 0 ? 0 : (ty == 1 ? (n == 4 ? 3 : 0) : (n >= 3 ? 1 : 0)); }
__host__ __device__ constexpr int s3_bi(int ty, int n) { return ty == 0 ? n + 1 : (ty == 1 ? (n == 4 ? 3 : n) : (n < 3 ? n : n - 2)); }
// SYM = 4 (the default): SYM = 3 with the 8 DIAGONAL blocks accumulated as three 16 x 16 sub-blocks each -- (0,0), (0,1), (1,1) by
// v_mfma_f32_16x16x4f32 over four samples: 3 x 8 passes against the 2 x 16 of two 32x32x2 steps, the lower sub-block is the
// mirror image of (0,1).  A SIMD then runs 34 block-equivalents per k-pair and workgroup instead of 36 (the symmetric half is
// 32.1): waves 0-3 keep their four off-diagonal blocks, waves 4-7 hold three off-diagonal blocks and two diagonal ones.
// Fragment f of a wave as for SYM = 3 (type 0: block row w, then block columns 4..7; types 1, 2: their block columns).
//   type 1 (base b): (b, b+1) (b, b+2) (b, b+3) + diagonal b, b+3      type 2: (b+1, b+2) (b+1, b+3) (b+2, b+3) + diagonal b+1, b+2
__host__ __device__ constexpr int d16_np(int ty) { return ty == 0 ? 4 : 3; }                       // off-diagonal blocks of the wave
__host__ __device__ constexpr int d16_ai(int ty, int n) { return (ty == 2 && n == 2) ? 1 : 0; }
__host__ __device__ constexpr int d16_bi(int ty, int n) { return ty == 2 ? (n == 0 ? 1 : 2) : n + 1; }
template <int V>
struct IntC {
    static constexpr int value = V;
};

// Class images of the k_pad = 256 symmetric kernel hold their 36 upper 32 x 32 blocks only, each 4 KB in one piece and row-major
// inside, laid out [group][block][class]: the classes of one (group, block) -- what one workgroup of class_sum_blocks_kernel adds
// up -- are ONE contiguous run of (2^R - 1) * 4 KB (252 KB at R = 6) instead of 63 pieces 144 KB apart; the scatter moves to the
// writer, whose 36 block stores per class image ride under the other class workgroup's MFMAs.
constexpr int64_t CLS_BLOCK = 1024;
__device__ __forceinline__ int cls_block(int ba, int bb) { return ba * 8 - ((ba * (ba - 1)) >> 1) + bb - ba; }

// CLS = 1 (with SYM = 3): class launches only (cls_cnt set) -- no targets, no dot products, no gradient, constant weight;
// only the 36 upper blocks are stored (class_sum_blocks_kernel mirrors them when it adds up a row's classes).
// The class-only build is compiled for FOUR waves per SIMD (<= 128 VGPRs: 80 of them accumulators) and uses one 32 KB image per
// stage (64 KB of LDS): two workgroups share a CU, and one multiplies while the other sits at its per-tile barrier.
template <int KP, int STAGGER = 1, int DIAG = 0, int SYM = 0, int CLS = 0>
__global__ __launch_bounds__(512, CLS ? 4 : 2) void row_hess_kernel(RowHessArgs g) {
    using C = RowHessCfg<KP>;
    static_assert(!SYM || KP == 256, "the symmetric block map is laid out for k_pad = 256");
    static_assert(!CLS || SYM >= 3, "the class-only build exists for the single-image symmetric kernel");
    constexpr bool ONE = SYM >= 3; // one staged image sqrt(w_j) o_j
    constexpr bool D16 = SYM == 4; // ... and the 8 diagonal blocks as three 16 x 16 sub-blocks each (see d16_* above)
    const bool late = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= 4;
    extern __shared__ __attribute__((aligned(16))) float rsm[];
    auto tile_of = [&](int b) { return rsm + (CLS ? 1 : 2) * b * C::TILE; }; // raw rows o_j      (B operand)
    auto wtile_of = [&](int b) { return rsm + (2 * b + 1) * C::TILE; };      // rows w_j * o_j    (A operand)

    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const bool cls = g.cls_cnt != nullptr;
    const bool split = !cls && g.nsplit > 1;
    const int chunk = split ? (int)(blockIdx.x / g.nrows) : 0;
    const int s0 = chunk * g.split_len;                      // first sample of this workgroup's share of the list
    const int64_t i = cls ? 0 : g.row0 + (split ? (int64_t)(blockIdx.x % g.nrows) : (int64_t)blockIdx.x);
    const int ns = cls ? g.cls_cnt[blockIdx.x] : (split ? (g.s - s0 < g.split_len ? g.s - s0 : g.split_len) : g.s); // samples of this row
    const int trow = t / C::LPR, tl16 = t % C::LPR;          // this thread's tile row and lane within the row
    const bool loader = t < 32 * C::LPR;                      // k_pad = 32: half of the threads cover the tile
    f32x4 u4[C::CPT];
#pragma unroll
    for (int q = 0; q < C::CPT; ++q) u4[q] = *reinterpret_cast<const f32x4 *>(g.F + i * KP + 4 * (q * C::LPR + tl16));
    const int32_t *list = cls ? g.idx + g.cls_off[blockIdx.x] : (g.idx ? g.idx + i * g.idx_stride + s0 : nullptr);
    const float *Ti = g.T + i * g.t_row;
    const int nt = (ns + 31) / 32;

    const int wm = wid / C::WN, wn = wid % C::WN;
    const bool mfma_wave = wid < C::WM * C::WN;
    const int wrow0 = wm * 32 * C::TM, wcol0 = wn * 32 * C::TN;

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 gacc[C::CPT];
#pragma unroll
    for (int q = 0; q < C::CPT; ++q) gacc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // SYM: block accumulators and this wave's block rows / columns (all wave-uniform)
    f32x16 hs[5];
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) hs[n][r] = 0.f;
    const int uw = __builtin_amdgcn_readfirstlane(wid);
    const int wty = uw < 4 ? 0 : ((uw & 1) ? 2 : 1);
    const int sbase = uw >= 6 ? 4 : 0;
    int ablk[2], bblk[4];
    if (wty == 0) {
        ablk[0] = ablk[1] = uw;
        bblk[0] = 4; bblk[1] = 5; bblk[2] = 6; bblk[3] = 7;
    } else if (wty == 1) {
        ablk[0] = sbase; ablk[1] = sbase + 3;
        bblk[0] = sbase; bblk[1] = sbase + 1; bblk[2] = sbase + 2; bblk[3] = sbase + 3;
    } else {
        ablk[0] = sbase + 1; ablk[1] = sbase + 2;
        bblk[0] = sbase + 1; bblk[1] = sbase + 2; bblk[2] = sbase + 3; bblk[3] = sbase + 3;
    }

    int jn = 0;          // index of the row about to be gathered
    f32x4 rr[C::CPT];    // this thread's chunks of the gathered row in flight
    float tt = 0.f;      // its target
    bool vv = false;     // sample exists (tail of the list)

    auto load_idx = [&](int tl) {
        const int q = 32 * tl + trow;
        const int qc = q < ns ? q : ns - 1;
        if constexpr (DIAG == 11) jn = trow + 32 * (int)(blockIdx.x & 63); // diagnostic: the same 32 rows every tile (cache hits), with the stage of DIAG 8
        else if constexpr (DIAG == 7) jn = qc + s0; // diagnostic: consecutive rows, no index stream
        else jn = list ? list[qc] : qc + s0;
    };
    auto gather = [&](int tl) {
        vv = 32 * tl + trow < ns;
        const float *src = g.O + (int64_t)jn * KP + 4 * tl16;
        if constexpr (DIAG == 9) { // diagnostic: one dword per lane instead of the row (with the stage of DIAG 8)
            rr[0][0] = src[0];
            return;
        }
        if constexpr (DIAG == 10) { // diagnostic: the same number of loads, one dword each (with the stage of DIAG 8)
#pragma unroll
            for (int q = 0; q < C::CPT; ++q) rr[q][0] = src[4 * q * C::LPR];
            tt = Ti[(int64_t)jn * g.t_col];
            return;
        }
#pragma unroll
        for (int q = 0; q < C::CPT; ++q) rr[q] = *reinterpret_cast<const f32x4 *>(src + 4 * q * C::LPR);
        if constexpr (!CLS && DIAG != 6) tt = Ti[(int64_t)jn * g.t_col]; // (DIAG 6: no target load)
    };
    // registers -> (z, r, w, gradient) -> LDS tile.  Branch-free in the link:
    // f = lk sigmoid(z) + (1 - lk) z, w = lk f (1 - f) + (1 - lk), lk = 0 / 1 -- exact, one term is always zero.
    const float lk = g.link ? 1.0f : 0.0f, nlk = 1.0f - lk;
    const float cls_sq = __builtin_amdgcn_sqrtf(fmaxf(g.scale, 0.0f));
    auto stage = [&](int nb) {
        // D16: odd sample rows keep the 16-column halves of every 32-column block swapped (16-byte chunk index ^ 4), so that the
        // 16-wide fragments of the diagonal sub-blocks -- lanes l and l + 16 on consecutive sample rows -- fall on different banks
        float *rdst = tile_of(nb) + trow * KP + 4 * (D16 ? (tl16 ^ ((trow & 1) << 2)) : tl16);
        float *wdst = wtile_of(nb) + trow * KP + 4 * tl16;
        if constexpr (CLS) { // linear link: every sample of the class weighs s
            const float sq = vv ? cls_sq : 0.0f;
#pragma unroll
            for (int q = 0; q < C::CPT; ++q) *reinterpret_cast<f32x4 *>(rdst + 4 * q * C::LPR) = sq * rr[q];
            return;
        }
        if constexpr (DIAG == 8 || DIAG == 9 || DIAG == 10 || DIAG == 11) { // diagnostic: wait for the gathered rows, nothing else
#pragma unroll
            for (int q = 0; q < C::CPT; ++q) gacc[0] += rr[q];
            if constexpr (DIAG == 10) gacc[0][0] += tt;
            return;
        }
        if constexpr (DIAG == 4) { // diagnostic: LDS writes only
#pragma unroll
            for (int q = 0; q < C::CPT; ++q) {
                *reinterpret_cast<f32x4 *>(rdst + 4 * q * C::LPR) = rr[q];
                *reinterpret_cast<f32x4 *>(wdst + 4 * q * C::LPR) = rr[q];
            }
            return;
        }
        float z = 0.f;
#pragma unroll
        for (int q = 0; q < C::CPT; ++q)
            z += u4[q][0] * rr[q][0] + u4[q][1] * rr[q][1] + u4[q][2] * rr[q][2] + u4[q][3] * rr[q][3];
        z = group_sum<C::LPR>(z);
        float slope;
        const float sg = sigmoid_slope_(z, slope);   // (the slope without the cancellation of s (1 - s): cmf_kernels.hip.h)
        const float f = lk * sg + nlk * z;
        const float valid = vv ? g.scale : 0.0f;
        const float res = valid * (f - tt);
        const float wgt = valid * (lk * slope + nlk);
        const float sqw = __builtin_amdgcn_sqrtf(fmaxf(wgt, 0.0f)); // SYM == 3 only (launched when the weights are non-negative)
#pragma unroll
        for (int q = 0; q < C::CPT; ++q) {
            gacc[q] += res * rr[q];
            if constexpr (DIAG == 5) { // diagnostic: the arithmetic only
                gacc[q] += wgt * rr[q];
                continue;
            }
            if constexpr (ONE) {
                *reinterpret_cast<f32x4 *>(rdst + 4 * q * C::LPR) = sqw * rr[q];
            } else {
                *reinterpret_cast<f32x4 *>(rdst + 4 * q * C::LPR) = rr[q];
                *reinterpret_cast<f32x4 *>(wdst + 4 * q * C::LPR) = wgt * rr[q];
            }
        }
    };
    auto mfma_tile = [&](int cb, bool do_stage, int nb, bool do_gather, int tl_gather, bool do_idx, int tl_idx) {
        const float *Rt = tile_of(cb);
        const float *Wt = wtile_of(cb);
        float a[2][C::TM], b[2][C::TN];
        // exactly the TN GEMM's inner loop: A fragments from the pre-scaled image, B fragments from the raw one
        auto ld_frag = [&](int sidx, float *da, float *db) {
            const int kk = 2 * sidx + lh;
            VecLoad<C::TM>::ld(Wt + kk * KP + wrow0 + C::TM * l31, da);
            VecLoad<C::TN>::ld(Rt + kk * KP + wcol0 + C::TN * l31, db);
        };
        if (mfma_wave) ld_frag(0, a[0], b[0]);
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) {
            if (mfma_wave && sidx + 1 < 16) ld_frag(sidx + 1, a[(sidx + 1) & 1], b[(sidx + 1) & 1]);
            // the VALU-heavy staging of waves 4-7 runs half a step after that of waves 0-3, so that on every
            // SIMD one wave's dot products / sigmoids sit beside its partner's MFMAs (STAGGER = 0: lockstep)
            const int g0 = (STAGGER && late) ? 8 : 0;
            if (sidx == g0) {
                if (DIAG != 1 && DIAG != 3 && do_stage && loader) stage(nb);
            } else if (sidx == g0 + 1) {
                if (DIAG != 1 && DIAG != 2 && do_gather && loader) gather(tl_gather);
                if (DIAG != 1 && DIAG != 2 && do_idx && loader) load_idx(tl_idx);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (mfma_wave) {
#pragma unroll
                for (int x = 0; x < C::TM; ++x)
#pragma unroll
                    for (int y = 0; y < C::TN; ++y)
                        acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sidx & 1][x], b[sidx & 1][y], acc[x][y], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto sym_tile = [&](auto typ, int cb, bool do_stage, int nb, bool do_gather, int tl_gather, bool do_idx, int tl_idx) {
        constexpr int TY = decltype(typ)::value;
        constexpr int NP = sym_np(TY), NA = sym_na(TY), NB = sym_nb(TY);
        const float *Rt = tile_of(cb) + l31;
        const float *Wt = wtile_of(cb) + l31;
        float a[2][2], b[2][5];
        auto ld_frag = [&](int sidx, float *da, float *db) {
            const int kk = 2 * sidx + lh;
            if constexpr (ONE) { // one image: db[f] = fragment f of this wave (da unused)
#pragma unroll
                for (int f = 0; f < s3_nf(TY); ++f) db[f] = Rt[kk * KP + 32 * (TY == 0 ? (f == 0 ? ablk[0] : bblk[f - 1]) : bblk[f])];
            } else {
#pragma unroll
                for (int x = 0; x < NA; ++x) da[x] = Wt[kk * KP + 32 * ablk[x]];
#pragma unroll
                for (int y = 0; y < NB; ++y) db[y] = Rt[kk * KP + 32 * bblk[y]];
            }
        };
        ld_frag(0, a[0], b[0]);
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) {
            if (sidx + 1 < 16) ld_frag(sidx + 1, a[(sidx + 1) & 1], b[(sidx + 1) & 1]);
            const int g0 = (STAGGER && late) ? 8 : 0;
            if (sidx == g0) {
                if (DIAG != 1 && DIAG != 3 && do_stage && loader) stage(nb);
            } else if (sidx == g0 + 1) {
                if (DIAG != 1 && DIAG != 2 && do_gather && loader) gather(tl_gather);
                if (DIAG != 1 && DIAG != 2 && do_idx && loader) load_idx(tl_idx);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                if constexpr (ONE)
                    hs[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[sidx & 1][s3_ai(TY, n)], b[sidx & 1][s3_bi(TY, n)], hs[n], 0, 0, 0);
                else
                    hs[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sidx & 1][sym_ai(TY, n)], b[sidx & 1][sym_bi(TY, n)], hs[n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // the last tile of a list when it holds fewer than 29 samples: only the K-steps that carry samples (two per MFMA, four per
    // turn of this loop; the rows behind them are staged as zeros anyway).  Class lists of ~256 samples end in a half-empty
    // tile on average: 6 % of their K-steps.
    auto sym_tail = [&](auto typ, int cb, int nsteps) {
        constexpr int TY = decltype(typ)::value;
        constexpr int NP = sym_np(TY), NA = sym_na(TY), NB = sym_nb(TY);
        const float *Rt = tile_of(cb) + l31;
        const float *Wt = wtile_of(cb) + l31;
        float a[2][2], b[2][5];
        auto ld_frag = [&](int sidx, float *da, float *db) {
            const int kk = 2 * sidx + lh;
            if constexpr (ONE) {
#pragma unroll
                for (int f = 0; f < s3_nf(TY); ++f) db[f] = Rt[kk * KP + 32 * (TY == 0 ? (f == 0 ? ablk[0] : bblk[f - 1]) : bblk[f])];
            } else {
#pragma unroll
                for (int x = 0; x < NA; ++x) da[x] = Wt[kk * KP + 32 * ablk[x]];
#pragma unroll
                for (int y = 0; y < NB; ++y) db[y] = Rt[kk * KP + 32 * bblk[y]];
            }
        };
        auto mm = [&](const float *fa, const float *fb) {
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                if constexpr (ONE) hs[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[s3_ai(TY, n)], fb[s3_bi(TY, n)], hs[n], 0, 0, 0);
                else hs[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[sym_ai(TY, n)], fb[sym_bi(TY, n)], hs[n], 0, 0, 0);
            }
        };
        ld_frag(0, a[0], b[0]);
        for (int sidx = 0; sidx < nsteps; sidx += 2) { // sidx + 1 <= 15 always (nsteps <= 14)
            ld_frag(sidx + 1, a[1], b[1]);
            mm(a[0], b[0]);
            ld_frag(sidx + 2, a[0], b[0]);
            mm(a[1], b[1]);
        }
    };
    // ---- SYM = 4: the same two loops with the diagonal blocks on the 16-wide instruction
    f32x4 hd[2][3]; // diagonal block d of the wave: sub-blocks (0,0), (0,1), (1,1)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int x = 0; x < 3; ++x) hd[d][x] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int dblk0 = wty == 1 ? sbase : sbase + 1, dblk1 = wty == 1 ? sbase + 3 : sbase + 2;
    const int kq16 = lane >> 4, i16 = lane & 15, hx16 = (kq16 & 1) << 4;
    // 32-wide fragment f at sample row kk = 2 sidx + lh (the swizzle: odd rows hold their halves swapped)
    auto d16_frag = [&](auto typ, const float *Rt, int sidx, float *db) {
        constexpr int TY = decltype(typ)::value;
        const int kk = 2 * sidx + lh;
#pragma unroll
        for (int f = 0; f < s3_nf(TY); ++f) db[f] = Rt[kk * KP + 32 * (TY == 0 ? (f == 0 ? ablk[0] : bblk[f - 1]) : bblk[f])];
    };
    // 16-wide fragments of the two diagonal blocks over the samples 4 tq .. 4 tq + 3: lane 16 k + i holds sample 4 tq + k, column 16 h + i
    auto d16_diag = [&](const float *R16, int tq, float (*de)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            de[0][h] = R16[4 * tq * KP + 32 * dblk0 + ((16 * h) ^ hx16)];
            de[1][h] = R16[4 * tq * KP + 32 * dblk1 + ((16 * h) ^ hx16)];
        }
    };
    auto d16_mm = [&](auto typ, const float *fb) {
        constexpr int TY = decltype(typ)::value;
#pragma unroll
        for (int n = 0; n < d16_np(TY); ++n)
            hs[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[d16_ai(TY, n)], fb[d16_bi(TY, n)], hs[n], 0, 0, 0);
    };
    auto d16_mmd = [&](int d, const float (*de)[2]) {
        hd[d][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(de[d][0], de[d][0], hd[d][0], 0, 0, 0);
        hd[d][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(de[d][0], de[d][1], hd[d][1], 0, 0, 0);
        hd[d][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(de[d][1], de[d][1], hd[d][2], 0, 0, 0);
    };
    auto d16_tile = [&](auto typ, int cb, bool do_stage, int nb, bool do_gather, int tl_gather, bool do_idx, int tl_idx) {
        constexpr int TY = decltype(typ)::value;
        const float *Rt = tile_of(cb) + (l31 ^ (lh << 4));
        const float *R16 = tile_of(cb) + kq16 * KP + i16;
        float b[2][5], e[2][2][2];
        d16_frag(typ, Rt, 0, b[0]);
        if constexpr (TY != 0) d16_diag(R16, 0, e[0]);
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) {
            if (sidx + 1 < 16) d16_frag(typ, Rt, sidx + 1, b[(sidx + 1) & 1]);
            if constexpr (TY != 0) {
                if ((sidx & 1) == 0 && sidx + 2 < 16) d16_diag(R16, sidx / 2 + 1, e[(sidx / 2 + 1) & 1]);
            }
            const int g0 = (STAGGER && late) ? 8 : 0;
            if (sidx == g0) {
                if (DIAG != 1 && DIAG != 3 && do_stage && loader) stage(nb);
            } else if (sidx == g0 + 1) {
                if (DIAG != 1 && DIAG != 2 && do_gather && loader) gather(tl_gather);
                if (DIAG != 1 && DIAG != 2 && do_idx && loader) load_idx(tl_idx);
            }
            __builtin_amdgcn_sched_barrier(0);
            d16_mm(typ, b[sidx & 1]);
            if constexpr (TY != 0) d16_mmd(sidx & 1, e[(sidx / 2) & 1]); // diagonal block 0 on the even steps, 1 on the odd ones
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto d16_tail = [&](auto typ, int cb, int nsteps) {
        constexpr int TY = decltype(typ)::value;
        const float *Rt = tile_of(cb) + (l31 ^ (lh << 4));
        const float *R16 = tile_of(cb) + kq16 * KP + i16;
        float b[2][5], e[2][2];
        d16_frag(typ, Rt, 0, b[0]);
        if constexpr (TY != 0) d16_diag(R16, 0, e);
        for (int sidx = 0; sidx < nsteps; sidx += 2) { // sidx + 2 <= 14 always (nsteps <= 14)
            d16_frag(typ, Rt, sidx + 1, b[1]);
            d16_mm(typ, b[0]);
            if constexpr (TY != 0) d16_mmd(0, e);
            d16_frag(typ, Rt, sidx + 2, b[0]);
            d16_mm(typ, b[1]);
            if constexpr (TY != 0) {
                d16_mmd(1, e);
                d16_diag(R16, sidx / 2 + 1, e);
            }
        }
    };
    const int tail_steps = nt > 0 ? (ns - 32 * (nt - 1) + 1) / 2 : 0; // K-steps of the last tile that carry samples (1..16)

    if (nt > 0) {
        if (loader) {
            load_idx(0);
            gather(0);
            stage(0);
            if (nt > 1) {
                load_idx(1);
                gather(1);
            }
            if (nt > 2) load_idx(2);
        }
        __syncthreads();
        if constexpr (SYM) {
            // the three wave types run their own copy of the loop (same barrier count in each)
            auto run = [&](auto typ) {
                const int full = tail_steps <= 14 ? nt - 1 : nt;
                for (int tl = 0; tl < full; ++tl) {
                    if constexpr (D16) d16_tile(typ, tl & 1, tl + 1 < nt, (tl + 1) & 1, tl + 2 < nt, tl + 2, tl + 3 < nt, tl + 3);
                    else sym_tile(typ, tl & 1, tl + 1 < nt, (tl + 1) & 1, tl + 2 < nt, tl + 2, tl + 3 < nt, tl + 3);
                    __syncthreads();
                }
                if (full < nt) {
                    if constexpr (D16) d16_tail(typ, (nt - 1) & 1, tail_steps);
                    else sym_tail(typ, (nt - 1) & 1, tail_steps);
                }
            };
            if (wty == 0) run(IntC<0>{});
            else if (wty == 1) run(IntC<1>{});
            else run(IntC<2>{});
        } else {
            for (int tl = 0; tl < nt; ++tl) {
                // tile tl is in LDS; tile tl+1 is in registers; jn holds the indices of tile tl+2
                mfma_tile(tl & 1, tl + 1 < nt, (tl + 1) & 1, tl + 2 < nt, tl + 2, tl + 3 < nt, tl + 3);
                __syncthreads();
            }
        }
    }

    // ---- H_i tile out (rows interleaved like the TN GEMM form)
    float *Hi = g.H + (int64_t)blockIdx.x * KP * KP;
    if constexpr (CLS) { // block b of class q of group grp at ((grp * 36 + b) * nc1 + q) * 4 KB
        const int64_t grp = blockIdx.x / g.cls_nc1, q = blockIdx.x % g.cls_nc1;
        Hi = g.H + (grp * 36 * g.cls_nc1 + q) * CLS_BLOCK;
    }
    const int64_t cls_bstride = CLS ? (int64_t)g.cls_nc1 * CLS_BLOCK : 0;
    if constexpr (SYM) {
        auto emit = [&](auto typ) {
            constexpr int TY = decltype(typ)::value;
            if constexpr (D16 && TY != 0) { // the two diagonal blocks: sub-block (ha, hb) at rows 16 ha + 4 (lane / 16) + r, column 16 hb + lane % 16
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int bd = d == 0 ? dblk0 : dblk1;
#pragma unroll
                    for (int x = 0; x < 3; ++x) {
                        const int ha = x == 2 ? 1 : 0, hb = x == 0 ? 0 : 1;
                        const int rloc = 16 * ha + 4 * kq16, cloc = 16 * hb + i16;
                        if constexpr (CLS) {
                            float *blk = Hi + cls_block(bd, bd) * cls_bstride;
#pragma unroll
                            for (int r = 0; r < 4; ++r) blk[(rloc + r) * 32 + cloc] = hd[d][x][r];
                            if (x == 1) *reinterpret_cast<f32x4 *>(blk + cloc * 32 + rloc) = hd[d][x];
                            continue;
                        }
                        float *dst = Hi + (32 * bd + rloc) * KP + 32 * bd + cloc;
                        f32x4 v = hd[d][x];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 32 * bd + rloc + r, col = 32 * bd + cloc;
                            if (g.accumulate & 1) v[r] += dst[r * KP];
                            else {
                                if (g.S) v[r] += g.S[row * KP + col];
                                if (row == col && row < g.kvalid) v[r] += g.diag;
                            }
                            dst[r * KP] = v[r];
                        }
                        if (x == 1) *reinterpret_cast<f32x4 *>(Hi + (32 * bd + cloc) * KP + 32 * bd + rloc) = v;
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < (D16 ? d16_np(TY) : sym_np(TY)); ++n) {
                const int ai_ = D16 ? d16_ai(TY, n) : 0, bi_ = D16 ? d16_bi(TY, n) : 0;
                // (D16: operands are fragment indices -- type 0: fragment 0 = block row w, 1..4 = block columns 4..7; else block columns)
                const int ba = D16 ? (TY == 0 ? ablk[0] : bblk[ai_]) : ablk[sym_ai(TY, n)];
                const int bb = D16 ? (TY == 0 ? bblk[bi_ > 0 ? bi_ - 1 : 0] : bblk[bi_]) : bblk[sym_bi(TY, n)];
                if constexpr (CLS) { // block-major image: upper block (ba, bb) is 4 KB in one piece, row-major inside
                    float *blk = Hi + cls_block(ba, bb) * cls_bstride + 4 * lh * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) blk[((r & 3) + 8 * (r >> 2)) * 32] = hs[n][r];
                    continue;
                }
                float *blk = Hi + (32 * ba + 4 * lh) * KP + 32 * bb + l31; // + (j + 8q) rows: register r = 4q + j
                const int col = 32 * bb + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * ba + 4 * lh + (r & 3) + 8 * (r >> 2);
                    float *dst = blk + ((r & 3) + 8 * (r >> 2)) * KP;
                    float v = hs[n][r];
                    if (g.accumulate & 1) v += *dst;
                    else { // the launch that starts H_i also adds the shared part and the diagonal (S is symmetric)
                        if (g.S) v += g.S[row * KP + col];
                        if (row == col && row < g.kvalid) v += g.diag;
                    }
                    hs[n][r] = v;
                    *dst = v;
                }
                if (!CLS && ba != bb) { // mirror image: row = this lane's column, four consecutive columns per register quad
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
    } else if (mfma_wave) {
#pragma unroll
        for (int x = 0; x < C::TM; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rrw = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int row = wrow0 + C::TM * rrw + x;
                const int col0 = wcol0 + C::TN * l31;
                float *dst = Hi + row * KP + col0;
                float v[C::TN];
#pragma unroll
                for (int y = 0; y < C::TN; ++y) {
                    v[y] = acc[x][y][r];
                    if (g.accumulate & 1) v[y] += dst[y];
                    else {
                        if (g.S) v[y] += g.S[row * KP + col0 + y];
                        if (row == col0 + y && row < g.kvalid) v[y] += g.diag;
                    }
                }
                if constexpr (C::TN == 4) *reinterpret_cast<f32x4 *>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                else if constexpr (C::TN == 2) *reinterpret_cast<f32x2 *>(dst) = f32x2{v[0], v[1]};
                else dst[0] = v[0];
            }
    }
    // ---- gradient part: sum the per-thread partials that share a column chunk
    if (CLS || cls) return;
    __syncthreads();
    constexpr int NREP = 32;          // one partial per tile row
    float *gr = rsm;                  // [NREP][KP] staging (the tiles are dead now)
    if (loader) {
#pragma unroll
        for (int q = 0; q < C::CPT; ++q) *reinterpret_cast<f32x4 *>(gr + trow * KP + 4 * (q * C::LPR + tl16)) = gacc[q];
    }
    __syncthreads();
    if (t < KP) {
        float sacc = 0.f;
        for (int rep = 0; rep < NREP; ++rep) sacc += gr[rep * KP + t];
        float *dst = g.G + (int64_t)chunk * g.g_split_stride + i * KP + t;
        *dst = sacc + ((g.accumulate & 2) ? *dst : 0.f);
    }
}

// ---- shared partial sums for linear-link sides with per-row sampling ("row classes") -----------------------------------
// With a linear link the Hessian weights do not depend on the row: H_i = s sum_{j in S_i} o_j o_j^T (cmf_solvers.py:414-428
// with the identity link: d2 = 1).  R consecutive rows split the candidates j into 2^R - 1 classes by the set of rows whose
// list holds j; the outer-product sum of a class is formed ONCE and every row adds up the 2^(R-1) classes it belongs to.
// At sg_sample_ratio = 0.5 and R = 4 that is 15/16 n samples per four rows instead of 2 n: 2.13x fewer MFMAs for the same sums.
//
// class_lists_kernel: one workgroup per group of R rows; membership patterns in an LDS byte per candidate, then the
// class lists in ascending candidate order (deterministic): cidx[g * cap + ...], class q = pattern q + 1 of group g at
// coff / ccnt [g * (2^R - 1) + q].  patg (optional): the pattern bytes themselves, [group][n] -- the sample mask of the
// group's rows in 1/R of the bytes (mask_from_patterns_kernel expands it for the gradient GEMM's epilogue).
// LDS layout: thread t owns the candidate words t W .. t W + W - 1 (4 candidates each, ascending), stored transposed
// (word w at (w % W) 256 + w / W) so that the 256 threads walk their ranges bank-conflict-free; hist[x][p] (pitch 2^R + 1,
// 16-bit: three workgroups per CU hide each other's list-read latency) counts thread x's candidates of pattern p,
// prefix-summed over x in 256 / 2^R segments per pattern.
__global__ __launch_bounds__(256) void class_lists_kernel(const int32_t *lists, int64_t per, int64_t nlists, int n, int R, int32_t *cidx,
                                                          int64_t cap, int64_t *coff, int32_t *ccnt, unsigned long long *gathered,
                                                          uint8_t *patg) {
    extern __shared__ unsigned patw[]; // W * 256 pattern words, then hist[256][NC + 1] (16-bit: a count is at most 4 W, a
                                       // prefix inside a segment of NC threads at most 4 W NC <= 32768 for n <= 131072)
    __shared__ int tot[64], cbase[64], segoff[256]; // segoff[seg * NC + p]
    const int t = threadIdx.x, NC = 1 << R, HP = NC + 1;
    const int64_t grp = blockIdx.x;
    const int nw = (n + 3) / 4, W = (nw + 255) / 256;
    unsigned short *hist = reinterpret_cast<unsigned short *>(patw + W * 256);
    for (int w = t; w < W * 256; w += 256) patw[w] = 0u;
    for (int x = t; x < 256 * HP; x += 256) hist[x] = 0;
    __syncthreads();
    for (int r = 0; r < R; ++r) {
        const int64_t row = grp * R + r;
        if (row >= nlists) break;
        const int32_t *l = lists + row * per;
        for (int64_t q0 = t; q0 < per; q0 += 256 * 16) { // sixteen list entries in flight per thread: the loop is bound by their latency
            int jv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) jv[u] = q0 + 256 * u < per ? l[q0 + 256 * u] : -1;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (jv[u] >= 0) {
                    const int w = jv[u] >> 2;
                    atomicOr(&patw[(w % W) * 256 + w / W], 1u << (8 * (jv[u] & 3) + r));
                }
        }
    }
    __syncthreads();
    if (patg) { // pattern bytes out, candidate order
        uint8_t *pg = patg + grp * (int64_t)n;
        for (int j = t; j < n; j += 256) {
            const int w = j >> 2;
            pg[j] = (uint8_t)((patw[(w % W) * 256 + w / W] >> (8 * (j & 3))) & 0xffu);
        }
    }
    for (int i = 0; i < W; ++i) {
        const unsigned word = patw[i * 256 + t];
        const int j = 4 * (t * W + i);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j + e < n) hist[t * HP + ((word >> (8 * e)) & 0xffu)]++;
    }
    __syncthreads();
    // exclusive prefix over x (threads) for every pattern: thread (p, seg) scans the NC entries of its segment
    const int pp = t & (NC - 1), seg = t / NC, segs = 256 / NC;
    {
        int run = 0;
        for (int x = seg * NC; x < (seg + 1) * NC; ++x) {
            const int v = hist[x * HP + pp];
            hist[x * HP + pp] = (unsigned short)run;
            run += v;
        }
        segoff[seg * NC + pp] = run; // segment total for now
    }
    __syncthreads();
    if (t < NC) { // segment totals -> segment offsets and the pattern total
        int run = 0;
        for (int sg = 0; sg < segs; ++sg) {
            const int v = segoff[sg * NC + t];
            segoff[sg * NC + t] = run;
            run += v;
        }
        tot[t] = run;
    }
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int q = 1; q < NC; ++q) {
            cbase[q] = run;
            coff[grp * (NC - 1) + q - 1] = grp * cap + run;
            ccnt[grp * (NC - 1) + q - 1] = tot[q];
            run += tot[q];
        }
        if (gathered) atomicAdd(gathered, (unsigned long long)run); // accounting only (cmf_rowhess_samples)
    }
    __syncthreads();
    int32_t *out = cidx + grp * cap;
    const int myseg = t / NC;
    for (int i = 0; i < W; ++i) {
        const unsigned word = patw[i * 256 + t];
        const int j = 4 * (t * W + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = (int)((word >> (8 * e)) & 0xffu);
            if (q && j + e < n) out[cbase[q] + segoff[myseg * NC + q] + hist[t * HP + q]++] = j + e;
        }
    }
}

// byte mask of a class side from the groups' pattern bytes (what scatter_mask_kernel builds from the lists, without its
// scattered byte stores).  by_row = 1: list i holds columns of image row i -> mask[i][j];  by_row = 0: list i holds rows
// of image column i -> mask[j][i].  i = g R + r  <->  bit r of patg[g][j].
__global__ __launch_bounds__(256) void mask_rows_from_patterns_kernel(uint8_t *mask, int64_t ld, const uint8_t *patg, int64_t ngroups, int n,
                                                                      int R, int64_t nlists) {
    const int64_t total = ngroups * (int64_t)n;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = x / n, j = x % n;
        const unsigned pb = patg[x];
        for (int r = 0; r < R; ++r) {
            const int64_t i = g * R + r;
            if (i < nlists) mask[i * ld + j] = (uint8_t)((pb >> r) & 1u);
        }
    }
}
__global__ __launch_bounds__(256) void mask_cols_from_patterns_kernel(uint8_t *mask, int64_t ld, const uint8_t *patg, int64_t ngroups, int n,
                                                                      int R, int64_t nlists) {
    // tile: 32 groups x 64 candidates through LDS; a candidate's 32 R mask bytes are contiguous in its image row
    __shared__ uint8_t tile[32][68];
    const int t = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 64;
    for (int x = t; x < 32 * 64; x += 256) {
        const int gg = x >> 6, jj = x & 63;
        uint8_t v = 0;
        if (g0 + gg < ngroups && j0 + jj < n) v = patg[(g0 + gg) * (int64_t)n + j0 + jj];
        tile[gg][jj] = v;
    }
    __syncthreads();
    // a candidate's 32 R bytes = 8 R words, consecutive lanes -> consecutive words of one image row
    const int wpr = 8 * R;
    for (int x = t; x < 64 * wpr; x += 256) {
        const int jj = x / wpr, wi = x % wpr;
        if (j0 + jj >= n) break;
        if (g0 * R + 4 * wi >= ld) continue; // past the padded width of the image row
        unsigned word = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int b = 4 * wi + e, gg = b / R, r = b % R;
            if ((g0 + gg) * R + r < nlists) word |= (unsigned)((tile[gg][jj] >> r) & 1u) << (8 * e);
        }
        *reinterpret_cast<unsigned *>(mask + (j0 + jj) * ld + g0 * R + 4 * wi) = word;
    }
}

// H_row = sum of the class images the row belongs to (+ S + diag I when it starts H, += otherwise).  C holds the images
// of the groups of this chunk: group (grow0 + row) / R - grow0 / R, class q at [(group * (2^R - 1) + q) * kp^2].
__global__ __launch_bounds__(256) void class_sum_kernel(float *H, const float *C, const float *S, float diag, int64_t nrows, int64_t grow0,
                                                        int R, int kp, int kvalid, int accumulate) {
    const int64_t kk4 = (int64_t)kp * kp / 4, total = nrows * kk4;
    const int NC = 1 << R;
    const int64_t g0 = grow0 / R;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = x / kk4, e4 = x % kk4;
        const int64_t grp = (grow0 + row) / R - g0;
        const int bit = (int)((grow0 + row) % R);
        const float *base = C + grp * (NC - 1) * kk4 * 4 + e4 * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int q = 1; q < NC; ++q)
            if ((q >> bit) & 1) acc += *reinterpret_cast<const f32x4 *>(base + (int64_t)(q - 1) * kk4 * 4);
        f32x4 *dst = reinterpret_cast<f32x4 *>(H + row * kk4 * 4 + e4 * 4);
        if (accumulate) acc += *dst;
        else {
            const int e = (int)(e4 * 4), r = e / kp, c0 = e % kp;
            if (S) acc += *reinterpret_cast<const f32x4 *>(S + e);
            if (r < kvalid && r >= c0 && r < c0 + 4) acc[r - c0] += diag;
        }
        *dst = acc;
    }
}

// The same for k_pad = 256 class images that hold only their 36 upper 32 x 32 blocks (row_hess_kernel<..., CLS = 1>): one
// workgroup per (group, block) item reads every class block of the group ONCE, adds it to the running sums of the rows it
// belongs to (classes in ascending order for every row: the order does not depend on R or on the grid), then stores each
// row's block and -- through an LDS transpose -- its mirror image.  nrows rows starting at a group boundary.
// cert (optional): per group two more sums over classes, the samples common to ALL rows of its first `split` rows and to
// all of the remaining ones -- a positive semi-definite part of each of those rows' Hessians (upper blocks only: what the
// Cholesky kernel reads), image [2 group + half].
template <int NB>
__global__ __launch_bounds__(256) void class_sum_blocks_kernel(float *H, const float *C, const float *S, float diag, int64_t nrows,
                                                               int R, int kvalid, int accumulate, float *cert, int split) {
    constexpr int KP = 256, RMAX = 6;
    constexpr int64_t KK = (int64_t)KP * KP;
    __shared__ float tile[32][33];
    const int NC = 1 << R;
    const int64_t ngroups = (nrows + R - 1) / R, items = ngroups * 36;
    const int t = threadIdx.x, r = t >> 3, c4 = t & 7;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t grp = item / 36;
        int rem = (int)(item % 36), ba = 0;
        while (rem >= 8 - ba) { rem -= 8 - ba; ++ba; }
        const int bb = ba + rem;
        const int off = (32 * ba + r) * KP + 32 * bb + 4 * c4;
        const float *base = C + (grp * 36 + item % 36) * (NC - 1) * CLS_BLOCK + 32 * r + 4 * c4; // [group][block][class]: one run per item
        f32x4 acc[RMAX], pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < RMAX; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ma = (1 << split) - 1, mb = (NC - 1) ^ ma;
        // NB class blocks in flight per thread, added in ascending class order
        for (int q0 = 1; q0 < NC; q0 += NB) {
            f32x4 v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u)
                v[u] = (q0 + u < NC) ? *reinterpret_cast<const f32x4 *>(base + (int64_t)(q0 + u - 1) * CLS_BLOCK) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int q = q0 + u;
                if (q < NC) {
#pragma unroll
                    for (int m = 0; m < RMAX; ++m)
                        if ((q >> m) & 1) acc[m] += v[u]; // bits >= R are never set
                    if ((q & ma) == ma) pa += v[u];
                    if ((q & mb) == mb) pb += v[u];
                }
            }
        }
        if (cert) {
            *reinterpret_cast<f32x4 *>(cert + (2 * grp) * KK + off) = pa;
            *reinterpret_cast<f32x4 *>(cert + (2 * grp + 1) * KK + off) = pb;
        }
#pragma unroll
        for (int m = 0; m < RMAX; ++m) {
            const int64_t row = grp * R + m;
            if (m >= R || row >= nrows) break; // uniform over the workgroup
            f32x4 a = acc[m];
            f32x4 *dst = reinterpret_cast<f32x4 *>(H + row * KK + off);
            if (accumulate) a += *dst; // H is symmetric: the mirror image below receives the same totals
            else {
                if (S) a += *reinterpret_cast<const f32x4 *>(S + off);
                const int gr = 32 * ba + r;
                if (ba == bb && (r >> 2) == c4 && gr < kvalid) a[r & 3] += diag;
            }
            *dst = a;
            if (ba != bb) {
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[r][4 * c4 + e] = a[e];
                __syncthreads();
                const f32x4 mm = {tile[4 * c4][r], tile[4 * c4 + 1][r], tile[4 * c4 + 2][r], tile[4 * c4 + 3][r]};
                *reinterpret_cast<f32x4 *>(H + row * KK + (32 * bb + r) * KP + 32 * ba + 4 * c4) = mm;
                __syncthreads();
            }
        }
    }
}

} // namespace cmfk
