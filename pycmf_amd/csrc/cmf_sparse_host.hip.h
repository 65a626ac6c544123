// cmf_sparse_host.hip.h -- host side of the native CSR path (included by cmf_api.hip)

static int csr_upload(cmf_ctx *c, CsrDev &dst, const int64_t *indptr, const int32_t *indices, const float *vals, int64_t rows,
                      int64_t cols, int64_t nnz) {
    dst.rows = rows; dst.cols = cols; dst.nnz = nnz;
    CHK(dev_alloc(c, (void **)&dst.indptr, (size_t)(rows + 1) * sizeof(int64_t), false));
    CHK(dev_alloc(c, (void **)&dst.idx, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int32_t), false));
    CHK(dev_alloc(c, (void **)&dst.val, (size_t)std::max<int64_t>(nnz, 1) * sizeof(float), false));
    HIPCHK(hipMemcpyAsync(dst.indptr, indptr, (size_t)(rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    if (nnz > 0) {
        HIPCHK(hipMemcpyAsync(dst.idx, indices, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(dst.val, vals, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return CMF_OK;
}

// Regroup a host CSR matrix for spmm_blocked_kernel (see cmf_sparse.hip.h): accumulator rows = the matrix rows, a row with more
// non-zeros than a wave's fair share of a group cut into pieces (r06); groups of at most G accumulator rows, closed once they hold
// their share of non-zeros (nnz-balanced work items: the workgroups of an XCD class re-align in front of every stretch, so a round
// lasts as long as its slowest group); the accumulator rows of a group dealt to its eight waves longest-first; column blocks of B
// gathered rows; counting sort by (group, wave, block).
static int bcsr_build_upload(cmf_ctx *c, CsrDev &dst, const int64_t *indptr, const int32_t *indices, const float *vals, int64_t rows,
                             int64_t cols, int64_t nnz, int G, int64_t B) {
    // entries q0 + i * stride < q1 of the CSR arrays; slot: of the partial buffer (a piece of a split row), or -1.  The pieces of a row
    // INTERLEAVE (piece i takes every np-th entry from the i-th on): each spans the row's whole column range, so that it has work in every
    // stretch of the lockstep sweep over the column blocks -- contiguous pieces would each sit in one or two blocks and wait out the rest
    struct Piece { int32_t row; int32_t slot; int64_t q0, q1, stride; int64_t n() const { return q1 > q0 ? (q1 - q0 + stride - 1) / stride : 0; } };
    const bool split = c->opt_spmm_split != 0;
    const double avg_group = (double)nnz / (double)std::max<int64_t>(rows, 1) * G;          // non-zeros of G average rows
    // a piece: a quarter of one wave's fair share of a group (a wave then holds several, and longest-first evens the waves out to
    // ~10 %; at a whole share a group of ten pieces left two waves with double the work)
    const int64_t share = std::max<int64_t>(1024, (int64_t)(avg_group / (4 * cmfk::BCSR_NW)));
    std::vector<Piece> pc;
    pc.reserve((size_t)rows + 64);
    std::vector<int32_t> prow, pfirst, pcnt;      // split rows: output row, first slot, number of pieces
    int32_t nslots = 0;
    for (int64_t r = 0; r < rows; ++r) {
        const int64_t rn = indptr[r + 1] - indptr[r];
        if (split && rn > share + share / 4) {
            const int64_t np = (rn + share - 1) / share;
            prow.push_back((int32_t)r); pfirst.push_back(nslots); pcnt.push_back((int32_t)np);
            for (int64_t i = 0; i < np; ++i) pc.push_back(Piece{(int32_t)r, nslots + (int32_t)i, indptr[r] + i, indptr[r + 1], np});
            nslots += (int32_t)np;
        } else
            pc.push_back(Piece{(int32_t)r, -1, indptr[r], indptr[r + 1], 1});
    }
    const int64_t nacc = (int64_t)pc.size();
    // Accumulator rows need not follow the matrix's row order (vmap says where each one goes): heaviest first.  A group then holds
    // rows of like weight, and consecutive groups -- the ones the workgroups of an XCD class take in the same round, re-aligning at
    // every stretch -- weigh alike: with the rows in matrix order a group of 136 random words held one hot word or none, and every
    // round waited for its heaviest group (c5z, X^T U: 55 ms per launch against 7 for uniform columns).
    if (split) std::stable_sort(pc.begin(), pc.end(), [](const Piece &x, const Piece &y) { return x.n() > y.n(); });
    std::vector<int32_t> grow;
    grow.push_back(0);
    // (round 5: closed at twice the average share; with pieces no row exceeds a quarter of a group, and equal groups are what the
    // rendezvous wants)
    const double cap = std::max(64.0, (split ? 1.25 : 2.0) * avg_group);
    int64_t gn = 0;
    int cnt = 0;
    for (int64_t a = 0; a < nacc; ++a) {
        const int64_t rn = pc[a].n();
        if (cnt > 0 && (cnt == G || (double)(gn + rn) > cap)) { grow.push_back((int32_t)a); gn = 0; cnt = 0; }
        gn += rn; ++cnt;
    }
    grow.push_back((int32_t)nacc);
    const int64_t ngroups = (int64_t)grow.size() - 1;
    const int64_t nblocks = std::max<int64_t>(1, (cols + B - 1) / B);
    if (ngroups * nblocks * cmfk::BCSR_NW > ((int64_t)1 << 31)) return CMF_EUNSUPPORTED; // sort table too large: keep the plain CSR kernel
    // wave of every accumulator row: longest first onto the least loaded wave of its group (round 5: row % 8)
    std::vector<uint8_t> wave_of((size_t)nacc);
    std::vector<int32_t> vmap((size_t)nacc);
    {
        std::vector<int32_t> order;
        for (int64_t a = 0; a < nacc; ++a) vmap[(size_t)a] = pc[a].slot >= 0 ? -(1 + pc[a].slot) : pc[a].row;
        for (int64_t g = 0; g < ngroups; ++g) {
            const int32_t a0 = grow[g], a1 = grow[g + 1];
            if (!split) {
                for (int32_t a = a0; a < a1; ++a) wave_of[(size_t)a] = (uint8_t)((a - a0) & (cmfk::BCSR_NW - 1));
                continue;
            }
            order.resize((size_t)(a1 - a0));
            for (int32_t a = a0; a < a1; ++a) order[(size_t)(a - a0)] = a;
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return pc[x].n() > pc[y].n(); });
            int64_t load[cmfk::BCSR_NW] = {0};
            for (int32_t a : order) {
                int best = 0;
                for (int w = 1; w < cmfk::BCSR_NW; ++w)
                    if (load[w] < load[best]) best = w;
                wave_of[(size_t)a] = (uint8_t)best;
                load[best] += pc[a].n();
            }
        }
    }
    // counting sort by key (group, wave, block); inside a key the order (accumulator row, column) is kept
    const int64_t nkey = ngroups * cmfk::BCSR_NW * nblocks;
    std::vector<int64_t> kcnt((size_t)nkey + 1, 0);
    for (int64_t g = 0; g < ngroups; ++g)
        for (int64_t a = grow[g]; a < grow[g + 1]; ++a) {
            const int64_t base = (g * cmfk::BCSR_NW + wave_of[(size_t)a]) * nblocks;
            for (int64_t q = pc[a].q0; q < pc[a].q1; q += pc[a].stride) kcnt[(size_t)(base + indices[q] / B) + 1]++;
        }
    for (int64_t i = 0; i < nkey; ++i) kcnt[i + 1] += kcnt[i];
    // stretches between two re-alignments of an XCD class: about 8k entries of a group (~80 us; 16k: +0.5 ms on X^T U at C5, 32k: +1.7 ms), whole blocks
    const double per_block = (double)nnz / (double)std::max<int64_t>(ngroups * nblocks, 1);
    const double stretch = c->opt_spmm_stretch > 0 ? (double)c->opt_spmm_stretch : 8192.0;
    const int64_t kblk = std::max<int64_t>(1, (int64_t)(stretch / std::max(per_block, 1.0)));
    const int64_t nsync = (nblocks + kblk - 1) / kblk;
    std::vector<int64_t> seg((size_t)(ngroups * cmfk::BCSR_NW * nsync) + 1);
    for (int64_t i = 0; i < ngroups * cmfk::BCSR_NW; ++i)
        for (int64_t st = 0; st < nsync; ++st) seg[(size_t)(i * nsync + st)] = kcnt[(size_t)(i * nblocks + std::min(st * kblk, nblocks))];
    seg[(size_t)(ngroups * cmfk::BCSR_NW * nsync)] = nnz;
    std::vector<cmfk::BcsrEntry> ent((size_t)std::max<int64_t>(nnz, 1));
    for (int64_t g = 0; g < ngroups; ++g)
        for (int64_t a = grow[g]; a < grow[g + 1]; ++a) {
            const int32_t rl = (int32_t)(a - grow[g]);
            const int64_t base = (g * cmfk::BCSR_NW + wave_of[(size_t)a]) * nblocks;
            for (int64_t q = pc[a].q0; q < pc[a].q1; q += pc[a].stride) {
                const int64_t pos = kcnt[(size_t)(base + indices[q] / B)]++;
                ent[pos] = cmfk::BcsrEntry{indices[q], rl, vals[q], 0};
            }
        }
    auto up = [&](void **dptr, const void *src, size_t bytes) -> int {
        CHK(dev_alloc(c, dptr, std::max<size_t>(bytes, 16), false));
        if (bytes) HIPCHK(hipMemcpyAsync(*dptr, src, bytes, hipMemcpyHostToDevice, c->stream));
        return CMF_OK;
    };
    CHK(up((void **)&dst.b_ent, ent.data(), ent.size() * sizeof(cmfk::BcsrEntry)));
    CHK(up((void **)&dst.b_seg, seg.data(), seg.size() * sizeof(int64_t)));
    CHK(up((void **)&dst.b_grow, grow.data(), grow.size() * sizeof(int32_t)));
    CHK(up((void **)&dst.b_vmap, vmap.data(), vmap.size() * sizeof(int32_t)));
    CHK(up((void **)&dst.b_prow, prow.data(), prow.size() * sizeof(int32_t)));
    CHK(up((void **)&dst.b_pfirst, pfirst.data(), pfirst.size() * sizeof(int32_t)));
    CHK(up((void **)&dst.b_pcnt, pcnt.data(), pcnt.size() * sizeof(int32_t)));
    HIPCHK(hipStreamSynchronize(c->stream));
    dst.b_ngroups = (int)ngroups; dst.b_nblocks = (int)nblocks; dst.b_rows_per_group = G; dst.b_nsync = (int)nsync;
    dst.b_nsplit = (int)prow.size(); dst.b_nslots = (int)nslots;
    return CMF_OK;
}

// rows per group: as many as the LDS holds (150 KB of accumulators), then shrunk so that the groups fill whole rounds of the
// persistent grid (782 groups on 256 workgroups would leave a fourth round with 14 of them busy)
static int bcsr_group_rows(const cmf_ctx *c, int64_t rows) {
    const int64_t gmax = std::max<int64_t>(8, std::min<int64_t>(256, (150 * 1024) / (c->kp * 4)));
    const int64_t grid = std::max(8, (c->num_cu / 8) * 8);
    const int64_t rounds = std::max<int64_t>(1, (rows + gmax * grid - 1) / (gmax * grid));
    int64_t g = (rows + rounds * grid - 1) / (rounds * grid);
    g = std::min(gmax, std::max<int64_t>(8, (g + 7) / 8 * 8)); // whole octets: wave w owns the rows with row % 8 == w
    return (int)g;
}

// should this orientation get the blocked regrouping?  (k_pad 64 / 128 / 256; gathered operand = cols x k_pad floats)
static bool want_blocked(const cmf_ctx *c, int64_t cols) {
    if (c->opt_spmm_blocked == 0 || !(c->kp == 64 || c->kp == 128 || c->kp == 256)) return false;
    return c->opt_spmm_blocked == 2 || (double)cols * c->kp * 4.0 > 16.0 * 1024 * 1024;
}

// keep A as CSR and as CSR of A^T (counting sort on the host, O(nnz))
static int set_data_csr_native(cmf_ctx *c, int which, const int64_t *indptr, const int32_t *indices, const double *data, int64_t nnz,
                               int64_t rows, int64_t cols) {
    for (int64_t q = 0; q < nnz; ++q)
        if (indices[q] < 0 || indices[q] >= cols) return fail(CMF_EINVAL, "CSR column index out of range");
    // canonical form (column indices strictly increasing inside a row, duplicates summed -- what scipy's
    // sum_duplicates produces): ||A||^2 of the sparse error expansion is a sum over MERGED entries
    bool canonical = true;
    for (int64_t r = 0; r < rows && canonical; ++r)
        for (int64_t q = indptr[r] + 1; q < indptr[r + 1]; ++q)
            if (indices[q] <= indices[q - 1]) { canonical = false; break; }
    std::vector<int64_t> cptr;
    std::vector<int32_t> cidx;
    std::vector<double> cval;
    if (!canonical) {
        cptr.assign((size_t)rows + 1, 0);
        cidx.reserve((size_t)nnz); cval.reserve((size_t)nnz);
        std::vector<std::pair<int32_t, double>> row;
        for (int64_t r = 0; r < rows; ++r) {
            row.clear();
            for (int64_t q = indptr[r]; q < indptr[r + 1]; ++q) row.emplace_back(indices[q], data[q]);
            std::stable_sort(row.begin(), row.end(), [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &b) { return a.first < b.first; });
            for (size_t i = 0; i < row.size(); ++i) {
                if (!cidx.empty() && (int64_t)cidx.size() > cptr[r] && cidx.back() == row[i].first) cval.back() += row[i].second;
                else { cidx.push_back(row[i].first); cval.push_back(row[i].second); }
            }
            cptr[r + 1] = (int64_t)cidx.size();
        }
        indptr = cptr.data(); indices = cidx.data(); data = cval.data(); nnz = (int64_t)cidx.size();
    }
    std::vector<float> vals((size_t)std::max<int64_t>(nnz, 1));
    double sq = 0.0;
    for (int64_t q = 0; q < nnz; ++q) {
        vals[q] = (float)data[q];
        sq += (double)vals[q] * (double)vals[q];
    }
    CHK(csr_upload(c, c->sp[which][0], indptr, indices, vals.data(), rows, cols, nnz));
    const int64_t B = c->opt_spmm_block_cols > 0 ? c->opt_spmm_block_cols : std::max<int64_t>(256, (3 * 1024 * 1024) / (c->kp * 4)); // 3 MB of the 4 MB L2 (C5, round 4: 1 / 1.5 / 2 / 3 / 4 MB -> 46.4 / 47.6 / 48.5 / 49.1 / 47.7 it/s)
    if (want_blocked(c, cols)) {
        const int rc = bcsr_build_upload(c, c->sp[which][0], indptr, indices, vals.data(), rows, cols, nnz, bcsr_group_rows(c, rows), B);
        if (rc != CMF_OK && rc != CMF_EUNSUPPORTED) return rc;
    }
    // transpose
    std::vector<int64_t> tptr((size_t)cols + 1, 0);
    for (int64_t q = 0; q < nnz; ++q) tptr[indices[q] + 1]++;
    for (int64_t j = 0; j < cols; ++j) tptr[j + 1] += tptr[j];
    std::vector<int32_t> tidx((size_t)std::max<int64_t>(nnz, 1));
    std::vector<float> tval((size_t)std::max<int64_t>(nnz, 1));
    std::vector<int64_t> fill(tptr.begin(), tptr.end() - 1);
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t q = indptr[r]; q < indptr[r + 1]; ++q) {
            const int64_t pos = fill[indices[q]]++;
            tidx[pos] = (int32_t)r;
            tval[pos] = vals[q];
        }
    CHK(csr_upload(c, c->sp[which][1], tptr.data(), tidx.data(), tval.data(), cols, rows, nnz));
    if (want_blocked(c, rows)) {
        const int rc = bcsr_build_upload(c, c->sp[which][1], tptr.data(), tidx.data(), tval.data(), cols, rows, nnz, bcsr_group_rows(c, cols), B);
        if (rc != CMF_OK && rc != CMF_EUNSUPPORTED) return rc;
    }
    c->sparse[which] = true;
    c->sp_sq[which] = sq;
    double sm = 0.0;
    for (int64_t q = 0; q < nnz; ++q) sm += (double)vals[q];
    c->sp_sum[which] = sm;
    return CMF_OK;
}

template <int GL, int CH>
static void launch_spmm(cmf_ctx *c, const CsrView &v, const float *F, float *out, bool accumulate, int width) {
    constexpr int RPW = 64 / GL;
    const int64_t waves = (v.rows + RPW - 1) / RPW;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    if (blocks) hipLaunchKernelGGL((spmm_csr_kernel<GL, CH>), dim3(blocks), dim3(256), 0, c->stream, v, F, width, out, accumulate ? 1 : 0);
}

// out[rows x width] (+)= A F for a device CSR matrix A (width = k_pad unless stated)
// `update`: 0 plain product; 1 / 2 (blocked form only): the output is a Newton factor update -- columns >= k are zeroed and, for 2,
// negatives clamp to 0 (see spmm_blocked_kernel)
static bool spmm_can_update(const cmf_ctx *c, const CsrDev &A) { return A.b_ent != nullptr; }
static int spmm(cmf_ctx *c, const CsrDev &A, const float *F, float *out, int64_t rows_pad, bool accumulate, int width = 0, int update = 0) {
    if (width <= 0) width = c->kp;
    if (update && !(A.b_ent && width == c->kp)) return fail(CMF_EINVAL, "spmm: the update epilogue exists in the blocked form only");
    if (!accumulate && rows_pad > A.rows)
        HIPCHK(hipMemsetAsync(out + A.rows * width, 0, (size_t)(rows_pad - A.rows) * width * sizeof(float), c->stream));
    CsrView v{A.indptr, A.idx, A.val, A.rows};
    Timed tm(c, CMF_K_SPMM, 2.0 * (double)A.nnz * (double)width);
    if (A.b_ent && width == c->kp) { // column-blocked, output-stationary form
        CHK(ensure(c, c->spmm_bar, 8 * 16));
        HIPCHK(hipMemsetAsync(c->spmm_bar.p, 0, 8 * 16, c->stream));
        if (A.b_nslots > 0) CHK(ensure(c, c->spmm_part, (size_t)A.b_nslots * width * sizeof(float)));
        BcsrView bv{A.b_ent, A.b_seg, A.b_grow, A.b_ngroups, A.b_nsync, A.b_vmap, (float *)c->spmm_part.p};
        const unsigned grid = (unsigned)std::max(8, (c->num_cu / 8) * 8);
        const size_t lds = (size_t)A.b_rows_per_group * width * sizeof(float);
#define CMF_SPMMB(V_)                                                                                                     \
    do {                                                                                                                  \
        CHK(allow_big_lds(c, reinterpret_cast<const void *>(&spmm_blocked_kernel<V_>), 156 * 1024));                      \
        hipLaunchKernelGGL((spmm_blocked_kernel<V_>), dim3(grid), dim3(64 * cmfk::BCSR_NW), lds, c->stream, bv, F, out,                   \
                           (accumulate ? 1 : 0) | (update == 2 ? 2 : 0), (unsigned *)c->spmm_bar.p, update ? c->k : 0);                \
        if (A.b_nsplit > 0)                                                                                               \
            hipLaunchKernelGGL((spmm_partial_sum_kernel<V_>), dim3((unsigned)A.b_nsplit), dim3(64), 0, c->stream, (const float *)c->spmm_part.p, \
                               (const int32_t *)A.b_prow, (const int32_t *)A.b_pfirst, (const int32_t *)A.b_pcnt, out,           \
                               (accumulate ? 1 : 0) | (update == 2 ? 2 : 0), update ? c->k : 0);                                  \
    } while (0)
        if (width == 256) CMF_SPMMB(4);
        else if (width == 128) CMF_SPMMB(2);
        else CMF_SPMMB(1);
#undef CMF_SPMMB
        HIPCHK(hipGetLastError());
        return CMF_OK;
    }
    switch (width) {
    case 32: launch_spmm<8, 1>(c, v, F, out, accumulate, width); break;
    case 64: launch_spmm<16, 1>(c, v, F, out, accumulate, width); break;
    case 128: launch_spmm<32, 1>(c, v, F, out, accumulate, width); break;
    case 256: launch_spmm<64, 1>(c, v, F, out, accumulate, width); break;
    case 512: launch_spmm<64, 2>(c, v, F, out, accumulate, width); break;
    case 768: launch_spmm<64, 3>(c, v, F, out, accumulate, width); break;
    case 1024: launch_spmm<64, 4>(c, v, F, out, accumulate, width); break;
    default: return fail(CMF_EUNSUPPORTED, "native CSR path supports operand widths 32..1024 (got %d)", width);
    }
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// expand a natively-sparse data matrix into the dense layout (per-row Newton images need it)
static int need_dense(cmf_ctx *c, int which) {
    float **slot = which == 0 ? &c->X : &c->Y;
    if (*slot) return CMF_OK;
    if (!c->sparse[which]) return fail(CMF_EINVAL, "%s has not been set", which == 0 ? "X" : "Y");
    const int64_t rp = which == 0 ? c->mp : c->dp, cp = which == 0 ? c->dp : c->pp;
    if ((double)rp * (double)cp * 4.0 > 128e9)
        return fail(CMF_EUNSUPPORTED, "this solver configuration needs a dense %lld x %lld image of a sparse input (logit link or "
                                      "sg_sample_ratio < 1); it does not fit", (long long)rp, (long long)cp);
    CHK(dev_alloc(c, (void **)slot, (size_t)rp * cp * sizeof(float)));
    const CsrDev &A = c->sp[which][0];
    CsrView v{A.indptr, A.idx, A.val, A.rows};
    Timed tm(c, CMF_K_ELEMWISE);
    const unsigned blocks = (unsigned)((A.rows + 3) / 4);
    if (blocks) hipLaunchKernelGGL(csr_to_dense_kernel, dim3(blocks), dim3(256), 0, c->stream, v, *slot, cp);
    HIPCHK(hipGetLastError());
    return CMF_OK;
}

// data-times-factor products of the update steps:
//   which = 0: X (m x d)   trans = false: X B (B has d rows) -> m rows ; true: X^T B (B has m rows) -> d rows
//   which = 1: Y (d x p)   trans = false: Y B (B has p rows) -> d rows ; true: Y^T B (B has d rows) -> p rows
// gemm_arith = 1 (cmf_bf16x6.hip.h): out[R x 256] (+)= op(A)[R x K] * B[K x 256] on the bf16 matrix pipe with three planes
// per operand.  The planes of op(A) are built on first use and stay resident (6 bytes per element and orientation).
static int data_times_bf16x6(cmf_ctx *c, int which, bool trans, const float *A, const float *B, float *out, bool accumulate) {
    const int64_t rp = which == 0 ? c->mp : c->dp, cp = which == 0 ? c->dp : c->pp;
    const int64_t R = trans ? cp : rp, K = trans ? rp : cp;   // rows of op(A), reduction length
    const int o = trans ? 1 : 0;
    const size_t plane = (size_t)R * K * sizeof(unsigned short);
    if (!c->bfp_valid[which][o]) {
        CHK(ensure(c, c->bfp[which][o], 3 * plane));
        Timed tm(c, CMF_K_ELEMWISE);
        unsigned short *P = (unsigned short *)c->bfp[which][o].p;
        hipLaunchKernelGGL(bf16x3_split_kernel, dim3(8192), dim3(256), 0, c->stream, A, cp, trans ? 1 : 0, R, K, P, P + (size_t)R * K,
                           P + 2 * (size_t)R * K, 256);
        HIPCHK(hipGetLastError());
        c->bfp_valid[which][o] = true;
    }
    // factor operand: B^T planes [k_pad x K]; their tiles hold k_pad rows (k_pad * 16 entries per k tile)
    const int64_t bn = c->kp;
    const size_t fplane = (size_t)bn * K * sizeof(unsigned short);
    CHK(ensure(c, c->bff, 3 * fplane));
    unsigned short *F = (unsigned short *)c->bff.p;
    {
        Timed tm(c, CMF_K_ELEMWISE);
        hipLaunchKernelGGL(bf16x3_split_kernel, dim3(1024), dim3(256), 0, c->stream, B, (int64_t)c->kp, 1, bn, K, F, F + (size_t)bn * K,
                           F + 2 * (size_t)bn * K, (int)bn);
        HIPCHK(hipGetLastError());
    }
    Bf16x6Args g;
    const unsigned short *P = (const unsigned short *)c->bfp[which][o].p;
    for (int p = 0; p < 3; ++p) { g.A[p] = P + (size_t)p * R * K; g.B[p] = F + (size_t)p * bn * K; }
    // fewer row tiles than CUs: split the reduction over blockIdx.y into slabs (deterministic, summed afterwards)
    const int64_t tiles = R / 256, KT = K / 16;
    int64_t nsplit = 1;
    if (tiles < 192) nsplit = std::min<int64_t>(std::max<int64_t>(1, 256 / tiles), std::max<int64_t>(1, KT / 64));
    int64_t per = (KT + nsplit - 1) / nsplit;
    per += per & 1;                              // whole pairs of k tiles per split
    nsplit = (KT + per - 1) / per;
    g.KT = KT; g.kt_per_split = per; g.accumulate = (accumulate && nsplit == 1) ? 1 : 0;
    g.C = out; g.slab_stride = 0;
    if (nsplit > 1) {
        CHK(ensure(c, c->slabs, (size_t)nsplit * R * bn * sizeof(float)));
        g.C = (float *)c->slabs.p; g.slab_stride = R * bn;
    }
    {
        Timed tm(c, trans ? CMF_K_GEMM_TN : CMF_K_GEMM_NN, 2.0 * (double)R * (double)bn * (double)K);
        const dim3 grid((unsigned)tiles, (unsigned)nsplit);
        if (bn == 256) {
            CHK(allow_big_lds(c, reinterpret_cast<const void *>(&bf16x6_gemm_kernel<4>), BxCfg<4>::LDS_BYTES));
            hipLaunchKernelGGL((bf16x6_gemm_kernel<4>), grid, dim3(512), BxCfg<4>::LDS_BYTES, c->stream, g);
        } else {
            CHK(allow_big_lds(c, reinterpret_cast<const void *>(&bf16x6_gemm_kernel<2>), BxCfg<2>::LDS_BYTES));
            hipLaunchKernelGGL((bf16x6_gemm_kernel<2>), grid, dim3(512), BxCfg<2>::LDS_BYTES, c->stream, g);
        }
        HIPCHK(hipGetLastError());
    }
    if (nsplit > 1) CHK(sum_slabs(c, out, (const float *)c->slabs.p, R * bn, (int)nsplit, g.slab_stride, accumulate));
    return CMF_OK;
}

static int data_times(cmf_ctx *c, int which, bool trans, const float *B, float *out, bool accumulate = false, SlabRef *defer = nullptr) {
    if (defer) *defer = SlabRef();
    const int64_t rp = which == 0 ? c->mp : c->dp, cp = which == 0 ? c->dp : c->pp;
    if (c->sparse[which] && !(which == 0 ? c->X : c->Y)) return spmm(c, c->sp[which][trans ? 1 : 0], B, out, trans ? cp : rp, accumulate);
    const float *A = which == 0 ? c->X : c->Y;
    if (!A) return fail(CMF_EINVAL, "%s has not been set", which == 0 ? "X" : "Y");
    // optional arithmetic (k_pad = 256 or 128; tiny operands stay on the fp32 kernels)
    if (c->opt_arith == 1 && (c->kp == 256 || c->kp == 128) && (trans ? cp : rp) >= (int64_t)c->opt_arith_min_tiles * 256) return data_times_bf16x6(c, which, trans, A, B, out, accumulate);
    if (!trans) return gemm(c, MODE_NN, A, cp, B, c->kp, out, rp, c->kp, cp, accumulate, nullptr, defer);
    return gemm(c, MODE_TN, A, cp, B, c->kp, out, cp, c->kp, rp, accumulate, nullptr, defer);
}

static bool have_data(const cmf_ctx *c, int which) { return (which == 0 ? c->X : c->Y) != nullptr || c->sparse[which]; }

template <int GL, int CH>
static void launch_sddmm(cmf_ctx *c, const CsrView &v, const float *L, const float *R, double *partials, unsigned blocks, int logit) {
    hipLaunchKernelGGL((sddmm_cross_kernel<GL, CH>), dim3(blocks), dim3(256), 0, c->stream, v, L, R, c->kp, partials, logit);
}

// native CSR target:
//   linear  ||A - L R^T||^2          = ||A||^2 - 2 sum_nnz a_ij (l_i . r_j) + <L^T L, R^T R>       (sklearn's expansion, cmf_solvers.py:40)
//   logit   ||A - sigmoid(L R^T)||^2 = ||A||^2 - 2 sum_nnz a_ij sigmoid(l_i . r_j) + sum_all sigmoid(l_i . r_j)^2   (:42)
// the last term of the logit form is the dense NT pass with zero targets: compute without any m x d image
static int sparse_residual_sq(cmf_ctx *c, int which, double *dev_out, int link) {
    const CsrDev &A = c->sp[which][0];
    const float *L = which == 0 ? c->F[CMF_U] : c->F[CMF_V];
    const float *R = which == 0 ? c->F[CMF_V] : c->F[CMF_Z];
    const int64_t lrows = which == 0 ? c->mp : c->dp, rrows = which == 0 ? c->dp : c->pp;
    const bool logit = link == CMF_LINK_LOGIT;
    if (logit) {
        NtOut o; o.T = nullptr; o.ldt = 0; o.sq = c->dscalar + 5; o.link = link;
        CHK(gemm_nt(c, L, lrows, which == 0 ? c->m : c->d, R, rrows, which == 0 ? c->d : c->p, o));
    } else {
        CHK(gemm(c, MODE_TN, L, c->kp, L, c->kp, c->G, c->kp, c->kp, lrows));
        CHK(gemm(c, MODE_TN, R, c->kp, R, c->kp, c->G2, c->kp, c->kp, rrows));
    }
    const int gl = std::min(64, c->kp / 4);
    const int rpw = 64 / gl;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, ((A.rows + rpw - 1) / rpw + 3) / 4);
    CHK(ensure(c, c->dpart, (size_t)blocks * sizeof(double)));
    CsrView v{A.indptr, A.idx, A.val, A.rows};
    {
        Timed tm(c, CMF_K_SPMM, 2.0 * (double)A.nnz * (double)c->kp);
        switch (c->kp) {
        case 32: launch_sddmm<8, 1>(c, v, L, R, (double *)c->dpart.p, blocks, logit ? 1 : 0); break;
        case 64: launch_sddmm<16, 1>(c, v, L, R, (double *)c->dpart.p, blocks, logit ? 1 : 0); break;
        case 128: launch_sddmm<32, 1>(c, v, L, R, (double *)c->dpart.p, blocks, logit ? 1 : 0); break;
        case 256: launch_sddmm<64, 1>(c, v, L, R, (double *)c->dpart.p, blocks, logit ? 1 : 0); break;
        case 512: launch_sddmm<64, 2>(c, v, L, R, (double *)c->dpart.p, blocks, logit ? 1 : 0); break;
        case 768: launch_sddmm<64, 3>(c, v, L, R, (double *)c->dpart.p, blocks, logit ? 1 : 0); break;
        case 1024: launch_sddmm<64, 4>(c, v, L, R, (double *)c->dpart.p, blocks, logit ? 1 : 0); break;
        default: return fail(CMF_EUNSUPPORTED, "native CSR path supports n_components <= 1024");
        }
        HIPCHK(hipGetLastError());
    }
    Timed tm(c, CMF_K_ELEMWISE);
    hipLaunchKernelGGL(sum_doubles_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)c->dpart.p, (int64_t)blocks, c->dscalar + 4);
    if (!logit) hipLaunchKernelGGL(frob_inner_kernel, dim3(1), dim3(256), 0, c->stream, (const float *)c->G, (const float *)c->G2, c->kp * c->kp, c->dscalar + 5);
    HIPCHK(hipGetLastError());
    double h[2];
    HIPCHK(hipMemcpyAsync(h, c->dscalar + 4, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    double r2 = c->sp_sq[which] - 2.0 * h[0] + h[1];
    if (r2 < 0) r2 = 0; // rounding of the expansion near a perfect fit
    HIPCHK(hipMemcpyAsync(dev_out, &r2, sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return CMF_OK;
}

