// Standalone check + timing of chol_solve_mfma_kernel (pycmf_amd/csrc/cmf_chol_mfma.hip.h) against a float64 host solve and
// against the rank-1 register kernel it replaces (chol_solve_kernel<16>).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pycmf_amd/csrc -o tools/ab/chol_mfma_test tools/chol_mfma_test.hip
//   tools/ab/chol_mfma_test [nmat_timing]
#include "cmf_kernels.hip.h"
#include "cmf_eigen.hip.h"
#include "cmf_chol_mfma.hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)

static void host_solve(const std::vector<double> &H, int n, const std::vector<double> &g, std::vector<double> &x, double *lmin_piv) {
    std::vector<double> L((size_t)n * n, 0.0);
    double pm = 1e300;
    for (int j = 0; j < n; ++j) {
        double s = H[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) s -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        pm = std::min(pm, s);
        const double d = std::sqrt(s > 0 ? s : 1e-300);
        L[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = H[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = v / d;
        }
    }
    if (lmin_piv) *lmin_piv = pm;
    std::vector<double> y(n);
    for (int i = 0; i < n; ++i) {
        double v = g[i];
        for (int k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * y[k];
        y[i] = v / L[(size_t)i * n + i];
    }
    x.assign(n, 0.0);
    for (int i = n - 1; i >= 0; --i) {
        double v = y[i];
        for (int k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * x[k];
        x[i] = v / L[(size_t)i * n + i];
    }
}

int main(int argc, char **argv) {
    const int kp = 256;
    const int nt = argc > 1 ? atoi(argv[1]) : 8192;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&cmfk::chol_solve_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                           (int)cmfk::CholMfma::LDS_BYTES));
    int bad = 0;
    for (int n : {256, 200, 129, 97, 33}) {
        const int nm = 6;
        std::vector<float> H((size_t)nm * kp * kp, 0.f), g((size_t)nm * kp, 0.f);
        std::vector<std::vector<double>> Hd(nm), gd(nm);
        for (int mtx = 0; mtx < nm; ++mtx) {
            // H = B^T B / s + delta I with s samples (s < n for the last two: rank-deficient Gram + ridge)
            const int s = (mtx < 4) ? 2 * n : n / 2;
            const double delta = (mtx % 2) ? 0.35 : 3.0;     // matrices 1, 3, 5 sit close to pert = 0.3 (0.35 - 0.3 = 0.05 above)
            std::vector<float> B((size_t)s * n);
            for (auto &v : B) v = nd(rng) * 0.3f;
            Hd[mtx].assign((size_t)n * n, 0.0);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j) {
                    double acc = 0;
                    for (int q = 0; q < s; ++q) acc += (double)B[(size_t)q * n + i] * B[(size_t)q * n + j];
                    acc += (i == j) ? delta : 0.0;
                    const float f = (float)acc;
                    H[(size_t)mtx * kp * kp + (size_t)i * kp + j] = f;
                    H[(size_t)mtx * kp * kp + (size_t)j * kp + i] = f;
                    Hd[mtx][(size_t)i * n + j] = Hd[mtx][(size_t)j * n + i] = (double)f;
                }
            gd[mtx].resize(n);
            for (int i = 0; i < n; ++i) { const float v = nd(rng); g[(size_t)mtx * kp + i] = v; gd[mtx][i] = v; }
        }
        float *dH, *dg, *ds, *dc;
        int *df;
        CK(hipMalloc(&dH, H.size() * 4)); CK(hipMalloc(&dg, g.size() * 4)); CK(hipMalloc(&ds, g.size() * 4)); CK(hipMalloc(&df, nm * 4)); CK(hipMalloc(&dc, nm * 4));
        CK(hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice));
        for (float pert : {0.3f, 0.5f}) {
            CK(hipMemset(ds, 0xff, g.size() * 4));
            CK(hipMemset(df, 0xff, nm * 4));
            hipLaunchKernelGGL(cmfk::chol_solve_mfma_kernel, dim3(nm), dim3(256), cmfk::CholMfma::LDS_BYTES, 0, dH, dg, ds, df, n, kp, (int64_t)kp * kp, pert, nm,
                               (const int *)nullptr, (const int *)nullptr, 1, 0, dc);
            CK(hipDeviceSynchronize());
            std::vector<float> st(g.size()), ce(nm);
            std::vector<int> fl(nm);
            CK(hipMemcpy(st.data(), ds, st.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(fl.data(), df, nm * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ce.data(), dc, nm * 4, hipMemcpyDeviceToHost));
            for (int mtx = 0; mtx < nm; ++mtx) {
                // expected flag: lambda_min(H) >= pert  <=>  Cholesky of H - pert I succeeds
                std::vector<double> Hs = Hd[mtx];
                for (int i = 0; i < n; ++i) Hs[(size_t)i * n + i] -= pert;
                std::vector<double> x;
                double pm;
                host_solve(Hs, n, gd[mtx], x, &pm);
                const int want_flag = pm > 0 ? 0 : 1;
                double pm0;
                host_solve(Hd[mtx], n, gd[mtx], x, &pm0);
                double err = 0, ref = 0;
                for (int i = 0; i < n; ++i) { err = std::max(err, std::fabs((double)st[(size_t)mtx * kp + i] - x[i])); ref = std::max(ref, std::fabs(x[i])); }
                double dm = 0;
                for (int i = 0; i < n; ++i) dm = std::max(dm, Hd[mtx][(size_t)i * n + i]);
                const bool okflag = fl[mtx] == want_flag || std::fabs(pm) < 1e-3; // a pivot within rounding of the threshold may go either way
                bool ok = okflag;
                if (fl[mtx] == 0) ok = ok && err <= 2e-4 * ref && std::fabs(ce[mtx] - dm / pm0) <= 2e-3 * dm / pm0;
                for (int i = n; i < kp && fl[mtx] == 0; ++i) ok = ok && st[(size_t)mtx * kp + i] == 0.f;
                printf("n %3d pert %.1f mat %d: flag %d (want %d, min pivot of H - pert I %.3e)  max|x - x64| %.2e of %.2e  condest %.3e (host %.3e)  %s\n", n, pert, mtx,
                       fl[mtx], want_flag, pm, err, ref, ce[mtx], dm / pm0, ok ? "ok" : "FAIL");
                bad += ok ? 0 : 1;
            }
        }
        CK(hipFree(dH)); CK(hipFree(dg)); CK(hipFree(ds)); CK(hipFree(df)); CK(hipFree(dc));
    }
    // ---- timing: nt well-conditioned 256 x 256 systems (the same matrix image replicated: values do not change the time)
    {
        const int n = 256;
        std::vector<float> H((size_t)kp * kp, 0.f);
        std::vector<float> B((size_t)2 * n * n);
        for (auto &v : B) v = nd(rng) * 0.3f;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                double acc = (i == j) ? 1.0 : 0.0;
                for (int q = 0; q < 2 * n; ++q) acc += (double)B[(size_t)q * n + i] * B[(size_t)q * n + j];
                H[(size_t)i * kp + j] = H[(size_t)j * kp + i] = (float)acc;
            }
        float *dH, *dg, *ds, *dc;
        int *df;
        CK(hipMalloc(&dH, (size_t)nt * kp * kp * 4)); CK(hipMalloc(&dg, (size_t)nt * kp * 4)); CK(hipMalloc(&ds, (size_t)nt * kp * 4));
        CK(hipMalloc(&df, (size_t)nt * 4)); CK(hipMalloc(&dc, (size_t)nt * 4));
        for (int i = 0; i < nt; ++i) CK(hipMemcpy(dH + (size_t)i * kp * kp, H.data(), H.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dg, 0, (size_t)nt * kp * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int variant = 0; variant < 4; ++variant) {
            // 0: new kernel, threshold test + solve; 1: new kernel, certified (solve only); 2, 3: the same with chol_solve_kernel<16>
            int *cert = nullptr;
            if (variant & 1) { CK(hipMalloc(&cert, 2 * sizeof(int))); CK(hipMemset(cert, 0, 2 * sizeof(int))); }
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (variant < 2)
                    hipLaunchKernelGGL(cmfk::chol_solve_mfma_kernel, dim3(nt), dim3(256), cmfk::CholMfma::LDS_BYTES, 0, dH, dg, ds, df, n, kp, (int64_t)kp * kp, 0.2f, nt,
                                       (const int *)nullptr, (const int *)cert, nt, 0, dc);
                else
                    hipLaunchKernelGGL((cmfk::chol_solve_kernel<16>), dim3(nt), dim3(256), 0, 0, dH, dg, ds, df, n, kp, (int64_t)kp * kp, 0.2f, nt, 0, (const int *)nullptr, 1,
                                       (const int *)cert, nt, 0, dc);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
#ifdef CMF_DIAG_BUILD
            if (variant == 1) {
                unsigned long long pr[128];
                CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(cmfk::cm_prof), sizeof pr));
                const double tick = 1.0; // s_memtime: shader-clock cycles here
                printf("profile of one workgroup (cycles): load %.0f", (pr[1] - pr[0]) * tick);
                unsigned long long prev = pr[1];
                for (int J = 0; J < 8; ++J) {
                    printf(" | J%d chain %.0f panel %.0f trail %.0f", J, (pr[8 + 3 * J] - prev) * tick, (pr[9 + 3 * J] - pr[8 + 3 * J]) * tick, (pr[10 + 3 * J] - pr[9 + 3 * J]) * tick);
                    prev = pr[10 + 3 * J];
                }
                printf(" | to-solve %.0f forward %.0f backward %.0f total %.0f\n", (pr[2] - prev) * tick, (pr[3] - pr[2]) * tick, (pr[4] - pr[3]) * tick, (pr[4] - pr[0]) * tick);
            }
#endif
            printf("%s, %s: %d systems of 256 in %.3f ms = %.3f us each (chip-wide)\n", variant < 2 ? "chol_solve_mfma_kernel" : "chol_solve_kernel<16>   ",
                   (variant & 1) ? "certified (one factorisation)" : "threshold test + solve (two factorisations)", nt, best, best * 1e3 / nt);
            if (cert) CK(hipFree(cert));
        }
    }
    printf(bad ? "FAILURES: %d\n" : "all checks passed\n", bad);
    return bad ? 1 : 0;
}
