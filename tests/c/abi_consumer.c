/* A host that is neither Python nor C++: plain C99 against include/cmfhip.h and libcmfhip.so.
 *
 * Reads a problem file (int64 m, d, p, k, steps; double l1, l2; then X (m x d), Y (d x p), U, V, Z row-major float64), runs `steps`
 * calls of cmf_mu_step -- the body of MUSolver.update_step, pycmf/cmf_solvers.py:248-263 -- and writes U, V, Z and the two squared
 * residuals (pycmf/cmf_solvers.py:36-42) to the output file.  tests/test_gpu_c_consumer.py builds it with gcc, feeds it the golden
 * fixture g2 and compares with the reference's results; tests/test_host_logic.py compiles and links it on the CPU box (the header is
 * valid C, every entry point it uses is exported) and checks that without a GPU it fails loudly, not silently.
 *
 *   gcc -std=c99 -Wall -Wextra -Werror -I include tests/c/abi_consumer.c -L pycmf_amd -lcmfhip -Wl,-rpath,$PWD/pycmf_amd -o abi_consumer */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "cmfhip.h"

#define TRY(call)                                                                      \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != CMF_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cmf_last_error());           \
            return 3;                                                                  \
        }                                                                              \
    } while (0)

static double *read_doubles(FILE *f, size_t n) {
    double *p = (double *)malloc((n ? n : 1) * sizeof(double));
    if (!p || fread(p, sizeof(double), n, f) != n) {
        fprintf(stderr, "short problem file\n");
        exit(2);
    }
    return p;
}

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int64_t hdr[5];
    double reg[2];
    if (fread(hdr, sizeof(int64_t), 5, f) != 5 || fread(reg, sizeof(double), 2, f) != 2) { fprintf(stderr, "bad header\n"); return 2; }
    const int64_t m = hdr[0], d = hdr[1], p = hdr[2], k = hdr[3], steps = hdr[4];
    double *X = read_doubles(f, (size_t)(m * d)), *Y = read_doubles(f, (size_t)(d * p));
    double *U = read_doubles(f, (size_t)(m * k)), *V = read_doubles(f, (size_t)(d * k)), *Z = read_doubles(f, (size_t)(p * k));
    fclose(f);

    int ndev = 0;
    TRY(cmf_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no GPU visible: libcmfhip has no CPU path\n"); return 4; }
    cmf_ctx *ctx = NULL;
    TRY(cmf_ctx_create(&ctx, 0, NULL));
    TRY(cmf_set_problem(ctx, m, d, p, (int)k));
    TRY(cmf_set_data_f64(ctx, 0, X, d, 1));
    TRY(cmf_set_data_f64(ctx, 1, Y, p, 1));
    TRY(cmf_set_factor_f64(ctx, CMF_U, U, k, 1));
    TRY(cmf_set_factor_f64(ctx, CMF_V, V, k, 1));
    TRY(cmf_set_factor_f64(ctx, CMF_Z, Z, k, 1));
    for (int64_t s = 0; s < steps; ++s) TRY(cmf_mu_step(ctx, reg[0], reg[1], CMF_UPD_U | CMF_UPD_V | CMF_UPD_Z));
    TRY(cmf_get_factor_f64(ctx, CMF_U, U, k, 1));
    TRY(cmf_get_factor_f64(ctx, CMF_V, V, k, 1));
    TRY(cmf_get_factor_f64(ctx, CMF_Z, Z, k, 1));
    double err[2] = {0.0, 0.0};
    TRY(cmf_residual_sq(ctx, CMF_LINK_LINEAR, CMF_LINK_LINEAR, &err[0], &err[1]));
    TRY(cmf_ctx_destroy(ctx));

    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 2; }
    fwrite(U, sizeof(double), (size_t)(m * k), f);
    fwrite(V, sizeof(double), (size_t)(d * k), f);
    fwrite(Z, sizeof(double), (size_t)(p * k), f);
    fwrite(err, sizeof(double), 2, f);
    fclose(f);
    free(X); free(Y); free(U); free(V); free(Z);
    printf("abi_consumer: %lld MU steps on %lld x %lld / %lld x %lld, k = %lld: ||X - U V^T||^2 = %.6g, ||Y - V Z^T||^2 = %.6g\n",
           (long long)steps, (long long)m, (long long)d, (long long)d, (long long)p, (long long)k, err[0], err[1]);
    return 0;
}
