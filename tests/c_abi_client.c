/* A plain-C client of libsslam_hip.so: proves the boundary is language-neutral (no Python, no
 * torch, no C++ types).  Built and run by tests/test_abi_c_client.py:
 *     gcc -std=c99 -I include tests/c_abi_client.c -L opencv-simpleslam_amd/lib -lsslam_hip -lm
 * Checks (a) the BA residual of a point on the optical axis through an identity pose (closed form),
 * (b) a device LM solve that must not increase the cost, (c) the RANSAC filter on exact two-view
 * correspondences with planted outliers. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "sslam_hip.h"

#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "FAIL %s: %s\n", #call, sslam_last_error()); return 1; } } while (0)

int main(void) {
    sslam_ctx* ctx = NULL;
    int ndev = 0;
    CHECK(sslam_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no GPU\n"); return 2; }
    CHECK(sslam_ctx_create(0, NULL, &ctx));

    /* (a) identity pose, points (0,0,5) and (1,2,4): u = fx x/z + cx, v = fy y/z + cy */
    const double q[4] = {0, 0, 0, 1}, t[3] = {0, 0, 0}, intr[4] = {700, 710, 600, 180};
    const double X[6] = {0, 0, 5, 1, 2, 4};
    const int32_t pi[2] = {0, 0}, xi[2] = {0, 1};
    const double uv[4] = {600, 180, 770, 530};
    double r[4], Jq[16], Jt[12], JX[12];
    CHECK(sslam_ba_residual_jacobian_host(ctx, 2, pi, xi, uv, 1, q, t, 2, X, intr, r, Jq, Jt, JX));
    const double want[4] = {0.0, 0.0, 700.0 * 0.25 + 600 - 770, 710.0 * 0.5 + 180 - 530};
    for (int i = 0; i < 4; ++i)
        if (fabs(r[i] - want[i]) > 1e-9) { fprintf(stderr, "residual %d: %g != %g\n", i, r[i], want[i]); return 1; }
    if (fabs(Jt[0] - 700.0 / 5) > 1e-9 || fabs(JX[0] - 700.0 / 5) > 1e-9) { fprintf(stderr, "jacobian\n"); return 1; }

    /* (b) two poses (first fixed), 40 points, noisy start: the LM must lower the cost */
    enum { NP = 40, NO = 80 };
    double q2[8] = {0, 0, 0, 1, 0, 0.02, 0, 0.9998}, t2[6] = {0, 0, 0, -0.5, 0.02, 0.03}, Xs[3 * NP], uv2[2 * NO];
    int32_t pj[NO], xj[NO];
    unsigned char fixed[2] = {1, 0};
    srand(1);
    for (int j = 0; j < NP; ++j) {
        Xs[3 * j] = (rand() % 2000) / 250.0 - 4; Xs[3 * j + 1] = (rand() % 1000) / 250.0 - 2; Xs[3 * j + 2] = 8 + (rand() % 1000) / 100.0;
    }
    /* observations generated from the true geometry (second camera translated by -0.5 in x), then the
       second pose starts off by the small rotation above */
    for (int j = 0; j < NP; ++j)
        for (int c = 0; c < 2; ++c) {
            const int o = 2 * j + c;
            const double x = Xs[3 * j] + (c ? -0.5 : 0.0), y = Xs[3 * j + 1], z = Xs[3 * j + 2];
            pj[o] = c; xj[o] = j;
            uv2[2 * o] = intr[0] * x / z + intr[2]; uv2[2 * o + 1] = intr[1] * y / z + intr[3];
        }
    double summary[8];
    CHECK(sslam_ba_solve_host(ctx, NO, pj, xj, uv2, 2, q2, t2, fixed, NP, Xs, intr, 15, 2.0, 0, summary));
    if (!(summary[3] < 0.05 * summary[2]) || summary[1] < 1) { fprintf(stderr, "LM: cost %g -> %g\n", summary[2], summary[3]); return 1; }

    /* (c) RANSAC: 60 exact correspondences of a sideways translation + 20 planted outliers */
    enum { NM = 80 };
    float p1[2 * NM], p2[2 * NM];
    unsigned char mask[NM];
    for (int i = 0; i < NM; ++i) {
        const double x = (rand() % 2000) / 100.0 - 10, y = (rand() % 800) / 100.0 - 4, z = 6 + (rand() % 3000) / 100.0;
        p1[2 * i] = (float)(700 * x / z + 600); p1[2 * i + 1] = (float)(700 * y / z + 180);
        p2[2 * i] = (float)(700 * (x - 0.8) / z + 600); p2[2 * i + 1] = (float)(700 * (y + 0.1) / (z - 0.3) + 180);
        if (i >= 60) { p2[2 * i] = (float)(rand() % 1200); p2[2 * i + 1] = (float)(rand() % 370); }
    }
    double F[9];
    int info[4];
    CHECK(sslam_fmat_ransac_host(ctx, NM, p1, p2, 1.0, 0.99, 1000, mask, F, info));
    int in_true = 0, in_out = 0;
    for (int i = 0; i < NM; ++i) { if (i < 60) in_true += mask[i]; else in_out += mask[i]; }
    if (in_true < 50 || in_out > 4) { fprintf(stderr, "RANSAC: %d true inliers, %d outliers kept\n", in_true, in_out); return 1; }

    CHECK(sslam_ctx_destroy(ctx));
    printf("c client ok: residuals exact, LM %g -> %g in %d steps, RANSAC kept %d/60 + %d/20\n", summary[2], summary[3],
           (int)summary[1], in_true, in_out);
    return 0;
}
