/* sanitize_main.c -- runs the oracle's entry points on the reference's cat pair under AddressSanitizer + UBSan
 * (oracle/Makefile target `sanitize`; tests/test_oracle_pins.py builds and runs it on the CPU).  Test infrastructure, like
 * everything under oracle/.  Usage: oracle_sanitize cat.pcd cat_out.pcd */
#include "symmicp_oracle.c"

#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s src.pcd tgt.pcd\n", argv[0]); return 64; }
    long ns = orc_pcd_read(argv[1], NULL, NULL, 0, NULL), nt = orc_pcd_read(argv[2], NULL, NULL, 0, NULL);
    if (ns <= 0 || nt <= 0) { fprintf(stderr, "cannot read the clouds\n"); return 2; }
    float *src = malloc(sizeof(float) * 3 * (size_t)ns), *tgt = malloc(sizeof(float) * 3 * (size_t)nt);
    float *sn = malloc(sizeof(float) * 3 * (size_t)ns), *tn = malloc(sizeof(float) * 3 * (size_t)nt);
    if (orc_pcd_read(argv[1], src, NULL, (size_t)ns, NULL) != ns || orc_pcd_read(argv[2], tgt, NULL, (size_t)nt, NULL) != nt) return 2;
    const float vp[3] = {0.f, 0.f, 0.f};
    if (orc_normals_knn(src, (size_t)ns, 10, vp, sn, NULL) != 0 || orc_normals_knn(tgt, (size_t)nt, 10, vp, tn, NULL) != 0) return 3;
    int bad = 0;
    const int modes[3] = {ORC_MODE_QUIRKS, ORC_MODE_PAPER, ORC_MODE_P2P};
    const int corrs[3] = {ORC_CORR_IDENTITY, ORC_CORR_BRUTE, ORC_CORR_GRID};
    for (int m = 0; m < 3; m++)
        for (int c = 0; c < 3; c++) {
            orc_config cfg;
            orc_config_default(&cfg);
            cfg.mode = modes[m]; cfg.corr = corrs[c]; cfg.max_iters = 6;
            cfg.apply = (modes[m] == ORC_MODE_QUIRKS) ? ORC_APPLY_INCREMENTAL : ORC_APPLY_CUMULATIVE;
            orc_result r;
            orc_align(&cfg, src, sn, (size_t)ns, tgt, tn, (size_t)nt, NULL, &r);
            printf("mode %d corr %d: status %d iters %d diff %.2f -> %.2f\n", modes[m], corrs[c], r.status, r.iters, r.diff_initial, r.diff_final);
            if (r.iters < 1) bad++;
        }
    /* the literal N x 3 SVD route of func.cpp:64-73 and the ragged / tiny sizes of the NN searches */
    float pb[3], qb[3], a[3], t[3];
    if (orc_solve_quirks_literal(src, sn, tgt, tn, (size_t)ns, pb, qb, a, t) != 0) bad++;
    float X[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (size_t n = 1; n <= 65; n += 16) {
        int32_t *idx = malloc(sizeof(int32_t) * n), *idx2 = malloc(sizeof(int32_t) * n);
        float *d2 = malloc(sizeof(float) * n), *d22 = malloc(sizeof(float) * n);
        orc_grid *g = orc_grid_build(tgt, n, 2.0f);
        orc_nn_brute(X, src, n, tgt, n, idx, d2);
        orc_nn_grid(g, X, src, n, tgt, idx2, d22);
        for (size_t k = 0; k < n; k++) if (idx[k] != idx2[k] || d2[k] != d22[k]) bad++;
        orc_grid_free(g);
        free(idx); free(idx2); free(d2); free(d22);
    }
    free(src); free(tgt); free(sn); free(tn);
    printf("oracle_sanitize: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
