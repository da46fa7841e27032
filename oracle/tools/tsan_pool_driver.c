/*
 * tsan_pool_driver.c -- TEST INFRASTRUCTURE: drives the CPU checker's worker pool (sc_oracle.c: sco_db_distance_batch_mt, the all-core
 * CPU baseline of bench.py) under ThreadSanitizer (`make sanitize`): pools of several sizes created, reused and re-made, calls of
 * different lengths, two databases sharing the pool, and the results compared with the single-threaded evaluation.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sc_oracle.h"

int main(void)
{
    sco_config c; sco_default_config(&c);
    const int R = c.num_ring, S = c.num_sector, N = 160;
    sco_db *a = sco_db_create(&c), *b = sco_db_create(&c);
    float *v = (float *)malloc(sizeof(float) * (size_t)R * S);
    unsigned long long s = 12345;
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < R * S; j++) { s = s * 6364136223846793005ull + 1442695040888963407ull; v[j] = (s >> 40) % 5 ? (float)((s >> 33) % 1200) * 0.01f : 0.0f; }
        sco_db_save_wire(a, v, 0, i);
        if (i % 2) sco_db_save_wire(b, v, 1, i);
    }
    int cand[160]; double d1[160], dm[160]; int s1[160], sm[160];
    for (int i = 0; i < N; i++) cand[i] = i;
    int bad = 0;
    for (int rep = 0; rep < 6; rep++) {
        static const int threads[6] = {4, 4, 2, 8, 3, 8};
        const int n = rep % 2 ? 37 : N - 1;
        sco_db_distance_batch(a, N - 1, cand, n, d1, s1, 0);
        sco_db_distance_batch_mt(a, N - 1, cand, n, dm, sm, 0, threads[rep]);
        for (int i = 0; i < n; i++) bad += memcmp(&d1[i], &dm[i], sizeof(double)) != 0 || s1[i] != sm[i];
        const int nb = sco_db_size(b);
        sco_db_distance_batch(b, nb - 1, cand, nb - 1, d1, s1, 1);
        sco_db_distance_batch_mt(b, nb - 1, cand, nb - 1, dm, sm, 1, threads[rep]);
        for (int i = 0; i < nb - 1; i++) bad += memcmp(&d1[i], &dm[i], sizeof(double)) != 0 || s1[i] != sm[i];
    }
    sco_db_destroy(a); sco_db_destroy(b); free(v);
    printf("tsan pool driver: %d differences between the pool and the single thread\n", bad);
    return bad ? 1 : 0;
}
