/*
 * atanf_exhaustive.c -- TEST INFRASTRUCTURE (oracle/): pins the restated float atan to THIS platform's libm.
 *
 * The reference's xy2theta (include/descriptor.h:1352-1374) calls std::atan(float) = the platform's atanf.  The checker
 * (oracle/sc_oracle.c: sco_atanf) and the device (csrc/device_common.hpp: atanf_glibc) restate glibc 2.35's float atanf
 * (sysdeps/ieee754/flt-32/s_atanf.c: fdlibm's argument reduction + odd/even split polynomial, fp32 operations only, no
 * contraction; `objdump -d libm.so.6` shows mulss/addss/subss/divss only and `readelf -s` shows a plain FUNC, no IFUNC
 * variant).  This program evaluates BOTH over all 2^32 float bit patterns, reports the number of differing results (NaN
 * payloads compared as "both NaN") and writes one order-independent checksum per block of 2^24 consecutive bit patterns -- of the RESTATEMENT's
 * results, which equal libm's when the difference count is 0 -- as JSON: tests/golden/atanf_blocks.json.  The GPU test
 * evaluates all 2^32 inputs on the device against those 256 checksums.
 *
 *   gcc -O2 -ffp-contract=off -fno-fast-math -pthread -I.. atanf_exhaustive.c ../sc_oracle.c ../icp_oracle.c ../iris_oracle.c -lm
 *   (the Makefile's `atanf-golden` target does it)
 */
#include <gnu/libc-version.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sc_oracle.h"

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* checksum of a block: sum mod 2^64 over its inputs of splitmix64((input bits << 32) | canonical result bits), every NaN result
 * counted as 0x7fc00000 -- order independent, so the device can form it with one atomic add per wave */
static inline uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

typedef struct { int first_block, last_block; uint64_t sums[256]; uint64_t diffs; uint32_t first_diff; int have_diff; } job_t;

static void *run(void *arg)
{
    job_t *j = (job_t *)arg;
    for (int blk = j->first_block; blk < j->last_block; blk++) {
        uint64_t h = 0;
        for (uint32_t i = 0; i < (1u << 24); i++) {
            const uint32_t bits = ((uint32_t)blk << 24) | i;
            const float x = u2f(bits);
            const float a = sco_atanf_glibc(x);
            const float b = atanf(x);
            uint32_t ua = f2u(a), ub = f2u(b);
            if (a != a) ua = 0x7fc00000u;
            if (b != b) ub = 0x7fc00000u;
            if (ua != ub) { if (!j->have_diff) { j->first_diff = bits; j->have_diff = 1; } j->diffs++; }
            h += mix64(((uint64_t)bits << 32) | ua);
        }
        j->sums[blk] = h;
    }
    return NULL;
}

int main(int argc, char **argv)
{
    const char *out = argc > 1 ? argv[1] : "atanf_blocks.json";
    int nthreads = argc > 2 ? atoi(argv[2]) : 8;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    static job_t jobs[64];
    pthread_t th[64];
    for (int t = 0; t < nthreads; t++) {
        jobs[t].first_block = 256 * t / nthreads; jobs[t].last_block = 256 * (t + 1) / nthreads;
        pthread_create(&th[t], NULL, run, &jobs[t]);
    }
    uint64_t sums[256]; uint64_t diffs = 0; int have = 0; uint32_t first = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        for (int b = jobs[t].first_block; b < jobs[t].last_block; b++) sums[b] = jobs[t].sums[b];
        diffs += jobs[t].diffs;
        if (jobs[t].have_diff && !have) { have = 1; first = jobs[t].first_diff; }
    }
    printf("inputs 4294967296, results differing from libm atanf: %llu", (unsigned long long)diffs);
    if (have) printf(" (first at bits 0x%08x: restated %.9g libm %.9g)", first, sco_atanf_glibc(u2f(first)), atanf(u2f(first)));
    printf("\n");
    FILE *f = fopen(out, "w");
    if (!f) { perror(out); return 2; }
    fprintf(f, "{\n  \"what\": \"checksums of atanf over blocks of 2^24 consecutive float bit patterns (block b = bits b<<24 .. (b<<24)+2^24-1): sum mod 2^64 of "
               "splitmix64((bits << 32) | result bits), NaN results counted as 0x7fc00000, from oracle/tools/atanf_exhaustive.c in the build container\",\n");
    fprintf(f, "  \"libm\": \"glibc %s atanf (plain FUNC, no IFUNC variant)\",\n  \"differences_vs_libm\": %llu,\n  \"blocks\": [",
            gnu_get_libc_version(), (unsigned long long)diffs);
    for (int b = 0; b < 256; b++) fprintf(f, "%s\"%016llx\"", b ? ", " : "", (unsigned long long)sums[b]);
    fprintf(f, "]\n}\n");
    fclose(f);
    return diffs ? 1 : 0;
}
