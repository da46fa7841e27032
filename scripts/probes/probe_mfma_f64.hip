// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: (1) operand/result lane layout, (2) whether the k-accumulation
// is the sequential chain fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c)))) bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

__global__ void mfma_once(const double *a, const double *b, const double *c, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

int main()
{
    double *da, *db, *dc, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 512); hipMalloc(&dd, 512);
    std::vector<double> a(64), b(64), c(64, 0.0), d(64);
    // ---- layout: unit impulses
    int a_i[64], a_k[64], a_blk[64], b_k[64], b_j[64], b_blk[64];
    // first find for each (la, lb) which output lanes light up
    static int hit[64][64];
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
        std::fill(a.begin(), a.end(), 0.0); std::fill(b.begin(), b.end(), 0.0);
        a[la] = 1.0; b[lb] = 1.0;
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
        hit[la][lb] = -1;
        for (int l = 0; l < 64; ++l) if (d[l] != 0.0) hit[la][lb] = l;
    }
    printf("hit[la][lb] (output lane, -1 none), rows la=0..15, cols lb=0..15 (block 0):\n");
    for (int la = 0; la < 16; ++la) { for (int lb = 0; lb < 16; ++lb) printf("%3d", hit[la][lb]); printf("\n"); }
    printf("cross-block sample hit[0][16]=%d hit[16][16]=%d hit[16][17]=%d hit[17][16]=%d hit[20][16]=%d hit[16][20]=%d\n",
           hit[0][16], hit[16][16], hit[16][17], hit[17][16], hit[20][16], hit[16][20]);
    (void)a_i; (void)a_k; (void)a_blk; (void)b_k; (void)b_j; (void)b_blk;

    // ---- derive k index: lanes la, lb interact iff same block and same k.  Print interaction classes for block 0.
    // ---- accumulation order: random data, compare against chains in k order 0..3 and 3..0
    std::mt19937_64 rng(5);
    auto rnd = [&]() { return (double)(float)((rng() >> 11) * (1.0 / 9007199254740992.0) * 12.0); };
    int bad_fwd = 0, bad_rev = 0, total = 0;
    for (int trial = 0; trial < 200; ++trial) {
        for (int l = 0; l < 64; ++l) { a[l] = rnd(); b[l] = rnd(); c[l] = rnd() * 1000.0 * rnd(); }
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
        // model: for output lane lo, find contributing (la, lb) pairs from the impulse table
        for (int lo = 0; lo < 64; ++lo) {
            std::vector<std::pair<int,int>> terms;
            for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (hit[la][lb] == lo) terms.push_back({la, lb});
            if (terms.size() != 4) { printf("lane %d has %zu terms\n", lo, terms.size()); return 1; }
            double f = c[lo], r = c[lo];
            for (int t = 0; t < 4; ++t) f = fma(a[terms[t].first], b[terms[t].second], f);
            for (int t = 3; t >= 0; --t) r = fma(a[terms[t].first], b[terms[t].second], r);
            bad_fwd += memcmp(&f, &d[lo], 8) != 0; bad_rev += memcmp(&r, &d[lo], 8) != 0; ++total;
        }
    }
    printf("accumulation: %d outputs; mismatches vs chain in ascending (la) order: %d, descending: %d\n", total, bad_fwd, bad_rev);
    // print the term order for lane 0 and lane 5
    for (int lo : {0, 5, 21}) {
        printf("output lane %d <- ", lo);
        for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (hit[la][lb] == lo) printf("(a%d,b%d) ", la, lb);
        printf("\n");
    }
    return 0;
}
