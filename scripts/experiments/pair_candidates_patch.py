import sys
p='scl_slam_amd/csrc/sc_distance.hip'
s=open(p).read()
def rep(old,new,cnt=1):
    global s
    assert s.count(old)>=1, old[:90]
    s=s.replace(old,new,cnt)

rep("""template <int RG, int W, int CH, int S, int MAXT, bool STAMP>
__global__ __launch_bounds__(MAXT) void sc_distance_wave_kernel(ScBatchArgs ab)""",
"""// NC = 2: every wave scores two candidates at a time through phase B -- one query window (7 x ds_read_b128 per
// ring) feeds the 2 x 26 fmas of both, so the ring windows, which keep the LDS array as busy as the fmas keep
// the fp64 pipe, cost half per candidate.  Alignment and phases C/D run once per candidate, one after the
// other, on the same code (a two-trip loop that is not unrolled; the second candidate's registers are moved
// into the first's before its trip).
template <int RG, int W, int CH, int S, int MAXT, bool STAMP, int NC = 1>
__global__ __launch_bounds__(MAXT) void sc_distance_wave_kernel(ScBatchArgs ab)""")
rep("    if (threadIdx.x == 0) *next_ticket = c_lo + nwaves;","    if (threadIdx.x == 0) *next_ticket = c_lo + nwaves * NC;")
rep("    int ci = c_lo + wave;\n    const bool have_work = ci < c_hi;","    int ci = c_lo + wave * NC;\n    int ci_y = ci + 1;                         // second candidate of the wave (NC == 2)\n    const bool have_work = ci < c_hi;")
rep("""    double2 nk = make_double2(0.0, 0.0);
    float4 k0[CH], k1[CH];
""","""    double2 nk = make_double2(0.0, 0.0);
    float4 k0[CH], k1[CH];
    // second candidate (NC == 2): same state, suffix _y; invalid candidates keep valid (stale) addresses
    int slot_y = (NC == 2 && have_work && ci_y < c_hi) ? (a.cand ? a.cand[ci_y] : a.slot_base + ci_y) : -1;
    int s_start_y = 0;
    const float4 *kp0_y = a.desc, *kp1_y = a.desc;
    double2 nk_y = make_double2(0.0, 0.0);
    float4 k0_y[CH], k1_y[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) { k0[u] = make_float4(0.f, 0.f, 0.f, 0.f); k1[u] = k0[u]; k0_y[u] = k0[u]; k1_y[u] = k0[u]; }
""")
rep("""    if (slot >= 0) {
        const double2 vk = *reinterpret_cast<const double2 *>(a.vkey + (size_t)slot * S + j0);
        st_t = stamp();
        align_and_fetch(slot, vk, s_start);
        st_a += stamp() - st_t;
    }
""","""    st_t = stamp();
#pragma unroll 1
    for (int c = NC - 1; c >= 0; --c) {                       // second candidate first: its results move to the _y registers
        const int sl = c ? slot_y : slot;
        if (sl >= 0) {
            const double2 vk = *reinterpret_cast<const double2 *>(a.vkey + (size_t)sl * S + j0);
            int s_tmp = 0;
            align_and_fetch(sl, vk, s_tmp);
            if (c) s_start_y = s_tmp; else s_start = s_tmp;
        }
        if (NC == 2 && c == 1) {
            nk_y = nk; kp0_y = kp0; kp1_y = kp1;
#pragma unroll
            for (int u = 0; u < CH; ++u) { k0_y[u] = k0[u]; k1_y[u] = k1[u]; }
        }
    }
    st_a += stamp() - st_t;
""")
rep("""        int ticket = 0;
        if (lane == 0) ticket = atomicAdd(next_ticket, 1);
        const int ci_next = __builtin_amdgcn_readfirstlane(ticket);
        int slot_next = -1;
        if (ci_next < c_hi) slot_next = a.cand ? a.cand[ci_next] : a.slot_base + ci_next;
        double2 vk_next = make_double2(0.0, 0.0);
        if (slot_next >= 0) vk_next = *reinterpret_cast<const double2 *>(a.vkey + (size_t)slot_next * S + j0);

        // the candidate's ring key travels under phase B (consumed by the fused ring-key metric after it)
        float4 rk_cand = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot >= 0 && rk_on && lane < RG) rk_cand = a.rkey4[(size_t)lane * a.rk_cap + slot];

        double acc0[W], acc1[W];
        const double2 nk_cur = nk;
        const int s_start_cur = s_start;
        st_t = stamp();
        if (slot >= 0) {""","""        int ticket = 0;
        if (lane == 0) ticket = atomicAdd(next_ticket, NC);
        const int ci_next = __builtin_amdgcn_readfirstlane(ticket);
        const int ci_next_y = ci_next + 1;
        int slot_next = -1, slot_next_y = -1;
        if (ci_next < c_hi) slot_next = a.cand ? a.cand[ci_next] : a.slot_base + ci_next;
        if (NC == 2 && ci_next_y < c_hi) slot_next_y = a.cand ? a.cand[ci_next_y] : a.slot_base + ci_next_y;
        double2 vk_next = make_double2(0.0, 0.0), vk_next_y = make_double2(0.0, 0.0);
        if (slot_next >= 0) vk_next = *reinterpret_cast<const double2 *>(a.vkey + (size_t)slot_next * S + j0);
        if (NC == 2 && slot_next_y >= 0) vk_next_y = *reinterpret_cast<const double2 *>(a.vkey + (size_t)slot_next_y * S + j0);

        // the candidate's ring key travels under phase B (consumed by the fused ring-key metric after it)
        float4 rk_cand = make_float4(0.f, 0.f, 0.f, 0.f), rk_cand_y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot >= 0 && rk_on && lane < RG) rk_cand = a.rkey4[(size_t)lane * a.rk_cap + slot];
        if (NC == 2 && slot_y >= 0 && rk_on && lane < RG) rk_cand_y = a.rkey4[(size_t)lane * a.rk_cap + slot_y];

        double acc0[W], acc1[W];
        double acc0_y[NC == 2 ? W : 1], acc1_y[NC == 2 ? W : 1];
        double2 nk_cur = nk;
        int s_start_cur = s_start;
        double2 nk_cur_y = nk_y;
        const int s_start_cur_y = s_start_y;
        PARK_BEFORE_B
        st_t = stamp();
        if (slot >= 0 || slot_y >= 0) {""")
rep("""            for (int t = 0; t < W; ++t) { acc0[t] = 0.0; acc1[t] = 0.0; }
            // Explicit two-stage""","""            for (int t = 0; t < W; ++t) { acc0[t] = 0.0; acc1[t] = 0.0; }
            if constexpr (NC == 2) {
#pragma unroll
                for (int t = 0; t < W; ++t) { acc0_y[t] = 0.0; acc1_y[t] = 0.0; }
            }
            // Explicit two-stage""")
rep("""                const float4 *np0 = kp0 + (size_t)(ch + 1) * CH * S, *np1 = kp1 + (size_t)(ch + 1) * CH * S;
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float a0[4] = {k0[u].x, k0[u].y, k0[u].z, k0[u].w};
                    const float a1[4] = {k1[u].x, k1[u].y, k1[u].z, k1[u].w};
                    if (more) { k0[u] = np0[u * S]; k1[u] = np1[u * S]; }
""","""                const float4 *np0 = kp0 + (size_t)(ch + 1) * CH * S, *np1 = kp1 + (size_t)(ch + 1) * CH * S;
                const float4 *np0_y = kp0_y + (size_t)(ch + 1) * CH * S, *np1_y = kp1_y + (size_t)(ch + 1) * CH * S;
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float a0[4] = {k0[u].x, k0[u].y, k0[u].z, k0[u].w};
                    const float a1[4] = {k1[u].x, k1[u].y, k1[u].z, k1[u].w};
                    const float b0[4] = {k0_y[u].x, k0_y[u].y, k0_y[u].z, k0_y[u].w};
                    const float b1[4] = {k1_y[u].x, k1_y[u].y, k1_y[u].z, k1_y[u].w};
                    if (more) { k0[u] = np0[u * S]; k1[u] = np1[u * S]; }
                    if (NC == 2 && more) { k0_y[u] = np0_y[u * S]; k1_y[u] = np1_y[u * S]; }
""")
rep("""                        pin_ring(acc0, acc1);
                        qp += qstep;                        // the last ring over-reads one row: it lands in""","""                        pin_ring(acc0, acc1);
                        if constexpr (NC == 2) pin_ring(acc0_y, acc1_y);
                        qp += qstep;                        // the last ring over-reads one row: it lands in""")
rep("""                        for (int t = 0; t < W; ++t) {
                            acc0[t] = fma(kd0, q[t], acc0[t]);
                            acc1[t] = fma(kd1, q[t + 1], acc1[t]);
                        }
                        if (MAXT > 512) {""","""                        for (int t = 0; t < W; ++t) {
                            acc0[t] = fma(kd0, q[t], acc0[t]);
                            acc1[t] = fma(kd1, q[t + 1], acc1[t]);
                        }
                        if constexpr (NC == 2) {            // the second candidate meets the same window
                            const double ke0 = (double)b0[i], ke1 = (double)b1[i];
#pragma unroll
                            for (int t = 0; t < W; ++t) {
                                acc0_y[t] = fma(ke0, q[t], acc0_y[t]);
                                acc1_y[t] = fma(ke1, q[t + 1], acc1_y[t]);
                            }
                        }
                        if (MAXT > 512) {""")
rep("""        if (slot_next >= 0) align_and_fetch(slot_next, vk_next, s_start);
        { const unsigned long long t1 = stamp(); st_a += t1 - st_t; st_t = t1; }
""","""        UNPARK_AFTER_B
#pragma unroll 1
        for (int c = NC - 1; c >= 0; --c) {                   // second candidate first: its results move to the _y registers
            const int sl = c ? slot_next_y : slot_next;
            if (sl >= 0) {
                int s_tmp = 0;
                align_and_fetch(sl, c ? vk_next_y : vk_next, s_tmp);
                if (c) s_start_y = s_tmp; else s_start = s_tmp;
            }
            if (NC == 2 && c == 1) {
                nk_y = nk; kp0_y = kp0; kp1_y = kp1;
#pragma unroll
                for (int u = 0; u < CH; ++u) { k0_y[u] = k0[u]; k1_y[u] = k1[u]; }
            }
        }
        { const unsigned long long t1 = stamp(); st_a += t1 - st_t; st_t = t1; }
""")
rep("""        // ---- phases C/D in two passes of HSH shifts ------------------------------
        if (slot >= 0 && rk_on) {""","""        // ---- phases C/D in two passes of HSH shifts, one candidate after the other ----
        int ci_w = ci, slot_w = slot;
        float4 rk_w = rk_cand;
#pragma unroll 1
        for (int c = 0; c < NC; ++c) {
        if constexpr (NC == 2) {
            if (c == 1) {                                    // second trip: the second candidate's registers
                ci_w = ci_y; slot_w = slot_y; rk_w = rk_cand_y; nk_cur = nk_cur_y; s_start_cur = s_start_cur_y;
#pragma unroll
                for (int t = 0; t < W; ++t) { acc0[t] = acc0_y[t]; acc1[t] = acc1_y[t]; }
                if (ci_w >= c_hi) break;
            }
        }
        if (slot_w >= 0 && rk_on) {""")
a0=s.index("        if (slot_w >= 0 && rk_on) {")
tail="        if (ci_next >= c_hi) break;\n        ci = ci_next;\n        slot = slot_next;\n    }"
a1=s.index(tail)
region=s[a0:a1]
region=region.replace("const float4 b = rk_cand;","const float4 b = rk_w;")
region=region.replace("if (lane == 0) a.out_d2[ci] = result;","if (lane == 0) a.out_d2[ci_w] = result;")
region=region.replace("unsigned long long key = ((unsigned long long)(unsigned)rbits << 32) | (unsigned)ci;","unsigned long long key = ((unsigned long long)(unsigned)rbits << 32) | (unsigned)ci_w;")
region=region.replace("        if (slot >= 0) {\n            double dmin = kInf;","        if (slot_w >= 0) {\n            double dmin = kInf;")
region=region.replace("                a.out_dist[ci] = ok ? dmin : kBigDist;\n                a.out_shift[ci] = ok ? smin : 0;","                a.out_dist[ci_w] = ok ? dmin : kBigDist;\n                a.out_shift[ci_w] = ok ? smin : 0;")
region=region.replace("if (ds < kBigDist && ((ds < w_best) | ((ds == w_best) & (ci < w_bidx)))) { w_best = ds; w_bidx = ci; w_bshift = ss; }","if (ds < kBigDist && ((ds < w_best) | ((ds == w_best) & (ci_w < w_bidx)))) { w_best = ds; w_bidx = ci_w; w_bshift = ss; }")
region=region.replace("        } else if (lane == 0) {\n            a.out_dist[ci] = kBigDist;\n            a.out_shift[ci] = 0;\n        }\n","        } else if (lane == 0) {\n            a.out_dist[ci_w] = kBigDist;\n            a.out_shift[ci_w] = 0;\n        }\n        }                                                    // next candidate of the wave\n")
assert "[ci]" not in region
s=s[:a0]+region+"""        if (ci_next >= c_hi) break;
        ci = ci_next;
        slot = slot_next;
        ci_y = ci_next_y;
        slot_y = slot_next_y;
    }"""+s[a1+len(tail):]
rep("""template <int RG, int W, int CH, int S, int MAXT = 512, bool STAMP = false>
hipError_t launch_wave(""","""template <int RG, int W, int CH, int S, int MAXT = 512, bool STAMP = false, int NC = 1>
hipError_t launch_wave(""")
s=s.replace("sc_distance_wave_kernel<RG, W, CH, S, MAXT, STAMP>","sc_distance_wave_kernel<RG, W, CH, S, MAXT, STAMP, NC>")
CHP=sys.argv[1] if len(sys.argv)>1 else "2"
rep("""    if (wave_ok && db.RG == 16 && W == 13 && db.S == 120) return launch_wave<16, 13, 4, 120, 512>(one, num_cu, stream);""",
"""    if (wave_ok && db.RG == 16 && W == 13 && db.S == 120 && pair_candidates()) return launch_wave<16, 13, %s, 120, 512, false, 2>(one, num_cu, stream);
    if (wave_ok && db.RG == 16 && W == 13 && db.S == 120) return launch_wave<16, 13, 4, 120, 512>(one, num_cu, stream);""" % CHP)
rep("""    if (db.RG == 5 && W == 7 && db.S == 60) return launch_wave<5, 7, 5, 60>(ab, num_cu, stream);
    return launch_wave<16, 13, 4, 120, 512>(ab, num_cu, stream);""","""    if (db.RG == 5 && W == 7 && db.S == 60) return launch_wave<5, 7, 5, 60>(ab, num_cu, stream);
    if (pair_candidates()) return launch_wave<16, 13, %s, 120, 512, false, 2>(ab, num_cu, stream);
    return launch_wave<16, 13, 4, 120, 512>(ab, num_cu, stream);""" % CHP)
rep("""int ablate_flags()
{""","""// two candidates per wave through phase B (SCL_SC_PAIR=1; default one)
int pair_candidates()
{
    static const int on = [] { const char *e = getenv("SCL_SC_PAIR"); return (e && e[0] == '1') ? 1 : 0; }();
    return on;
}

int ablate_flags()
{""")
s=s.replace("            constexpr int BT = (MAXT > 512) ? 3 : 5;","            constexpr int BT = (MAXT > 512 || NC == 2) ? 3 : 5;")
s=s.replace("                        if (MAXT <= 512) {                  // nqe/vq (inside the allocation), values unused","                        if (MAXT <= 512 && NC == 1) {       // nqe/vq (inside the allocation), values unused")
s=s.replace("                        if (MAXT > 512) {                   // 3 waves/SIMD build: one window buffer, the partner","                        if (MAXT > 512 || NC == 2) {        // lean builds: one window buffer, refilled behind the fmas; the partner")
s=s.replace("                    constexpr int DB = (MAXT > 512) ? 3 : 5;","                    constexpr int DB = (MAXT > 512 || NC == 2) ? 3 : 5;")
s=s.replace("            constexpr int NG = S / 4, FB = 3;                 // sector groups of four, FB groups per batch","            constexpr int NG = S / 4, FB = (NC == 2) ? 2 : 3; // sector groups of four, FB groups per batch")
open(p,'w').write(s)
