p='scl_slam_amd/csrc/sc_distance.hip'
s=open(p).read()
park="""// NC == 2 only: values that are merely carried across phase B leave the register file for its duration -- the
        // next candidates' sector-key pairs and this pair's ring keys are fetched by LDS-DMA (global_load_lds: HBM ->
        // wave-private LDS, no VGPR), the column norms are stored there; phase B itself uses no wave scratch
        double2 *park = reinterpret_cast<double2 *>(vk2);
        if constexpr (NC == 2) {
            wave_fence();
            typedef const void __attribute__((address_space(1))) *gptr_t;
            typedef void __attribute__((address_space(3))) *lptr_t;
            const int rl = lane < RG ? lane : RG - 1;
            const size_t s0i = slot_next >= 0 ? (size_t)slot_next : 0, s1i = slot_next_y >= 0 ? (size_t)slot_next_y : 0;
            const size_t r0i = slot >= 0 ? (size_t)slot : 0, r1i = slot_y >= 0 ? (size_t)slot_y : 0;
            __builtin_amdgcn_global_load_lds((gptr_t)(a.vkey + s0i * S + j0), (lptr_t)(park + 0 * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(a.vkey + s1i * S + j0), (lptr_t)(park + 1 * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(a.rkey4 + (size_t)rl * a.rk_cap + r0i), (lptr_t)(park + 2 * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(a.rkey4 + (size_t)rl * a.rk_cap + r1i), (lptr_t)(park + 3 * 64), 16, 0, 0);
            park[4 * 64 + lane] = nk_cur;
            park[5 * 64 + lane] = nk_cur_y;
        }"""
unpark="""if constexpr (NC == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA fetches issued before phase B have landed
            wave_fence();
            vk_next = park[0 * 64 + lane]; vk_next_y = park[1 * 64 + lane];
            { const double2 t0 = park[2 * 64 + lane], t1 = park[3 * 64 + lane];
              rk_cand = *reinterpret_cast<const float4 *>(&t0); rk_cand_y = *reinterpret_cast<const float4 *>(&t1); }
            nk_cur = park[4 * 64 + lane]; nk_cur_y = park[5 * 64 + lane];
            wave_fence();
        }"""
assert "PARK_BEFORE_B" in s and "UNPARK_AFTER_B" in s
s=s.replace("PARK_BEFORE_B",park).replace("UNPARK_AFTER_B",unpark)
# in NC==2 mode do not load vk_next / rk_cand into registers before B
s=s.replace("        if (slot_next >= 0) vk_next = *reinterpret_cast<const double2 *>(a.vkey + (size_t)slot_next * S + j0);\n        if (NC == 2 && slot_next_y >= 0)","        if (NC == 1 && slot_next >= 0) vk_next = *reinterpret_cast<const double2 *>(a.vkey + (size_t)slot_next * S + j0);\n        if (false && slot_next_y >= 0)")
s=s.replace("        if (slot >= 0 && rk_on && lane < RG) rk_cand = a.rkey4[(size_t)lane * a.rk_cap + slot];\n        if (NC == 2 && slot_y >= 0 && rk_on && lane < RG)","        if (NC == 1 && slot >= 0 && rk_on && lane < RG) rk_cand = a.rkey4[(size_t)lane * a.rk_cap + slot];\n        if (false && slot_y >= 0 && rk_on && lane < RG)")
open(p,'w').write(s)
