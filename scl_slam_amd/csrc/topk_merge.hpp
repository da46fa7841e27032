// topk_merge.hpp -- the merge of the ring-key scan's per-workgroup lists, as a block-level device function: the one-workgroup kernel
// that CONSUMES the k nearest keys (sc_cand_exact_kernel: the reference-faithful detection; topk_merge_pack_kernel: the bare search)
// runs it in its prologue, so the search is one launch of the scan + the consumer -- no last-workgroup ticket, no fence.
// (Measured alternatives, 10k keyframes x 64 rings, k = 3: round 4's two launches 8.9 + 4.7 us; ONE launch with a ticketed last
//  workgroup 16.4 us -- a chain of eight dependent memory round trips: query key, rows, lists, fence, ticket, lists again, results.)
#pragma once

#include <float.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace scl {

constexpr unsigned long long kTopkNoKey = ~0ull;
constexpr int kTopkMergeLds = 3072;      // keys of the workgroups' lists a merging workgroup holds in LDS (twice: lists + candidates)

struct TopkMergeShared { unsigned long long tau; unsigned int nc; };

// L lists of k keys (distance bits << 32 | slot; sorted ascending, kTopkNoKey behind the valid keys) at `lists` in global memory,
// L * k <= kTopkMergeLds.  Every thread of the block calls; skey / scand: L * k keys of LDS each.  out_idx / out_d2 (k entries; LDS or
// global) receive the k smallest keys in ascending order, (-1, FLT_MAX) behind them.  Three steps, none with a dependent chain:
//   1. a threshold: with m = ceil(k / L) and q = ceil(k / m), tau = the q-th smallest of the lists' m-th keys -- q lists hold m keys
//      <= tau each, so at least k keys are <= tau and nothing above tau is among the k best;
//   2. the keys <= tau are compacted (a few dozen as a rule: the lists' heads are spread);
//   3. every candidate counts the candidates below it (keys are unique: the slot is their low word): rank < k -> output[rank].
// (A merge by binary search per (key, list) was built first: 39 x 5 DEPENDENT LDS reads per key -- 40 us at k = 25.)
// Ends with a barrier: the outputs are visible to the block.
// keys_in_lds: the caller has put the L * k keys into skey already (its own loads, issued beside others: the candidates' kernel).
__device__ __forceinline__ void topk_merge_lists(const unsigned long long *lists, int L, int k, unsigned long long *skey, unsigned long long *scand,
                                                 TopkMergeShared *sh, int *out_idx, float *out_d2, const bool keys_in_lds = false)
{
    const int T = (int)blockDim.x, tid = (int)threadIdx.x;
    const int count = L * k;
    if (!keys_in_lds) for (int p = tid; p < count; p += T) skey[p] = lists[p];
    for (int i = tid; i < k; i += T) { out_idx[i] = -1; out_d2[i] = FLT_MAX; }
    if (tid == 0) { sh->tau = kTopkNoKey; sh->nc = 0u; }
    __syncthreads();
    if (L > 0) {
        const int m = (k + L - 1) / L, q = (k + m - 1) / m;      // q <= L
        for (int l = tid; l < L; l += T) {
            const unsigned long long x = skey[l * k + (m - 1)];
            int r = 0;
            for (int j = 0; j < L; ++j) {
                const unsigned long long y = skey[j * k + (m - 1)];
                r += (y < x || (y == x && j < l)) ? 1 : 0;       // (ties only among kTopkNoKey entries: broken by the list's number)
            }
            if (r == q - 1) sh->tau = x;
        }
    }
    __syncthreads();
    const unsigned long long tau = sh->tau;
    for (int p = tid; p < count; p += T) {
        const unsigned long long x = skey[p];
        if (x != kTopkNoKey && x <= tau) scand[atomicAdd(&sh->nc, 1u)] = x;
    }
    __syncthreads();
    const int C = (int)sh->nc;
    for (int p = tid; p < C; p += T) {
        const unsigned long long x = scand[p];
        int r = 0;
#pragma unroll 4
        for (int j = 0; j < C; ++j) r += scand[j] < x ? 1 : 0;
        if (r < k) { out_idx[r] = (int)(unsigned)(x & 0xffffffffull); out_d2[r] = __int_as_float((int)(x >> 32)); }
    }
    __syncthreads();
}

}  // namespace scl
