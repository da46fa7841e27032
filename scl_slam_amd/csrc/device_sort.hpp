// device_sort.hpp -- the stable radix sort of (key, value) pairs and the prefix sums of the verification path (host-side interface).
// Written for the sizes that path has (1e5 .. 3e6 pairs, keys of 30 .. 40 bits, up to a few dozen equal segments): PCL's VoxelGrid
// orders its points by voxel (DM.h:1183-1185, 1200-1201 through pcl::VoxelGrid::applyFilter), the ICP batch orders its sources along
// a Hilbert curve and turns cell counts into cell starts.  csrc/device_sort.hip has the kernels and says how a pass works.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace scl {

constexpr int kSortMaxSegments = 64;

// Elements [off[s], off[s + 1]) form segment s; the segments follow each other (off[0] = 0) and each is sorted on its own.
struct SortSegments {
    int nseg;
    int off[kSortMaxSegments + 1];
};

// bytes of scratch a sort of n pairs in nseg segments needs (histograms of the passes)
size_t sort_scratch_bytes(size_t n, int nseg);

// Stable LSD radix sort on the key bits [0, bits).  keys_in / vals_in are used as the passes' second buffer (their contents are
// lost); the result is in keys_out / vals_out.  An odd number of passes is chosen so that it ends there.
hipError_t sort_pairs_u32(void *scratch, unsigned int *keys_in, unsigned int *keys_out, unsigned int *vals_in, unsigned int *vals_out,
                          int n, int bits, hipStream_t stream);
hipError_t sort_pairs_u64(void *scratch, unsigned long long *keys_in, unsigned long long *keys_out, unsigned int *vals_in,
                          unsigned int *vals_out, int n, int bits, hipStream_t stream);
// every segment sorted on its own on the key bits [0, bits) (the whole 64-bit key travels).  seg_hi (optional, bits <= 32): the caller
// says that every key of segment s has the high word seg_hi[s] -- the passes then move 8-byte records (value, low word) instead of a
// key and a value in two arrays, and the last one puts the high word back.
hipError_t sort_pairs_u64_segmented(void *scratch, unsigned long long *keys_in, unsigned long long *keys_out, unsigned int *vals_in,
                                    unsigned int *vals_out, const SortSegments &seg, int bits, hipStream_t stream,
                                    const unsigned int *seg_hi = nullptr);

// bytes of scratch a prefix sum of n values needs
size_t scan_scratch_bytes(size_t n);
// out[i] = in[0] + ... + in[i - 1] (exclusive) or ... + in[i] (inclusive); in == out is allowed
hipError_t prefix_sum_i32(void *scratch, const int *in, int *out, int n, bool inclusive, hipStream_t stream);

}  // namespace scl
