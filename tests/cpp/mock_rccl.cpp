// mock_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for the collective library of the sharded front (SCL_RCCL_LIB).
//
// RCCL refuses two ranks on one device and a one-GPU box has only one, so the control flow of the device-side exchange
// (scl_slam_amd/csrc/sharded_front.hip: pack kernels, grouped all-reduce, select kernels, second all-reduce, one D2H) could
// never run with G > 1 ranks there.  This library exports the five entry points the front loads from librccl with the same
// signatures; a group's all-reduces are recorded between ncclGroupStart and ncclGroupEnd, ncclGroupEnd waits for every rank's
// stream, forms the element-wise minimum of the ranks' input buffers on the host and writes it to every rank's output buffer.
// It proves the control flow, the packing and the selection -- not RCCL.  Nothing links it into the product: the tests build it
// (Makefile: tests/cpp/libmock_rccl.so) and point SCL_RCCL_LIB at it in a child process.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <vector>

namespace {
struct MockComm { int rank, size; };
struct MockOp { const void *in; void *out; size_t count; MockComm *comm; hipStream_t stream; };
thread_local std::vector<MockOp> g_ops;
thread_local bool g_open = false;
}  // namespace

extern "C" {

// the front asks for this symbol: a library that has it may run several ranks on one device
int scl_collective_allows_shared_devices(void) { return 1; }

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *)
{
    for (int c = 0; c < n; ++c) comms[c] = reinterpret_cast<ncclComm_t>(new MockComm{c, n});
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete reinterpret_cast<MockComm *>(c); return ncclSuccess; }
ncclResult_t ncclGroupStart() { g_ops.clear(); g_open = true; return ncclSuccess; }
ncclResult_t ncclAllReduce(const void *in, void *out, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    if (!g_open || dt != ncclUint64 || op != ncclMin) return ncclInvalidArgument;
    g_ops.push_back(MockOp{in, out, count, reinterpret_cast<MockComm *>(comm), stream});
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
    g_open = false;
    if (g_ops.empty()) return ncclSuccess;
    const size_t count = g_ops[0].count;
    const int size = g_ops[0].comm->size;
    if ((int)g_ops.size() != size) return ncclInvalidUsage;                      // every rank takes part exactly once
    std::vector<unsigned long long> acc(count, ~0ull), tmp(count);
    std::vector<char> seen((size_t)size, 0);
    for (const MockOp &o : g_ops) {
        if (o.count != count || o.comm->size != size || seen[(size_t)o.comm->rank]) return ncclInvalidUsage;
        seen[(size_t)o.comm->rank] = 1;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpy(tmp.data(), o.in, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        for (size_t i = 0; i < count; ++i) acc[i] = tmp[i] < acc[i] ? tmp[i] : acc[i];
    }
    for (const MockOp &o : g_ops)
        if (hipMemcpy(o.out, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    g_ops.clear();
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t) { return "mock collective (tests/cpp/mock_rccl.cpp)"; }

}  // extern "C"
