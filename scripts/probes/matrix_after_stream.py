import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
if os.environ.get('WITH_TORCH'):
    import torch
    torch.cuda.synchronize()
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors
R, S, n = 64, 120, 10000
eng = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=100, initial_capacity=n + 64)
eng.save_bulk(synth_descriptors(n, R, S, seed=1002, revisit_frac=0.01))
n_elig = n - 100
qs = (n_elig + (np.arange(64) % 100)).astype(np.int32)
def mat(tag):
    eng.sc_distance_matrix(qs[:16], 0, n_elig)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.sc_distance_matrix(qs, 0, n_elig); ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, "matrix call ms:", [round(t, 2) for t in ts], flush=True)
mat("fresh engine")
sq = (n_elig + (np.arange(2048) % 100)).astype(np.int32)
eng.detect_full_stream(sq[:64], 0, n_elig, 16, 2)
mat("after a stream call of 64 scans")
eng.detect_full_stream(sq, 0, n_elig, 16, 2)
mat("after a stream call of 2048 scans")
eng.profile_reset(); eng.profile_enable(3)
eng.detect_full_stream(sq, 0, n_elig, 16, 2)
eng.profile_enable(0)
mat("after a profiled stream call")
eng.profile_reset(); eng.profile_enable(2)
mat("with profile_enable(2)")
eng.close()
# ... and in bench.py's context: torch initialised on the device, the stream driven through FullScanStream
eng = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=100, initial_capacity=n + 64)
eng.save_bulk(synth_descriptors(n, R, S, seed=1002, revisit_frac=0.01))
mat("second engine, torch initialised")
from scl_slam_amd.sharded import FullScanStream
st = FullScanStream(eng, 0, 1, device=None, depth=2, merge_every=16, scans_per_launch=16, native_chunk=1024, exchange="allreduce")
st.submit_many(sq, 0, n_elig); res = st.drain()
mat("after FullScanStream")
eng.close()
