import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors, synth_structured_cloud, synth_scan
descs = synth_descriptors(600, 64, 120, seed=3)
cloud = synth_structured_cloud(20000, seed=1); scan = synth_scan(50000, seed=2)
def free_mb():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return f / 2**20
base = None
for it in range(12):
    e = ScanContextEngine(num_ring=64, num_sector=120, initial_capacity=64)
    e.save_bulk(descs)
    e.detect_full_range(599, 0, 500); e.detect_intra(599)
    ts = e.detect_full_submit_many([599, 598, 597], 0, 400); [e.detect_full_collect(t) for t in ts]
    e.detect_full_stream([599, 598, 597, 596, 595], 0, 450, 4, 2)
    e.make_and_save_filtered(scan, 0.4, 0, 600)
    for k in range(3): e.keyframe_put(0, k, cloud)
    e.loop_icp_from_store(0, 2, np.eye(4, dtype=np.float32), 1, 1, [np.eye(4, dtype=np.float32)] * 3, 0.3)
    e.icp_align_batch(cloud, [cloud, cloud, cloud])
    e.geometric_verification(cloud[::2], cloud, 200, 0.25, 0.45, 1)
    e.close()
    f = free_mb()
    if it == 1: base = f
    print("iteration %d: free device memory %.1f MiB" % (it, f), flush=True)
assert base is not None and abs(free_mb() - base) < 64.0, "device memory drifts across create/destroy cycles"
print("no leak: free memory within 64 MiB of the second cycle after 12 create/use/destroy cycles")
