import sys, time, numpy as np
sys.path.insert(0, '.')
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors
R,S=64,120
eng=ScanContextEngine(num_ring=R,num_sector=S,num_candidates=3,num_exclude_recent=100,initial_capacity=10064)
eng.save_bulk(synth_descriptors(9900,R,S,seed=1002)); eng.save_bulk(synth_descriptors(100,R,S,seed=424242,revisit_frac=0.0))
def run(n,depth=2):
    infl=[]
    for i in range(n):
        infl.append(eng.detect_full_submit(9900+(i%100),0,9900))
        if len(infl)>=depth: eng.detect_full_collect(infl.pop(0))
    while infl: eng.detect_full_collect(infl.pop(0))
for prof in (0,2,0,2):
    eng.profile_enable(prof); run(20)
    t0=time.perf_counter(); run(300); dt=(time.perf_counter()-t0)/300
    print("prof",prof,"us/step %.1f"%(dt*1e6))
for depth in (1,2,3,4):
    eng.profile_enable(0); run(20,depth)
    t0=time.perf_counter(); run(300,depth); dt=(time.perf_counter()-t0)/300
    print("depth",depth,"us/step %.1f"%(dt*1e6))
eng.close()
