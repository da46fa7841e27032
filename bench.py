#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Scan Context loop-closure hot path on MI355X.

Metric (BASELINE.json): loop-closure candidates/sec (+ SC-distance GB/s).  One "step" = one
batch of incoming scans -- `--scans-per-launch` of them (16: the scans of a robot team that
arrive together, one launch group of the engine) -- and for each scan of the batch the
place-recognition pass over the resident database: full ring-key scan (exact top-k) +
column-shifted SC distance against EVERY eligible keyframe + global arg-min.  `value` counts
(query, keyframe) pairs scored per second; `ms_per_scan` is the step time over the scans of a
step.  (Up to round 1 a step was a single scan: `--steps 20` then timed 20 scans = two launch
groups, a region of 0.3 ms that measured the pipeline's fill and drain, not its rate.)

Workloads
  N = 1  BASELINE configs[1]: 10 000 synthetic Velodyne-64 keyframes, 64x120 SC.
  N > 1  BASELINE configs[3]: the keyframe database sharded by keyframe index, 12 500 keyframes
         per GPU (100 000 at N = 8; weak scaling), every rank scores its own shard with no
         data-path collective, the per-scan winners are reduced with RCCL min all-reduces on
         packed 64-bit keys (scl_slam_amd/sharded.py), batched over `--native-chunk` scans.

Launch: `python bench.py --gpus N ...` starts its own N ranks (a child `python -m
torch.distributed.run`, created before this process touches the GPU; nothing is re-exec'ed);
under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.  WORLD_SIZE != --gpus is
an error.

Inputs are resident in HBM when the timed region starts (database shard + the query
keyframes); only the 24-byte result leaves the device per step.  The timed block of `--steps`
steps is repeated `--repeats` times (each bracketed by barrier + synchronize); the median
repetition is reported, min / max beside it.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP spreads a process's streams over this many hardware queues (default 4), decided when the runtime initialises: the engine has
# five streams of its own and the verification prepares a scan's candidates on eight more (INTEGRATION.md asks a host application
# for the same export).  Recorded in the line as config.hip_hw_queues.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

R, S = 64, 120                     # Velodyne-64 Scan Context grid of BASELINE configs[1]
N_KEYFRAMES_1GPU = 10000           # configs[1]
N_KEYFRAMES_SHARD = 12500          # configs[3]: 100k keyframes over 8 GPUs
N_EXCLUDE = 100                    # NUM_EXCLUDE_RECENT, descriptor.h:1314
ALGO_BYTES_PER_PAIR = R * S * 4 + S * 4 + S * 4      # 31 680 B at 64x120 (fp32 descriptor + fp64 sector key)
# What the screening launch group (products in their second form + the tail launch: finishing beside the next batch's alignment) reads per
# KEYFRAME by design, once per launch whatever the number of scans: the chunk-major fp16 image of the
# descriptor (2 halves x 4 chunks x (S + 16) sectors x 16 B = 17 408 B at 64x120: 13 % padding so that no fragment wraps), the first
# part of the alignment image (8 rotated copies of the unit fp16 sector key x 256 B + 16 B of norm = 2 064 B; the second part only for
# the 4 % of the keyframes whose alignment needs the split stage), the tiled ring key (4 * 4 ceil(R/4)) and the 32-byte sector mask
# = 19 760 B -- not SURVEY's 31 680 B (fp32 descriptor): the kernels must not get credit for bytes they do not move.
KERNEL_BYTES_PER_KEYFRAME = 2 * 4 * (S + 16) * 16 + (8 * 128 * 2 + 16) + 4 * 4 * ((R + 3) // 4) + 32
# ... and per PAIR, whatever stays on the chip or not: first shift (4 B written by the alignment, read by products and finish),
# the two ring halves' partial sums (2 x 64 B written, read once), the bound d~ and the ring-key metric (4 B each)
KERNEL_BYTES_PER_PAIR_IO = 4 + 2 * 4 + 2 * 128 + 4 + 4
# A kernel that streams the database once per SCAN reads at least the fp16 copy, keys and mask per pair (the first form's figure):
SINGLE_SCAN_BYTES_PER_PAIR = R * S * 2 + S * 8 + 4 * 4 * ((R + 3) // 4) + 32
L2_RATE_GBS = 17800.0              # rows served from the XCDs' L2s (MI355X_MICROARCH.md): bounds the first form of the products (80x180 still uses it)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128, help="timed steps; a step = one batch of --scans-per-launch incoming scans")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=5, help="repetitions of the timed block of --steps steps (median reported)")
    ap.add_argument("--keyframes", type=int, default=0, help="keyframes per GPU (0: 10 000 at N = 1, 12 500 at N > 1)")
    ap.add_argument("--pipeline", type=int, default=2, help="kernel launches enqueued ahead (1 = strictly one after another)")
    ap.add_argument("--merge-every", type=int, default=16, help="N > 1, --native-chunk 0: scans whose per-rank winners share one exchange")
    ap.add_argument("--exchange", choices=("allreduce", "allgather"), default="allreduce",
                    help="N > 1: min all-reduce on packed keys (default) or all-gather + host merge (fallback, for comparison)")
    ap.add_argument("--scans-per-launch", type=int, default=16,
                    help="incoming scans scored by one kernel launch (1..16; the reference runs several robots, whose scans "
                         "arrive together): the workgroups that walk the same keyframes for the scans of a launch share "
                         "them through an XCD's L2, so the database crosses HBM once per launch, not once per scan")
    ap.add_argument("--native-chunk", type=int, default=1024,
                    help="scans handed to the engine's native submit/collect pipeline per call (0: drive every scan from Python)")
    ap.add_argument("--front", type=int, default=0,
                    help="G > 0: measure the C-ABI sharded front instead (scl_create_sharded: ONE process, G shards of 12 500 keyframes dealt "
                         "over the visible devices; the stream form and the single passes with their exchange).  Not the driver's contract line.")
    ap.add_argument("--dry-run", action="store_true",
                    help="print the rank command line `--gpus N` / the device list `--front G` would use, check that the node shows that many "
                         "GPUs (torch.cuda.device_count(): does not initialise HIP) and exit -- 0 when the run could start, 3 otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[2] geometric-verification measurement")
    ap.add_argument("--only-secondary", default="", help="comma-separated names: run only these secondary measurements (experiments)")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the 1-thread CPU sample (0 = auto ~10 s)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no torchrun environment
# ------------------------------------------------------------------------------------------------
def rank_command(args, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--dry-run"]


def rank_env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return env


def launch_ranks(args):
    """Start the N ranks as a fresh child (`python -m torch.distributed.run`).  This parent has not imported torch
    or touched HIP; it only relays the child's output (rank 0 prints the JSON line) and exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return subprocess.call(rank_command(args, port), env=rank_env())


def dry_run(args):
    """What `--gpus N` (or `--front G`) would start, and whether this node can: the GPU count comes from
    torch.cuda.device_count(), which does not initialise HIP on this image, so nothing touches a device."""
    import torch
    have = torch.cuda.device_count()
    need = args.front if args.front > 0 else args.gpus
    plan = {"dry_run": True, "gpus_visible": have, "gpus_needed": need,
            "env": {k: rank_env().get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS", "GPU_MAX_HW_QUEUES")}}
    if args.front > 0:
        plan["mode"] = "C-ABI sharded front: one process, scl_create_sharded over the device list"
        plan["devices"] = [i % max(1, have) for i in range(args.front)]
        plan["exchange"] = "RCCL min all-reduce x2 (exchange = 2)" if have >= args.front and args.front > 1 else "host merge (exchange = 1): shards share a card"
        plan["command"] = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--dry-run"]
    else:
        plan["mode"] = "one process per GPU over RCCL (torch.distributed 'nccl')" if args.gpus > 1 else "one process, one GPU"
        plan["command"] = rank_command(args, "<free port>") if args.gpus > 1 else [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--dry-run"]
        plan["keyframes_per_gpu"] = args.keyframes or (N_KEYFRAMES_1GPU if args.gpus == 1 else N_KEYFRAMES_SHARD)
    ok = have >= need and need >= 1
    plan["ok"] = ok
    if not ok:
        plan["error"] = (f"this node shows {have} GPU(s) (torch.cuda.device_count()), the run needs {need}: "
                         f"{'the shards of --front would share cards (plumbing, not scaling)' if args.front > 0 else 'one rank per GPU, RCCL refuses two ranks on one device'}")
    print(json.dumps(plan), flush=True)
    return 0 if ok else 3


# ------------------------------------------------------------------------------------------------
# CPU baseline (oracle as the measured CPU port; never on the product path)
# ------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cgroup_cpu():
    """CPU bandwidth the container may use (cgroup quota / period) and the throttling counters: a host that shows 256 cores in the
    affinity mask but a quota of 16 runs 256 threads at 16 cores' worth -- the all-core figure then scales with the quota, not
    with the thread count."""
    out = {}
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        out["quota_cores"] = None if q == "max" else float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            out["quota_cores"] = None if q <= 0 else q / p
        except Exception:
            out["quota_cores"] = "unknown"
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            st = dict(l.split() for l in open(path).read().strip().splitlines())
            out["nr_throttled"] = int(st.get("nr_throttled", 0)); out["throttled_usec"] = int(st.get("throttled_usec", st.get("throttled_time", 0)))
            break
        except Exception:
            pass
    return out


def _native_oracle():
    """BASELINE.md §2 prescribes -O3 -march=native for the CPU baseline.  The committed liboracle.so is built
    without -march (like the reference, CMakeLists.txt:4) because it must run on any host; the native build is
    made here, on the box it runs on."""
    import tempfile
    out = os.path.join(tempfile.gettempdir(), f"liboracle_native_{os.getpid()}.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("sc_oracle.c", "icp_oracle.c")]
    cmd = ["gcc", "-O3", "-march=native", "-DNDEBUG", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-std=gnu11",
           "-shared", "-o", out] + srcs + ["-lm", "-lpthread"]
    try:
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        return out
    except Exception:
        return None


def cpu_baseline(descs, n_pairs_hint, budget_s=8.0):
    """Reference-shaped CPU path (oracle/sc_oracle.c: copy per shift, norms twice, descriptor.h:1538-1569,
    + the ring-key scan) on this box's host cores.  Headline `value`: 1 thread, like the reference (all omp
    pragmas of the SC code are commented out, descriptor.h:1417-1517).  Variants: every core this process may
    use (the denominator of north_star's ">= 50x"), the copy-free restatement, and -march=native builds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cfg = ob.make_config(R=R, S=S)
    n_db = min(descs.shape[0], 2100)
    n_hist = n_db - N_EXCLUDE
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def sample_1thread(db, budget):
        keys = db.ringkeys(n_hist)
        batch = 200
        target = n_pairs_hint if n_pairs_hint > 0 else 1 << 30
        done, q = 0, n_db - 1
        t0 = time.perf_counter()
        while done < target:
            lo = done % (n_hist - batch)
            if lo == 0:
                q = n_db - 1 - (done // (n_hist - batch)) % N_EXCLUDE
                ob.knn(keys, db.ringkey(q), 3)                       # the per-query ring-key search
            cand = np.arange(lo, lo + batch, dtype=np.int32)
            db.distance_batch(q, cand=cand, fast=False)
            done += batch
            if n_pairs_hint <= 0 and time.perf_counter() - t0 >= budget:
                break
        return done, time.perf_counter() - t0

    def sample_mt(db, fast, nthreads, budget):
        # one call = the history scored against several queries' worth of candidates: every thread of the (persistent) pool
        # gets >= 2 000 pairs per call, so that waking the team is noise on a 256-core host
        tiles = max(1, (2000 * nthreads + n_hist - 1) // n_hist)
        cand = np.tile(np.arange(0, n_hist, dtype=np.int32), tiles)
        db.distance_batch_mt(n_db - 1, cand[:max(64 * nthreads, 1024)], fast, nthreads)      # makes / resizes the pool, warms the scratch
        reps, t0 = 0, time.perf_counter()
        while True:
            db.distance_batch_mt(n_db - 1 - (reps % N_EXCLUDE), cand, fast, nthreads)
            reps += 1
            if time.perf_counter() - t0 >= budget:
                break
        return reps * cand.size, time.perf_counter() - t0

    def sweep(db, budget_each):
        """reference-shaped evaluation on 64 / 128 / 256 threads (and every core of the affinity mask): the best count is the
        all-core figure; the copy-free restatement at that count."""
        counts = sorted({c for c in (64, 128, 256, threads) if c <= threads} or {threads})
        tried, best = {}, None
        for c in counts:
            pairs, dtm = sample_mt(db, False, c, budget_each)
            tried[str(c)] = pairs / dtm
            if best is None or pairs / dtm > best[1]:
                best = (c, pairs / dtm, pairs, dtm)
        c = best[0]
        pf, dtf = sample_mt(db, True, c, budget_each)
        return ({"value": best[1], "unit": "pairs/s", "cores": c, "threads_tried": tried,
                 "sample": f"{best[2]} pairs in {best[3]:.1f} s on {c} threads (persistent pool, >= 2000 pairs per thread and call, "
                           f"per-thread scratch: no allocation per pair / shift / column; the per-shift matrix copy is kept)"},
                {"value": pf / dtf, "unit": "pairs/s", "cores": c,
                 "sample": f"{pf} pairs in {dtf:.1f} s on {c} threads, copy-free restatement (norms and keys once, shifts by index)"})

    db = ob.OracleDB(cfg)
    db.save_bulk(descs[:n_db])
    done, dt = sample_1thread(db, budget_s)
    res = {"value": done / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
           "sample": f"{done} (query, keyframe) pairs of the same 64x120 workload in {dt:.1f} s: "
                     f"reference-shaped sco_distance (per-shift matrix copy, double norm evaluation) "
                     f"+ ring-key scan per query, single thread, -O3 (no -march, as the reference builds)",
           "cpu_model": _cpu_model(), "host_cores_available": os.cpu_count(), "affinity_cores": threads}
    cg0 = _cgroup_cpu()
    t_sweep = time.perf_counter()
    cpu0 = time.process_time()
    res["all_cores_reference_shaped"], res["all_cores_copy_free"] = sweep(db, 2.0)
    cpu1, wall = time.process_time(), time.perf_counter() - t_sweep
    cg1 = _cgroup_cpu()
    res["all_cores_over_one_thread"] = res["all_cores_reference_shaped"]["value"] / res["value"]
    # why the all-core figure is not (threads x the one-thread figure): CPU time this process actually got during the sweep
    res["host_cpu_share"] = {"cgroup_quota_cores": cg0.get("quota_cores"), "cores_worth_of_cpu_time_during_sweep": (cpu1 - cpu0) / wall,
                             "cgroup_throttled_periods_during_sweep": (cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0)) if "nr_throttled" in cg0 else None,
                             "cgroup_throttled_ms_during_sweep": ((cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)) / 1e3) if "throttled_usec" in cg0 else None,
                             "note": "pairs/s per core-second of CPU time = all_cores value / cores_worth_of_cpu_time"}
    db.close()
    # the same figures from a -march=native build made on this host (BASELINE.md §2's flags)
    native = _native_oracle()
    if native:
        saved_lib, saved_path = ob._lib, ob.LIB
        try:
            ob._lib, ob.LIB = None, native
            dbn = ob.OracleDB(cfg)
            dbn.save_bulk(descs[:n_db])
            done, dt = sample_1thread(dbn, 3.0)
            res["march_native_1thread"] = {"value": done / dt, "unit": "pairs/s", "cores": 1,
                                           "sample": f"{done} pairs in {dt:.1f} s, gcc -O3 -march=native -ffp-contract=off"}
            res["march_native_all_cores_reference_shaped"], res["march_native_all_cores_copy_free"] = sweep(dbn, 1.5)
            dbn.close()
            ob._lib.sco_pool_shutdown()
        finally:
            ob._lib, ob.LIB = saved_lib, saved_path
            try:
                os.unlink(native)
            except OSError:
                pass
    return res


# ------------------------------------------------------------------------------------------------
# secondary: BASELINE configs[2] -- geometric verification of the top-25 candidates of one scan
# ------------------------------------------------------------------------------------------------
def _icp_clouds(n_cand, n_pts):
    """configs[2]'s synthetic verification problem: `n_cand` structured clouds (the loop candidates' submaps) and the scan = a
    moved, noisy copy of candidate 0 (the one true loop; the other 24 do not match, as for a real top-25 list)."""
    from scl_slam_amd.synth import rigid_transform, synth_structured_cloud
    tgts, src0 = [], None
    for c in range(n_cand):
        tgt = synth_structured_cloud(n_pts, seed=100 + c, extent=60.0)
        tgts.append(tgt)
        if c == 0:
            T = rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05)
            rs = np.random.RandomState(3)
            src0 = tgt.copy()
            p = tgt[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
            src0[:, :3] = (p + 0.01 * rs.standard_normal(p.shape)).astype(np.float32)
    return tgts, src0


def cpu_baseline_icp(n_cand=25, n_pts=100000, one_thread_sample=6):
    """The ICP half of north_star's denominator ("the reference nanoflann+CPU SC-distance+ICP path"; BASELINE.md section 2 variant B:
    "thread pool over candidates / ICP problems"): oracle/icp_oracle.c's restatement of pcl::IterativeClosestPoint::align +
    getFitnessScore (DM.h:1107-1121; exact 1-NN through a uniform grid, fp64 Umeyama) on the SAME 25 x 100 k problems as
    secondary.icp_verification, both estimators.  (A) one thread, as the reference runs it (loopClosureThread, DM.h:1078-1141), on
    the first `one_thread_sample` candidates (the matching one + non-matching ones; the query figure is scaled to 25 and says so);
    (B) a pool over the 25 problems on the host cores this process may use.  Clouds are host buffers (no voxel filter: compare
    with 'host_buffers', or with 'from_store' whose 0.05 m filter keeps ~99 % of the points)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_icp_binding as oi                       # (ctypes releases the GIL inside icpo_icp_align: the pool's threads run in parallel)
    tgts, src0 = _icp_clouds(n_cand, n_pts)
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    pool_threads = max(1, min(n_cand, threads))
    cg = _cgroup_cpu()
    out = {"kind": "port", "cpu_model": _cpu_model(), "affinity_cores": threads, "cgroup_quota_cores": cg.get("quota_cores"),
           "workload": f"{n_cand} alignments of a {n_pts}-point scan against {n_pts}-point candidates, max 30 iterations (oracle/icp_oracle.c: "
                       f"icpo_icp_align, grid NN; -O3, no -march, as the reference builds)"}
    for est, name in ((1, "point_to_plane"), (0, "point_to_point")):
        def one(c, est=est):
            pp = oi.default_params(max_iterations=30, estimator=est, normal_radius=1.0)
            T, fit, conv, it = oi.icp_align(src0, tgts[c], pp)
            return it, fit, conv
        k = max(1, min(one_thread_sample, n_cand))
        t0 = time.perf_counter()
        r1 = [one(c) for c in range(k)]
        dt1 = time.perf_counter() - t0
        cpu0, t0 = time.process_time(), time.perf_counter()
        with ThreadPoolExecutor(pool_threads) as ex:
            rp = list(ex.map(one, range(n_cand)))
        dtp = time.perf_counter() - t0
        cores_used = (time.process_time() - cpu0) / dtp
        out[name] = {
            "one_thread": {"value": k / dt1, "unit": "ICP problems/s", "cores": 1, "ms_per_candidate": dt1 * 1e3 / k,
                           "ms_per_query_of_25_scaled": dt1 * 1e3 / k * n_cand, "iterations_mean": float(np.mean([r[0] for r in r1])),
                           "sample": f"candidates 0..{k - 1} of the {n_cand} (candidate 0 is the match), {dt1:.1f} s; the per-query figure is this "
                                     f"mean x {n_cand}"},
            "pool_over_problems": {"value": n_cand / dtp, "unit": "ICP problems/s", "cores": pool_threads, "ms_per_query": dtp * 1e3,
                                   "cores_worth_of_cpu_time": cores_used, "iterations_mean": float(np.mean([r[0] for r in rp])),
                                   "matching_candidate_fitness": float(rp[0][1]), "converged": int(sum(1 for r in rp if r[2])),
                                   "sample": f"all {n_cand} problems, one per thread of a {pool_threads}-thread pool, {dtp:.1f} s"}}
    return out


def secondary_icp(eng, n_cand=25, n_pts=100000):
    """One scan against its 25 loop candidates (~100 k points per cloud, point-to-plane, 30 iterations max:
    configs[2]; and point-to-point, the reference's estimator, DM.h:1108), verified together by
    scl_icp_align_batch.  Outside the headline's timed region.  The ICP roofline follows SURVEY 8(d):
    (n_src + n_tgt) * 16 B per iteration against the HBM peak."""
    tgts, src0 = _icp_clouds(n_cand, n_pts)
    out = {"workload": f"BASELINE configs[2]: one scan vs its {n_cand} loop candidates, {n_pts} points per cloud, max 30 iterations; "
                       f"'from_store': clouds resident in the on-device keyframe store (scl_loop_icp_batch_from_store: submap assembly + "
                       f"fused ICP loops, nothing but poses crosses PCIe); 'host_buffers': the same alignments with every cloud handed "
                       f"over as a pageable host buffer (scl_icp_align_batch, PCIe inclusive)"}
    ident = np.eye(4, dtype=np.float32)
    for c in range(n_cand):
        eng.keyframe_put(0, c, tgts[c])
    eng.keyframe_put(0, n_cand, src0)
    keys = np.arange(n_cand, dtype=np.int32)
    poses = np.tile(ident.reshape(1, 1, 16), (n_cand, 1, 1))
    for est, name in ((1, "point_to_plane"), (0, "point_to_point")):
        pp = eng.icp_default_params(); pp.max_iterations = 30; pp.estimator = est; pp.normal_radius = 1.0
        res = {}
        for mode in ("from_store", "host_buffers"):
            def run():
                if mode == "from_store":                      # leaf 0.05 m: loopFindNearKeyframes' voxel filter (DM.h:1183-1185) keeps ~99 % of the points and leaves them in voxel order, as every submap of the reference is
                    T, f, cv, it, ns, nt = eng.loop_icp_batch_from_store(0, n_cand, ident, keys, 0, poses, 0.05, pp)
                    return f, cv, it, ns, float(np.mean(nt))
                T, f, cv, it = eng.icp_align_batch(src0, tgts, pp)
                return f, cv, it, src0.shape[0], float(n_pts)
            run()                                             # warm-up (workspaces, streams)
            t0 = time.perf_counter()
            fb, cb, ib, ns, nt = run()
            dt = time.perf_counter() - t0
            algo_bytes = float(np.sum(ib)) * (ns + nt) * 16.0
            res[mode] = {"value": n_cand / dt, "unit": "ICP problems/s", "ms_per_query": dt * 1e3, "ms_per_candidate": dt * 1e3 / n_cand,
                         "iterations_mean": float(np.mean(ib)), "converged": int(np.sum(cb)), "matching_candidate_fitness": float(fb[0]),
                         "points_src": int(ns), "points_tgt_mean": nt,
                         "roofline": {"bound": "hbm", "achieved": algo_bytes / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": algo_bytes / dt / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": algo_bytes,
                                      "note": "SURVEY 8(d): (n_src + n_tgt) * 16 B per iteration; wall time of the whole call"}}
        out[name] = res
    # the reference's own setting (DM.h:1108-1113: point to point, setMaximumIterations(50), correspondence distance 100 m) beside
    # BASELINE's 30: BASELINE.md C3 asks for both.  From the store only (the form a host application would call).
    pp = eng.icp_default_params(); pp.max_iterations = 50; pp.estimator = 0; pp.max_correspondence_dist = 100.0
    eng.loop_icp_batch_from_store(0, n_cand, ident, keys, 0, poses, 0.05, pp)
    t0 = time.perf_counter()
    T, f, cv, it, ns, nt = eng.loop_icp_batch_from_store(0, n_cand, ident, keys, 0, poses, 0.05, pp)
    dt = time.perf_counter() - t0
    out["point_to_point_reference_settings"] = {
        "settings": "DM.h:1108-1113: point to point, max 50 iterations, correspondence distance 100 m", "ms_per_query": dt * 1e3,
        "ms_per_candidate": dt * 1e3 / n_cand, "iterations_mean": float(np.mean(it)), "iterations_max": int(np.max(it)), "converged": int(np.sum(cv)),
        "matching_candidate_fitness": float(f[0])}
    return out


# ------------------------------------------------------------------------------------------------
# secondary: the EXACT all-pairs distance matrix (every pair through the fp64 kernel), one blocking scan, survivors
# ------------------------------------------------------------------------------------------------
def secondary_exact_all_pairs(eng, n_elig, n_query, queries=64):
    """north_star's "column-shifted SC distance matrix over the keyframe database": EVERY (scan, keyframe) pair gets the
    reference's fp64 distance and shift (descriptor.h:1538-1569).  Per 16 rows: alignment + screening (which leaves, per pair, the
    first shift and the shifts within 2 eps of the pair's smallest screened distance), then sc_masked_kernel evaluates exactly
    those shifts in the reference's fp64 arithmetic.  Bit-identical to the checker (tests/test_gpu_sc_distance.py compares uint64
    views, adversarial descriptors included).  `roofline.frac` prices the bytes a group of rows MOVES (the database once per group);
    SURVEY 8(d)'s per-pair price is kept as `survey_equivalent`."""
    qs = (n_elig + (np.arange(queries) % n_query)).astype(np.int32)
    # warm-up with the timed call's own shape: the first call of a process that has two groups' copies to the host in flight pays 7 ms
    # inside the runtime's first such hipMemcpyAsync (scripts/probes/matrix_after_stream.py; a 16-row warm-up left that in the timed call)
    eng.sc_distance_matrix(qs, 0, n_elig)
    eng.profile_reset(); eng.profile_enable(2)
    t0 = time.perf_counter()
    dist, shift = eng.sc_distance_matrix(qs, 0, n_elig)
    dt = time.perf_counter() - t0
    eng.profile_enable(0)
    prof = eng.profile()
    pairs = queries * n_elig
    k_ms = prof["sc_distance_ms"] / max(1, prof["sc_distance_launches"])
    k_pairs = prof["sc_distance_pairs"] / max(1, prof["sc_distance_launches"])
    k_rows = k_pairs / n_elig
    # Bytes one group of rows MOVES by design (SURVEY 8(d)'s rule for Q scans per database pass -- "DB_bytes / Q + per-query bytes;
    # never count bytes that were not moved"): every keyframe's fp32 rows + fp64 column norms ONCE per group (the workgroups of the
    # group's scan pairs share a keyframe through an XCD's L2), the screening group's bytes per keyframe (fp16 image, alignment image,
    # ring key, mask), the same per scan, and per pair the screening intermediates, the shift mask (written, read), the first shift
    # (read again) and the result (fp64 distance + shift).
    exact_kf = R * S * 4 + S * 8
    per_pair_io = KERNEL_BYTES_PER_PAIR_IO + 4 + 4 + 4 + 12
    moved = (n_elig + k_rows) * (exact_kf + KERNEL_BYTES_PER_KEYFRAME) + k_pairs * per_pair_io
    ach = moved / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    survey = k_pairs * ALGO_BYTES_PER_PAIR / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    # what crosses L2 -> CU by design: a keyframe's fp32 rows and norms once per PAIR of scans (a workgroup of the exact kernel holds
    # two scans), the screening image once per launch
    l2_bytes = n_elig * ((k_rows + 1) // 2) * exact_kf + (n_elig + k_rows) * KERNEL_BYTES_PER_KEYFRAME + k_pairs * per_pair_io
    l2 = l2_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    flops = 3 * S * S + 13 * S * 2 * R                                                      # 242 880 per pair
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_matrix.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))                      # PMC run of scripts/bench_matrix.py (scripts/profile_matrix.sh): HBM bytes per group of 16 rows
            if abs(k_rows - tj.get("rows_per_group", 0)) < 0.5:
                traffic = tj["hbm_bytes_per_group"] * n_elig / tj["eligible_keyframes"]
        except Exception:
            traffic = None
    return {"workload": f"{queries} scans x {n_elig} keyframes, 64x120: the fp64 distance and shift of EVERY pair (scl_sc_distance_matrix, "
                        f"results copied to the host)",
            "value": pairs / dt, "unit": "pairs/s", "ms_per_scan": dt / queries * 1e3, "dtype": "f64",
            "finite_distances": int(np.isfinite(dist).sum()),
            "kernel_ms": {"group_of_rows": k_ms, "pairs_per_group": k_pairs},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_group": moved, "rows_per_group": k_rows,
                         "pricing": f"moved bytes: per keyframe {exact_kf} B (fp32 rows + fp64 norms, once per group of rows) + {KERNEL_BYTES_PER_KEYFRAME} B "
                                    f"(screening group), per pair {per_pair_io} B (screening intermediates, mask, first shift, result)",
                         "kernel": "per group of 16 rows: sc_align2_kernel + sc_screen2_kernel + sc_screen2_finish_kernel (masks from the recorded fp16 "
                                   "rounding-error norms) + sc_matrix_kernel<16,120,13> (exact fp64 at the open shifts: a workgroup holds two scans, a wave "
                                   "scores one keyframe against both); HIP events around every group",
                         "l2_delivery": {"achieved": l2, "peak": L2_RATE_GBS, "frac": l2 / L2_RATE_GBS, "bytes_per_group": l2_bytes},
                         "single_scan_hbm_floor": {"pairs_per_s": HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_PAIR, "value_over_floor": (pairs / dt) / (HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_PAIR),
                                                   "note": "a kernel that streams the fp32 database once per SCAN at 8 TB/s (SURVEY 8(d)'s Q = 1 floor)"},
                         "survey_equivalent": {"bytes_per_pair": ALGO_BYTES_PER_PAIR, "achieved": survey, "frac": survey / HBM_PEAK_GBS,
                                               "fp64_vector_frac": (k_pairs * flops / (k_ms * 1e-3) / 1e12 / 78.6) if k_ms > 0 else 0.0,
                                               "note": "SURVEY 8(d)'s single-scan price (4 R S + 8 S bytes and 242 880 flop per pair) x pairs / time: "
                                                       "round 3's `frac`; NOT a rate of moved bytes or of evaluated flops (1-3 of the 13 shifts are evaluated)"}}}


def secondary_blocking_scan(eng, n_elig, n_query, scans=300):
    """SURVEY 8(d)'s headline shape and the reference's own call pattern (descriptor.h:1613-1674 from distributedMapping.h:1078): ONE
    incoming scan, blocking -- scl_detect_full_range call -> result on the host.  Two launches: the products in their first form with
    the workgroups aligning their own groups (fastAlignUsingVkey exactly + fp16 matrix-core screening of the 13 shifts: the database
    streamed once), then one workgroup that lists the keyframes within the margin, scores them at their open shifts in fp64, forms the
    ring-key top-k and writes the winner to pinned memory.  Also the reference-faithful call (ring-key top-3 -> 3 SC distances ->
    threshold: scl_detect_intra), as the loop-closure thread makes it."""
    def run(fn, reps):
        lat = []
        for i in range(reps + 20):
            t0 = time.perf_counter()
            fn(i)
            if i >= 20:
                lat.append((time.perf_counter() - t0) * 1e6)
        return np.array(lat)
    lat = run(lambda i: eng.detect_full_range(int(n_elig + i % n_query), 0, n_elig), scans)
    li = run(lambda i: eng.detect_intra(int(n_elig + i % n_query)), scans)
    p50 = float(np.percentile(lat, 50))
    moved = n_elig * SINGLE_SCAN_BYTES_PER_PAIR                      # the fp16 copy, keys, ring key and mask of every keyframe, once
    return {"p50": p50, "p99": float(np.percentile(lat, 99)), "mean": float(lat.mean()), "scans": scans,
            "pairs_per_s_at_p50": n_elig / (p50 * 1e-6),
            "roofline": {"bound": "hbm", "achieved": moved / (p50 * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": moved / (p50 * 1e-6) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": moved, "bytes_per_pair": SINGLE_SCAN_BYTES_PER_PAIR,
                         "hbm_floor_us": moved / (HBM_PEAK_GBS * 1e9) * 1e6,
                         "note": "wall clock of the blocking call (launch latency, two kernels, result pick-up) against the bytes one scan's pass streams"},
            "detect_intra_us": {"p50": float(np.percentile(li, 50)), "p99": float(np.percentile(li, 99)),
                                "note": "the reference-faithful call through the six virtuals: ring-key top-3, 3 exact SC distances, threshold (D.h:1613-1674)"},
            "note": "microseconds per blocking one-scan call over the whole database (no batch to share the database pass with)"}


def secondary_adversarial_survivors(device, shard, queries, n_elig, frac=0.05, scans=256):
    """The headline rate holds while few keyframes survive the screening.  Here `frac` of the database are noisy rolled copies
    of the incoming scan -- all within the screening margin (2 x 1.5e-3) of the winner -- so the exact pass scores that many
    keyframes per scan.  Same pass, same bit-exact winner; the rate is what a place with many near-identical views costs."""
    from scl_slam_amd import ScanContextEngine
    rs = np.random.RandomState(23)
    db = shard.copy()
    base = queries[1].copy()                                              # (query 1 is not a planted revisit)
    planted = rs.choice(n_elig, size=int(n_elig * frac), replace=False)
    for j in planted:
        d = np.roll(base, int(rs.randint(0, S)), axis=1)
        db[j] = np.clip(d + np.float32(rs.uniform(1e-4, 2e-3)) * rs.standard_normal(d.shape).astype(np.float32) * (d > 0), 0, None)
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=N_EXCLUDE, device=device, initial_capacity=n_elig + 64)
    eng.save_bulk(db); eng.save_bulk(queries[:N_EXCLUDE])
    del db
    qs = np.full(scans, n_elig + 1, dtype=np.int32)
    eng.detect_full_stream(qs[:32], 0, n_elig, 16, 2)
    eng.survivor_stats(reset=True)
    t0 = time.perf_counter()
    nn, sh, dd = eng.detect_full_stream(qs, 0, n_elig, 16, 2)
    dt = time.perf_counter() - t0
    q, tot, mx = eng.survivor_stats()
    eng.close()
    return {"workload": f"{int(frac * 100)} % of the {n_elig} keyframes within 2 eps of the winner ({len(planted)} planted near-copies of the scan), {scans} scans",
            "value": n_elig * scans / dt, "unit": "pairs/s", "ms_per_scan": dt / scans * 1e3,
            "survivors_per_scan": {"mean": tot / max(1, q), "max": mx}, "winner_distance": float(dd[0]), "winner_is_planted": bool(int(nn[0]) in set(planted.tolist()))}


# ------------------------------------------------------------------------------------------------
# secondary: the front half of the per-incoming-scan path -- raw points -> descriptor -> database slot (K3), ring-key scan (K2)
# ------------------------------------------------------------------------------------------------
def _distinct_scan(base, i):
    """scan i of a stream: a base cloud with its own heights on a third of the points -- a different descriptor (scans that repeat are
    exact duplicates of keyframes already in the database: dozens of survivors at distance 0 per scan, the adversarial case)"""
    rs = np.random.RandomState(9000 + i)
    c = base.copy()
    m = rs.rand(c.shape[0]) < 0.3
    c[m, 2] += rs.uniform(0.0, 3.0, int(m.sum())).astype(np.float32)
    return c


def _p50_us(fn, reps, warm=3):
    lat = []
    for i in range(reps + warm):
        t0 = time.perf_counter()
        fn()
        if i >= warm:
            lat.append((time.perf_counter() - t0) * 1e6)
    return float(np.percentile(lat, 50))


def secondary_ingest_per_scan(device, shapes=((64, 120, 120000), (80, 180, 240000)), batch=16):
    """K3 (makeScancontext D.h:1404-1461 + save D.h:1587-1602) per incoming scan of 16-byte point records: device time of the two
    launches a batch of 16 scans costs (HIP events around the scatter over all 16 clouds and around the ingest of their 16 slots),
    the call's wall clock from pageable and from pinned host memory, one scan at a time (scl_make_and_save) and sixteen
    (scl_make_and_save_many).  Roofline by SURVEY 8(d): n x 16 B in + R x S x 4 B out per scan over the kernels' time."""
    from scl_slam_amd import ScanContextEngine
    from scl_slam_amd.synth import synth_scan
    out = {}
    for (R2, S2, npts) in shapes:
        eng = ScanContextEngine(num_ring=R2, num_sector=S2, device=device, initial_capacity=8192)
        clouds = [np.ascontiguousarray(synth_scan(npts, seed=300 + i, stride_floats=4)) for i in range(batch)]
        pinned = []
        for c in clouds:
            a = eng.host_alloc(c.shape); a[:] = c; pinned.append(a)
        one_pageable = _p50_us(lambda: eng.make_and_save(clouds[0], 0, 0), 30)
        one_pinned = _p50_us(lambda: eng.make_and_save(pinned[0], 0, 0), 30)
        many_pageable = _p50_us(lambda: eng.make_and_save_many(clouds, want_values=False), 8, 2)
        many_pinned = _p50_us(lambda: eng.make_and_save_many(pinned, want_values=False), 12, 2)
        eng.profile_reset(); eng.profile_enable(1)
        for _ in range(10):
            eng.make_and_save_many(pinned, want_values=False)
        eng.profile_enable(0)
        prof = eng.profile()
        sc_us = prof["make_sc_ms"] / max(1, prof["make_sc_launches"]) * 1e3
        ing_us = prof["ingest_ms"] / max(1, prof["ingest_launches"]) * 1e3
        dev_us = sc_us + ing_us
        algo = batch * (npts * 16 + R2 * S2 * 4)
        ach = algo / (dev_us * 1e-6) / 1e9 if dev_us > 0 else 0.0
        ach_sc = batch * npts * 16 / (sc_us * 1e-6) / 1e9 if sc_us > 0 else 0.0
        eng.close()
        out[f"{R2}x{S2}_{npts // 1000}k_points"] = {
            "device_us_per_scan_in_a_batch_of_16": dev_us / batch, "device_us_per_batch": {"scatter_16_clouds": sc_us, "ingest_16_slots": ing_us},
            "call_us_per_scan": {"one_scan_pageable": one_pageable, "one_scan_pinned": one_pinned,
                                 "batch_of_16_pageable": many_pageable / batch, "batch_of_16_pinned": many_pinned / batch},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_batch": algo, "scatter_only": {"achieved": ach_sc, "frac": ach_sc / HBM_PEAK_GBS},
                         "pricing": "SURVEY 8(d) K3: n_pts x 16 B in + R x S x 4 B out per scan, x 16 scans, over scatter + ingest (HIP events on the engine's stream)",
                         "kernel": "make_sc_batch_scatter_kernel (workgroup = 4 096 points of one cloud, private LDS polar tile, atomicMax merge) + "
                                   "ingest_kernel (one workgroup per scan: finalize, keys, norms, fp32 / fp16 / alignment images of the slot)"}}
    return out


def secondary_ringkey_topk(eng, n_elig, n_query, reps=60):
    """K2: the exact ring-key scan (nanoflann's metric, NF:383-408) over the eligible keyframes -- ONE launch (distance + per-workgroup
    top-k + last-workgroup merge).  Roofline by SURVEY 8(d): N x R x 4 B per query."""
    res = {}
    for k in (3, 25):
        for i in range(5):
            eng.ringkey_topk(int(n_elig + i), 0, n_elig, k)
        eng.profile_reset(); eng.profile_enable(1)
        for i in range(reps):
            eng.ringkey_topk(int(n_elig + i % n_query), 0, n_elig, k)
        eng.profile_enable(0)
        prof = eng.profile()
        us = prof["ringkey_topk_ms"] / max(1, prof["ringkey_topk_launches"]) * 1e3
        call = _p50_us(lambda: eng.ringkey_topk(int(n_elig + 7), 0, n_elig, k), 40)
        algo = n_elig * R * 4
        ach = algo / (us * 1e-6) / 1e9 if us > 0 else 0.0
        res[f"k{k}"] = {"device_us": us, "call_us_p50": call,
                        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes": algo,
                                     "hbm_floor_us": algo / (HBM_PEAK_GBS * 1e9) * 1e6,
                                     "note": "2.5 MB per query: a launch is latency bound (one wave per 64 slots, sixteen 16-byte loads in flight per lane)"}}
    return res


def secondary_stream_from_points(device, n=N_KEYFRAMES_1GPU, npts=120000, n_scans=256, batch=16):
    """The per-incoming-scan pipeline from raw points in one call (scl_stream_from_points): per scan 120 k points of 16 B from PINNED
    host memory -> descriptor -> database slot -> full-database arg-min detection over [0, key - 100), on a 10k-keyframe database.
    PCIe is the floor: the clouds of group g + 1 are copied while group g is binned, ingested and searched for."""
    from scl_slam_amd import ScanContextEngine
    from scl_slam_amd.synth import synth_descriptors, synth_scan
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=N_EXCLUDE, device=device, initial_capacity=n + 2 * n_scans + 256)
    eng.save_bulk(synth_descriptors(n, R, S, seed=1002))
    bases = [synth_scan(npts, seed=500 + i, stride_floats=4) for i in range(batch)]
    # every scan its own cloud, one behind the other in ONE pinned arena (a ring of incoming scans): a group of 16 then travels as one copy
    arena = eng.host_alloc((n_scans + 2 * batch, npts, 4))
    for i in range(n_scans + 2 * batch):
        arena[i] = _distinct_scan(bases[i % batch], i)
    pinned = [arena[i] for i in range(n_scans + 2 * batch)]
    separate = []                                                            # ... and 64 scans in pinned buffers of their own (one copy per cloud, two copy streams)
    for i in range(4 * batch):
        a = eng.host_alloc(bases[0].shape); a[:] = arena[i]; separate.append(a)
    pageable = [np.array(a) for a in pinned[:4 * batch]]
    warm, seq = pinned[n_scans:], pinned[:n_scans]
    eng.stream_from_points(warm)                                             # warm-up: buffers, streams, the stream form's sets
    n_before = eng.get_size()
    t0 = time.perf_counter()
    nn, sh, dd = eng.stream_from_points(seq)
    dt = time.perf_counter() - t0
    pairs = int(sum(max(0, n_before + i - N_EXCLUDE) for i in range(n_scans)))
    seq_p = pageable
    t0 = time.perf_counter()
    eng.stream_from_points(seq_p)
    dt_p = time.perf_counter() - t0
    eng.stream_from_points(separate[:2 * batch])
    t0 = time.perf_counter()
    eng.stream_from_points(separate)
    dt_s = time.perf_counter() - t0
    h2d = eng.host_copy_rate(64 << 20, 8)            # one large pinned copy at a time, HIP events on the engine's copy stream
    h2d_cloud = eng.host_copy_rate(npts * 16, 64)    # ... and copies of one cloud's size back to back on ONE stream
    bytes_scan = npts * 16
    floor_us = bytes_scan / (h2d * 1e9) * 1e6
    eng.close()
    return {"workload": f"{n_scans} scans of {npts} points (16-byte records, one behind the other in a pinned arena) through scl_stream_from_points on a {n}-keyframe 64x120 database: "
                        f"descriptor + append + full-database detection per scan, groups of {batch}",
            "scans_per_s": n_scans / dt, "us_per_scan": dt / n_scans * 1e6, "value": pairs / dt, "unit": "pairs/s",
            "winners_found": int((nn >= 0).sum()),
            "pcie": {"h2d_GBps_measured": h2d, "h2d_GBps_cloud_sized_copies_one_stream": h2d_cloud, "bytes_per_scan": bytes_scan, "floor_us_per_scan": floor_us, "us_per_scan_over_floor": dt / n_scans * 1e6 / floor_us,
                     "note": "floor = bytes per scan / the rate one large pinned hipMemcpyAsync reaches on this box"},
            "pinned_buffer_per_cloud": {"us_per_scan": dt_s / len(separate) * 1e6, "scans": len(separate), "us_per_scan_over_floor": dt_s / len(separate) * 1e6 / floor_us,
                                        "note": "every cloud in a pinned allocation of its own: one copy per cloud (17 us of start-up each), dealt over two copy streams"},
            "pageable_host_memory": {"us_per_scan": dt_p / len(seq_p) * 1e6, "scans": len(seq_p),
                                     "note": "the same call from ordinary (pageable) buffers: the runtime stages every copy through its own pinned memory"}}


def secondary_stream_from_resident_points(device, n=N_KEYFRAMES_1GPU, npts=120000, n_scans=512):
    """The per-incoming-scan path of BASELINE configs[1] with its INPUTS RESIDENT IN HBM (the clouds in the on-device keyframe store):
    per scan descriptor build (K3) + append + ring-key scan (K2) + shifted SC distance against every eligible keyframe + arg-min (K1),
    through scl_stream_from_store.  This is the rate `value` would be if its timed region started at the raw points instead of at the
    ingested keyframe."""
    from scl_slam_amd import ScanContextEngine
    from scl_slam_amd.synth import synth_descriptors, synth_scan
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=N_EXCLUDE, device=device, initial_capacity=n + 3 * n_scans + 128)
    eng.save_bulk(synth_descriptors(n, R, S, seed=1002))
    bases = [np.ascontiguousarray(synth_scan(npts, seed=700 + i, stride_floats=4)) for i in range(16)]
    for i in range(2 * n_scans + 64):
        eng.keyframe_put(0, i, _distinct_scan(bases[i % 16], i))         # every keyframe its own cloud
    eng.stream_from_store(0, 0, n_scans)                                    # warm-up (buffers, the stream form's sets)
    n_before = eng.get_size()
    t0 = time.perf_counter()
    nn, sh, dd = eng.stream_from_store(0, n_scans, n_scans)
    dt = time.perf_counter() - t0
    eng.profile_reset(); eng.profile_enable(1)                              # the front's share by HIP events, in a call of its own (an event pair per launch slows the call)
    eng.stream_from_store(0, 2 * n_scans, 64)
    eng.profile_enable(0)
    prof = eng.profile()
    eng.close()
    pairs = int(sum(max(0, n_before + i - N_EXCLUDE) for i in range(n_scans)))
    front_us = (prof["make_sc_ms"] + prof["ingest_ms"]) * 1e3 / 64
    # bytes by SURVEY 8(d): K3 per scan + K1's launch groups (the database once per 16 scans) + K2 per scan
    groups = n_scans / 16.0
    algo = n_scans * (npts * 16 + R * S * 4) + groups * (n + n_scans / 2) * KERNEL_BYTES_PER_KEYFRAME + pairs * KERNEL_BYTES_PER_PAIR_IO
    return {"workload": f"{n_scans} scans of {npts} points resident in HBM (keyframe store) through scl_stream_from_store on a {n}-keyframe 64x120 database: "
                        f"descriptor + append + full-database detection per scan",
            "scans_per_s": n_scans / dt, "us_per_scan": dt / n_scans * 1e6, "value": pairs / dt, "unit": "pairs/s", "winners_found": int((nn >= 0).sum()),
            "front_device_us_per_scan": front_us,
            "roofline": {"bound": "hbm", "achieved": algo / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / dt / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": algo,
                         "pricing": "SURVEY 8(d): K3 n x 16 + R x S x 4 B per scan, + the screening launch groups' bytes (the database once per 16 scans, per-pair "
                                    "intermediates) over the wall clock of the whole call"}}


# ------------------------------------------------------------------------------------------------
# secondary: the SC-distance pass on BASELINE configs[4]'s grid (80 x 180), 10k keyframes
# ------------------------------------------------------------------------------------------------
def secondary_80x180(device, n=10000, steps=512):
    """pairs/s of the full-DB pass on the 80x180 grid of configs[4] (Livox): alignment + screening products (W = 19 shifts in two
    passes) + exact pass on the survivors, sixteen scans per launch group."""
    from scl_slam_amd import ScanContextEngine
    from scl_slam_amd.synth import synth_descriptors
    R2, S2 = 80, 180
    descs = synth_descriptors(n, R2, S2, seed=1005, revisit_frac=0.0)
    eng = ScanContextEngine(num_ring=R2, num_sector=S2, num_exclude_recent=N_EXCLUDE, device=device, initial_capacity=n + 64)
    eng.save_bulk(descs)
    del descs
    n_elig = n - N_EXCLUDE
    qs = (n_elig + (np.arange(steps) % N_EXCLUDE)).astype(np.int32)
    eng.detect_full_stream(qs[:8], 0, n_elig, 16, 2)
    eng.profile_reset(); eng.profile_enable(3)
    eng.survivor_stats(reset=True)
    t0 = time.perf_counter()
    nn, sh, dd = eng.detect_full_stream(qs, 0, n_elig, 16, 2)
    dt = time.perf_counter() - t0
    eng.profile_enable(0)
    prof = eng.profile()
    sv_q, sv_sum, sv_max = eng.survivor_stats()
    # every pair's exact fp64 distance and shift on this grid (scl_sc_distance_matrix: 64 scans against the database, as
    # secondary.exact_all_pairs on 64x120), median of three calls
    mq = qs[:64]
    eng.sc_distance_matrix(mq[:16], 0, n_elig)
    mts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.sc_distance_matrix(mq, 0, n_elig); mts.append(time.perf_counter() - t0)
    matrix_pairs_per_s = len(mq) * n_elig / sorted(mts)[1]
    eng.close()
    survey_pair = R2 * S2 * 4 + S2 * 4 + S2 * 4                                 # 59 040 B at 80x180
    # what the launch group (products in their second form + finish + next alignment) reads per keyframe by design, once per
    # launch of 16 scans: the chunk-major fp16 image (5 ring slices of 16 x 2 chunks x S sectors x 16 B: the 80 rings, no padding;
    # the sectors a fragment load repeats past the last come out of the L2), the first part of the alignment image (4 rotated copies of
    # the fp16 sector key x 384 B + norm), the tiled ring key, the sector mask; per pair: first shift, the five slices' packed
    # partial sums (13 + 6 shift rows in 16 + 8 floats, written and read), bound, ring-key metric
    bytes_kf = 5 * 2 * S2 * 16 + (4 * 192 * 2 + 16) + 4 * 4 * ((R2 + 3) // 4) + 32
    bytes_pair_io = 4 + 2 * 4 + 2 * (5 * 24 * 4) + 4 + 4
    k_ms = prof["sc_distance_ms"] / max(1, prof["sc_distance_launches"])
    k_pairs = prof["sc_distance_pairs"] / max(1, prof["sc_distance_launches"])
    k_scans = k_pairs / n_elig
    per_launch = (n_elig + k_scans) * bytes_kf + k_pairs * bytes_pair_io        # DB once per launch + per-scan bytes + per-pair intermediates
    ach = per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    return {"workload": f"{n} synthetic keyframes, 80x180 SC (BASELINE configs[4]'s grid), full ring-key + shifted SC distance (19 shifts) "
                        f"over the whole DB per scan, {steps} scans",
            "value": n_elig * steps / dt, "unit": "pairs/s", "ms_per_scan": dt / steps * 1e3,
            "kernel_ms": {"screening_launch_group": k_ms},
            "survivors_per_scan": {"mean": sv_sum / max(1, sv_q), "max": sv_max, "scans": sv_q},
            "exact_all_pairs": {"value": matrix_pairs_per_s, "unit": "pairs/s", "rows": int(len(mq)),
                                "note": "scl_sc_distance_matrix through the call, results in host memory; groups of 16 rows alternate between two streams"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": per_launch, "scans_per_launch": k_scans,
                         "bytes_per_keyframe": bytes_kf, "bytes_per_pair_intermediates": bytes_pair_io, "survey_bytes_per_pair": survey_pair,
                         "kernel": "screening launch group of the 80x180 grid: sc_screen2_kernel<20,180,19> (five ring slices of 16, two sectors per "
                                   "k-step, 19 shifts in two passes that share the scans' fragments, 16 scans per launch) + finish + alignment of "
                                   "the next group; one event pair around every second chunk's launch groups, its time divided by their number"}}


# ------------------------------------------------------------------------------------------------
# secondary: BASELINE configs[4]-shaped stream on ONE GPU -- dense Livox-like scans, 80x180, ICP verification
# ------------------------------------------------------------------------------------------------
def secondary_livox_stream(device, n0=1000, n_scans=600):
    """End-to-end latency per incoming scan of ~240 k points handed over as a host buffer (PCIe inclusive): voxel filter ->
    80x180 descriptor -> append (one call) -> reference-faithful detection (top-10 + SC distance) -> full-database pass ->
    ICP verification of a loop candidate (point-to-point, <= 30 iterations, 50 k vs 100 k points).  The database grows from
    1 000 keyframes; configs[4] names 8 GPUs and 10 Hz -- this is the one-GPU figure against the 100 ms budget."""
    from scl_slam_amd import ScanContextEngine
    from scl_slam_amd.synth import rigid_transform, synth_descriptors, synth_scan, synth_structured_cloud
    R2, S2 = 80, 180
    eng = ScanContextEngine(num_ring=R2, num_sector=S2, num_candidates=10, device=device, initial_capacity=2048)
    eng.save_bulk(synth_descriptors(n0, R2, S2, seed=1005))
    scans = [synth_scan(240000, seed=100 + i) for i in range(6)]           # reused round robin
    submap = synth_structured_cloud(100000, seed=7, extent=60.0)
    Tinv = np.linalg.inv(rigid_transform(0.004, -0.006, 0.02, 0.25, -0.15, 0.05))
    src = submap[::2].copy()
    src[:, :3] = (submap[::2, :3].astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]).astype(np.float32)
    p = eng.icp_default_params(); p.max_iterations = 30
    lat, stage = [], {"voxel+descriptor+append": [], "detect_topk": [], "detect_full_db": [], "icp": []}
    for i in range(n_scans + 5):
        t0 = time.perf_counter()
        eng.make_and_save_filtered(scans[i % len(scans)], 0.4, 0, n0 + i)   # makeDescriptors, DM.h:996-1002
        t1 = time.perf_counter()
        eng.detect_intra(n0 + i)
        t2 = time.perf_counter()
        eng.detect_full(n0 + i)
        t3 = time.perf_counter()
        _, fit, conv, iters = eng.icp_align(src, submap, p)
        t4 = time.perf_counter()
        if i >= 5:
            lat.append((t4 - t0) * 1e3)
            for k, v in zip(stage, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                stage[k].append(v * 1e3)
    eng.close()
    lat = np.array(lat)
    return {"workload": f"BASELINE configs[4]-shaped, one GPU: {n_scans} scans of 240000 points (host buffers; BASELINE.md C5: 600 scans at 10 Hz), 80x180 SC, database "
                        f"growing from {n0} keyframes; per scan voxel filter + descriptor + append, top-10 detection, full-database pass, ICP (p2p, <= 30 it, 50k vs 100k)",
            "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
            "budget_ms_at_10hz": 100.0, "dropped_frames": int((lat > 100.0).sum()),
            "stage_ms_p50": {k: float(np.percentile(v, 50)) for k, v in stage.items()},
            "icp": {"iterations": int(iters), "converged": bool(conv), "fitness": float(fit)},
            "scans_per_s": 1e3 / float(np.mean(lat)), "missed_10hz_budget": int((lat > 100.0).sum())}


# ------------------------------------------------------------------------------------------------
# the C-ABI sharded front (one process, all GPUs of the node): SURVEY 8(e) through scl_create_sharded
# ------------------------------------------------------------------------------------------------
def bench_front(G, steps, spl):
    """BASELINE configs[3] through the C ABI: ONE engine over G shards (keyframe g on shard g % G; devices = the visible GPUs, round
    robin, so on a one-GPU box the shards share the card and the exchange is the host merge -- or, SCL_FRONT_EXCHANGE=2 with
    SCL_RCCL_LIB=tests/cpp/libmock_rccl.so, the tests' stand-in collective; with G distinct devices the two RCCL min all-reduces).  Stream form (per-shard streams + host merge of the
    winners) and the single blocking pass (per-shard pass + exchange)."""
    import torch
    from scl_slam_amd import ScanContextEngine
    from scl_slam_amd.synth import synth_descriptors
    ndev = max(1, torch.cuda.device_count())
    devices = [i % ndev for i in range(G)]
    distinct = len(set(devices)) == G
    exch = int(os.environ.get("SCL_FRONT_EXCHANGE", "2" if (distinct and G > 1) else "1"))
    n = N_KEYFRAMES_SHARD * G
    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=N_EXCLUDE, initial_capacity=n + 64, devices=devices, exchange=exch)
    for c in range(G):
        eng.save_bulk(synth_descriptors(N_KEYFRAMES_SHARD, R, S, seed=1002 + 7919 * c, revisit_frac=0.0))
    n_elig = n - N_EXCLUDE
    scans = steps * spl
    qs = (n_elig + (np.arange(scans) % N_EXCLUDE)).astype(np.int32)
    eng.detect_full_stream(qs[:2 * spl], 0, n_elig, spl, 2)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        eng.detect_full_stream(qs, 0, n_elig, spl, 2)
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    lat = []
    for i in range(60):
        t0 = time.perf_counter()
        eng.detect_full_range(int(qs[i]), 0, n_elig)
        if i >= 10:
            lat.append((time.perf_counter() - t0) * 1e6)
    info = eng.shard_info()
    eng.close()
    print(json.dumps({"metric": "loop-closure candidates/sec through the C-ABI sharded front (scl_create_sharded), one process", "value": n_elig * scans / dt,
                      "unit": "pairs/s", "n_gpus": len(set(devices)), "shards": G, "devices": devices, "exchange": {1: "host merge", 2: "min all-reduce x2 through " + (os.environ.get("SCL_RCCL_LIB") or "librccl")}.get(info[1], str(info[1])),
                      "steps": steps, "ms_per_step": dt / steps * 1e3, "ms_per_scan": dt / scans * 1e3, "higher_is_better": True, "scaling": "weak",
                      "config": {"workload": f"BASELINE configs[3]-shaped: {n} keyframes over {G} shards ({N_KEYFRAMES_SHARD} each), 64x120, {scans} scans through the front's stream form",
                                 "scans_per_launch": spl},
                      "blocking_pass_us": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                                           "note": "one scan: per-shard passes enqueued on every device, winners reduced by the exchange above"}}), flush=True)


def main():
    args = parse_args()
    if args.dry_run:
        sys.exit(dry_run(args))
    if args.front > 0:
        return bench_front(args.front, args.steps, max(1, args.scans_per_launch))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}; launch with "
                         f"`python bench.py --gpus N` or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # SCL_BENCH_SHARE_GPU=1: rehearsal of the multi-rank path on a one-GPU box (all ranks on cuda:0, gloo
    # for the exchange); the real runs use one GPU per rank and RCCL ("nccl").
    share_gpu = os.environ.get("SCL_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cpu" if share_gpu else "cuda"
    if world > 1:
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from scl_slam_amd import ScanContextEngine
    from scl_slam_amd.synth import synth_descriptors

    n_local = args.keyframes or (N_KEYFRAMES_1GPU if world == 1 else N_KEYFRAMES_SHARD)
    n_query = N_EXCLUDE                                  # the newest 100 keyframes are the queries
    n_elig = n_local - n_query
    # shard of this rank (its own trajectory segment) + the shared query keyframes at the end
    shard = synth_descriptors(n_elig, R, S, seed=1002 + 7919 * rank)
    queries = synth_descriptors(n_query, R, S, seed=424242, revisit_frac=0.0)
    if rank == 0:
        # plant rotated noisy copies of shard keyframes so some queries are true loops
        rs = np.random.RandomState(5)
        for i in range(0, n_query, 4):
            src = int(rs.randint(0, n_elig))
            queries[i] = np.roll(shard[src], int(rs.randint(0, S)), axis=1)
    if world > 1:
        qt = torch.from_numpy(queries).to(coll_dev)
        dist.broadcast(qt, src=0)
        queries = qt.cpu().numpy()

    eng = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=N_EXCLUDE,
                            device=local_rank, initial_capacity=n_local + 64)
    eng.save_bulk(shard)
    eng.save_bulk(queries)
    assert eng.get_size() == n_local

    from scl_slam_amd.sharded import FullScanStream

    _slots = {}

    def query_slots(first, count):
        """database slots of the scans first .. first + count - 1 (the scans are keyframes resident in HBM; the index array is built once,
        outside the timed region)"""
        key = (first, count)
        if key not in _slots:
            _slots[key] = np.ascontiguousarray((n_elig + ((first + np.arange(count)) % n_query)).astype(np.int32))
        return _slots[key]

    def run(first, count):
        """`count` steps.  Each step = one scan's full pass over this rank's shard (ring-key top-k + SC distance
        + arg-min, one launch); `--pipeline` passes are in flight, and for N > 1 the per-rank winners of a chunk
        of scans travel in one asynchronous exchange (RCCL), merged one batch later."""
        q = query_slots(first, count)
        if world == 1 and args.native_chunk > 0:
            # one GPU: the C ABI's own stream call, arrays of scans in, arrays of winners out (scl_detect_full_stream), `native_chunk`
            # scans per call -- FullScanStream, the merge of several shards' winners, has nothing to merge here, and turning 320
            # winners into Python tuples cost the driver's 20-step run 3 % of its time
            out = []
            for s0 in range(0, count, args.native_chunk):
                nn, sh, dd = eng.detect_full_stream(q[s0:s0 + args.native_chunk], 0, n_elig, args.scans_per_launch, args.pipeline)
                out.append((dd, nn, sh))
            assert sum(len(o[1]) for o in out) == count
            return out
        st = FullScanStream(eng, rank, world, device=coll_dev, depth=args.pipeline, merge_every=args.merge_every,
                            scans_per_launch=args.scans_per_launch, native_chunk=args.native_chunk, exchange=args.exchange)
        st.submit_many(q, 0, n_elig)
        res = st.drain()
        assert len(res) == count
        return res

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    spl = max(1, args.scans_per_launch)                      # scans of one step (= of one launch group)
    warm_scans, timed_scans = args.warmup * spl, args.steps * spl
    run(0, warm_scans)
    query_slots(warm_scans, timed_scans)
    eng.profile_reset()
    eng.survivor_stats(reset=True)
    eng.profile_enable(3)          # HIP events around the dominant kernel, one launch in thirteen (an event pair keeps two launches back by 4-6 us each)
    times = []
    for rep in range(max(1, args.repeats)):
        fence()
        t0 = time.perf_counter()
        timed_results = run(warm_scans, timed_scans)
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        times.append(elapsed)
        # outside the timed region: every planted revisit (query 4j = a rolled copy of one of rank 0's keyframes) must
        # have been found by the merged result, on every rank
        if timed_results and isinstance(timed_results[0][0], np.ndarray):      # the one-GPU form's arrays, call by call
            flat = []
            for dd_, nn_, sh_ in timed_results:
                flat.extend(zip(dd_.tolist(), nn_.tolist(), sh_.tolist()))
            timed_results = flat
        for i, (d, g, sh) in enumerate(timed_results):
            if os.environ.get("SCL_ABLATE"):             # diagnostic build with phases switched off: results are wrong on purpose
                break
            if ((warm_scans + i) % n_query) % 4 == 0:
                assert d < 1e-6 and g >= 0 and g % world == 0, (i, d, g, sh)
    eng.profile_enable(False)
    prof = eng.profile()
    al_pairs, al_fallbacks = eng.alignment_stats()
    sv_q, sv_sum, sv_max = eng.survivor_stats()
    elapsed = float(np.median(times))

    pairs_per_step = n_elig * world * spl
    value = pairs_per_step * args.steps / elapsed
    k1_ms = prof["sc_distance_ms"] / max(1, prof["sc_distance_launches"])
    k1_pairs = prof["sc_distance_pairs"] / max(1, prof["sc_distance_launches"])
    k1_scans = k1_pairs / n_elig                          # scans whose products one launch group holds
    # Algorithmic bytes of one launch group by SURVEY 8(d)'s rule for Q scans per database pass ("DB_bytes / Q + per-query bytes:
    # never count bytes that were not moved"): every eligible keyframe's screening image and keys once, + per scan its own
    # images, + what the group writes and re-reads per pair (first shifts, partial sums, bound, ring-key metric).
    per_launch = (n_elig + k1_scans) * KERNEL_BYTES_PER_KEYFRAME + k1_pairs * KERNEL_BYTES_PER_PAIR_IO
    achieved = per_launch / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
    survey_equiv = (ALGO_BYTES_PER_PAIR * k1_pairs) / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
    single_floor = HBM_PEAK_GBS * 1e9 / SINGLE_SCAN_BYTES_PER_PAIR

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_sc_distance.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))                      # PMC run of this same command (scripts/profile_k1.sh)
            # HBM bytes of a launch follow the database (read once per launch), not the pair count: valid for the profiled
            # launch shape only
            if abs(k1_scans - tj.get("scans_per_launch", 0)) < 0.5:
                traffic = tj["hbm_bytes_per_launch"] * n_elig / tj["eligible_keyframes"]
        except Exception:
            traffic = None

    if rank == 0:
        workload = ("BASELINE configs[1]: 10k synthetic Velodyne-64 keyframes, 64x120 SC, full ring-key scan + shifted SC "
                    "distance over the whole DB per incoming scan") if world == 1 else \
                   (f"BASELINE configs[3]: keyframe database sharded by keyframe index over {world} GPUs, {n_local} keyframes per "
                    f"GPU ({n_local * world} in total; 100k at 8 GPUs), 64x120 SC, full ring-key scan + shifted SC distance over "
                    f"every shard per incoming scan, RCCL min all-reduce on the per-scan (distance, index, shift) winners")
        out = {
            "metric": "loop-closure candidates/sec: (scan, keyframe) pairs taken through full-database ARG-MIN detection per second "
                      "(every pair aligned exactly and bounded by the f16 matrix-core screening; only the survivors get the f64 distance), "
                      "batches of 16 scans, 10k-keyframe DB per GPU",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_scan": elapsed / timed_scans * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 results; f16-operand / f32-accumulate MFMA screening + f64 exact pass on the survivors", "data": "synthetic",
            "repeats": len(times), "ms_per_step_min": min(times) / args.steps * 1e3, "ms_per_step_max": max(times) / args.steps * 1e3,
            "config": {"workload": workload,
                       "step": f"one batch of {spl} incoming scans, each scored against the whole database (one launch group)",
                       "scans_per_step": spl, "scans_timed": timed_scans,
                       "keyframes_per_gpu": n_local, "eligible_per_query": n_elig, "rings": R, "sectors": S,
                       "shifts_per_pair": 13, "scans_per_launch": args.scans_per_launch, "launches_in_flight": args.pipeline, "native_chunk": args.native_chunk,
                       "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "4 (HIP's default)"),
                       "sharding": (f"keyframe-index shards x{world}; exchange = {args.exchange} "
                                    f"({'two 8-byte min all-reduces' if args.exchange == 'allreduce' else 'one 24-byte all-gather'} "
                                    f"per scan, batched over {args.native_chunk or args.merge_every} scans, asynchronous)") if world > 1 else "none (one GPU)"},
            # BASELINE's second figure, "SC-distance GB/s": algorithmic bytes of the launch groups (the database once per
            # launch) per second of the whole job
            "sc_distance_GBps": value * per_launch / max(1.0, k1_pairs) / 1e9,
            "kernel_ms": {"sc_distance": k1_ms},
            # fastAlignUsingVkey: pairs whose first shift the matrix-core filters left to the exact fp64 evaluation
            "alignment": {"pairs": al_pairs, "exact_fallbacks": al_fallbacks, "fallback_rate": al_fallbacks / max(1, al_pairs)},
            # what the exact fp64 pass scored: the rate above depends on it (worst case -- every keyframe survives -- is
            # secondary.exact_all_pairs' rate; secondary.adversarial_survivors is a database with 5 % of the keyframes inside the margin)
            "survivors_per_scan": {"mean": sv_sum / max(1, sv_q), "max": sv_max, "scans": sv_q,
                                   "screening_margin": "d~ <= min d~ + 2 x the launch's largest per-pair bound (<= 1.5e-3; ~6e-4 on this database)"},
            "semantics": "value = pairs through arg-min detection (winner index / shift / f64 distance bit-identical to the CPU restatement); "
                         "NOT every pair's f64 distance -- that is secondary.exact_all_pairs; a step is 16 scans that arrive together, "
                         "secondary.detect_full_blocking_us is one scan on its own.  STAGE: the timed scans are keyframes already ingested "
                         "(descriptor built, slot written) and resident in HBM: value times the DETECTION stage of the per-incoming-scan path.  The "
                         "front stage (raw points -> descriptor -> slot) is secondary.ingest_per_scan (device time and roofline, K3), the ring-key "
                         "scan on its own secondary.ringkey_topk (K2), and the whole path from host point clouds -- PCIe inclusive, descriptor + "
                         "append + full-database detection per scan -- is secondary.stream_from_points, an order of magnitude below value because "
                         "a scan's 1.9 MB cross PCIe in 34 us; the same path with the raw points resident in HBM (K3 + K2 + K1 per scan, nothing "
                         "but results over PCIe) is secondary.stream_from_resident_points",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": f"screening launch group of {k1_scans:.0f} scans x {n_elig} keyframes: sc_screen2_kernel (products: one keyframe "
                                   f"against the launch's scans per matrix-core tile) + sc_screen2_tail2_kernel (finishing beside the next batch's alignment, "
                                   f"itself one keyframe against the scans per tile); HIP events around the launch groups of every second chunk of the stream (eight groups), "
                                   f"their time divided by the groups' number",
                         "algorithmic_bytes_per_launch": per_launch,
                         "scans_per_launch": k1_scans, "pairs_per_launch": k1_pairs,
                         "pricing": "SURVEY 8(d), Q scans per database pass: DB bytes once per launch + per-scan bytes + per-pair intermediates "
                                    f"({KERNEL_BYTES_PER_KEYFRAME} B per keyframe: chunk-major fp16 image, alignment image, tiled ring key, mask; "
                                    f"{KERNEL_BYTES_PER_PAIR_IO} B per pair: first shift, partial sums, bound, ring-key metric)",
                         "note": "the database crosses HBM once per launch of Q scans; the per-pair operand (the scan rotated by the pair's first "
                                 "shift) comes out of LDS.  `single_scan_hbm_floor`: what any kernel that streams the database once per scan "
                                 "could reach at 100 % of the HBM peak",
                         "single_scan_hbm_floor": {"pairs_per_s": single_floor, "bytes_per_pair": SINGLE_SCAN_BYTES_PER_PAIR,
                                                   "pairs_per_s_survey_bytes": HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_PAIR,
                                                   "value_over_floor": value / world / single_floor},
                         "survey_equivalent": {"bytes_per_pair": ALGO_BYTES_PER_PAIR, "achieved": survey_equiv,
                                               "note": "SURVEY 8(d)'s single-scan price (fp32 descriptor per pair) x pairs / time, for "
                                                       "comparison with round 1's figures; not a rate of moved bytes"}},
            "device": eng.device_name(),
        }
        if world > 1:
            out["collective"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                                 "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0,
                                 "library": "RCCL (torch.distributed 'nccl' on ROCm)" if dist.get_backend() == "nccl" else "gloo (CPU rehearsal: SCL_BENCH_SHARE_GPU=1)",
                                 "exchange": args.exchange, "devices_distinct": not share_gpu}
        if world == 1 and not args.no_secondary:
            out["secondary"] = {}
            for name, fn in (("exact_all_pairs", lambda: secondary_exact_all_pairs(eng, n_elig, n_query)),
                             ("detect_full_blocking_us", lambda: secondary_blocking_scan(eng, n_elig, n_query)),
                             ("sc_distance_80x180", lambda: secondary_80x180(local_rank)),
                             ("adversarial_survivors", lambda: secondary_adversarial_survivors(local_rank, shard, queries, n_elig)),
                             ("icp_verification", lambda: secondary_icp(eng)),
                             ("livox_stream_80x180", lambda: secondary_livox_stream(local_rank)),
                             # (the front's measurements last: their copy streams are two more of the process's streams, and with them
                             #  created first the verification's side stream shared a hardware queue with its main stream -- 4.3 against
                             #  3.8 ms per point-to-plane query)
                             ("ringkey_topk", lambda: secondary_ringkey_topk(eng, n_elig, n_query)),
                             ("ingest_per_scan", lambda: secondary_ingest_per_scan(local_rank)),
                             ("stream_from_points", lambda: secondary_stream_from_points(local_rank)),
                             ("stream_from_resident_points", lambda: secondary_stream_from_resident_points(local_rank))):
                if args.only_secondary and name not in args.only_secondary.split(","):
                    continue
                try:
                    out["secondary"][name] = fn()
                except Exception as ex:                      # never lose the headline line to a secondary measurement
                    out["secondary"][name] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(shard, args.cpu_pairs)
            icp_gpu = out.get("secondary", {}).get("icp_verification")
            if isinstance(icp_gpu, dict) and "error" not in icp_gpu:        # the ICP half of the denominator, beside the GPU figures it belongs to
                try:
                    icp_cpu = cpu_baseline_icp()
                    for name in ("point_to_plane", "point_to_point"):
                        g = icp_gpu[name]["from_store"]["ms_per_query"]
                        icp_cpu[name]["gpu_from_store_speedup_vs_pool"] = icp_cpu[name]["pool_over_problems"]["ms_per_query"] / g
                        icp_cpu[name]["gpu_from_store_speedup_vs_one_thread"] = icp_cpu[name]["one_thread"]["ms_per_query_of_25_scaled"] / g
                    icp_gpu["cpu_baseline"] = icp_cpu
                except Exception as ex:
                    icp_gpu["cpu_baseline"] = {"error": repr(ex)}
        # the figures a reader needs first, once more as short keys at the END of the line (a driver that keeps only the tail of the
        # line keeps these): every value is a copy of a field above
        sec = out.get("secondary", {})

        def pick(d, *path):
            for k in path:
                if not isinstance(d, dict) or k not in d:
                    return None
                d = d[k]
            return d
        out["summary"] = {
            "value_pairs_per_s_batches_of_16": value, "roofline_frac": out["roofline"]["frac"],
            "q1_blocking_scan_us_p50": pick(sec, "detect_full_blocking_us", "p50"),
            "q1_pairs_per_s": pick(sec, "detect_full_blocking_us", "pairs_per_s_at_p50"),
            "detect_intra_us_p50": pick(sec, "detect_full_blocking_us", "detect_intra_us", "p50"),
            "exact_all_pairs_per_s": pick(sec, "exact_all_pairs", "value"),
            "ringkey_topk_k3_device_us": pick(sec, "ringkey_topk", "k3", "device_us"),
            "ingest_device_us_per_scan_64x120_120k": pick(sec, "ingest_per_scan", "64x120_120k_points", "device_us_per_scan_in_a_batch_of_16"),
            "ingest_roofline_frac_64x120_120k": pick(sec, "ingest_per_scan", "64x120_120k_points", "roofline", "frac"),
            "resident_points_pairs_per_s": pick(sec, "stream_from_resident_points", "value"),
            "resident_points_us_per_scan": pick(sec, "stream_from_resident_points", "us_per_scan"),
            "stream_from_points_scans_per_s": pick(sec, "stream_from_points", "scans_per_s"),
            "stream_from_points_pairs_per_s": pick(sec, "stream_from_points", "value"),
            "stream_from_points_over_pcie_floor": pick(sec, "stream_from_points", "pcie", "us_per_scan_over_floor"),
            "sc_distance_80x180_pairs_per_s": pick(sec, "sc_distance_80x180", "value"),
            "exact_all_pairs_80x180_per_s": pick(sec, "sc_distance_80x180", "exact_all_pairs", "value"),
            "icp_p2plane_ms_per_query": pick(sec, "icp_verification", "point_to_plane", "from_store", "ms_per_query"),
            "icp_p2p_ms_per_query": pick(sec, "icp_verification", "point_to_point", "from_store", "ms_per_query"),
            "livox_stream_p50_ms": pick(sec, "livox_stream_80x180", "latency_ms", "p50"),
            "livox_stream_dropped_frames": pick(sec, "livox_stream_80x180", "dropped_frames"),
            "cpu_one_thread_pairs_per_s": pick(out, "cpu_baseline", "value"),
            "cpu_all_core_pairs_per_s": pick(out, "cpu_baseline", "all_cores_reference_shaped", "value"),
            "cpu_all_core_threads": pick(out, "cpu_baseline", "all_cores_reference_shaped", "cores")}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
