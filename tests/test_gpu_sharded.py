"""BASELINE configs[3] on the HIP path: ONE keyframe database sharded over G engine states
(scl_create_sharded, keyframe g on shard g % G) must return what one database returns, bit for bit.

A one-GPU box has one device, so the shards share it (`devices=[0] * G`): the sharding arithmetic, the staging of
query keyframes on the shards that do not own them, the per-shard passes and the reduction of the winners are
exactly what runs on a node with G devices; only the device ordinals differ.  Reference rule under test:
the search range [0, cur - NUM_EXCLUDE_RECENT) of descriptor.h:1627 applies to GLOBAL indices."""
import os
import socket

import numpy as np
import pytest

import oracle_binding as ob
from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors, synth_scan, synth_structured_cloud

pytestmark = pytest.mark.gpu


def _same_bits(a, b):
    return np.float64(a).view(np.uint64) == np.float64(b).view(np.uint64)


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_sharded_engine_equals_single_engine_and_oracle(G):
    R, S, k, n = 20, 60, 3, 640
    descs, truth = synth_descriptors(n, R, S, seed=1004, revisit_frac=0.06, return_truth=True)
    one = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=k, initial_capacity=64)
    sh = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=k, initial_capacity=64, devices=[0] * G, exchange=1)
    assert sh.shard_info() == (G, 1)
    db = ob.OracleDB(ob.make_config(R=R, S=S, k=k))
    # mixed ingest: bulk, wire one by one, raw scans -- the shards must end up with keyframe g on shard g % G
    one.save_bulk(descs[:301]); sh.save_bulk(descs[:301]); db.save_bulk(descs[:301])
    robots = (np.arange(n) % 3).astype(np.int8)
    for i in range(301, n - 2):
        one.save_from_wire(descs[i], int(robots[i]), 7 * i); sh.save_from_wire(descs[i], int(robots[i]), 7 * i)
        db.save_wire(descs[i], int(robots[i]), 7 * i)
    for j in range(2):
        cloud = synth_scan(15000, seed=40 + j)
        a = one.make_and_save(cloud, 1, 9000 + j); b = sh.make_and_save(cloud, 1, 9000 + j); c = db.make_and_save(cloud, 1, 9000 + j)
        assert np.array_equal(a, b) and np.array_equal(a, c)
    assert sh.get_size() == one.get_size() == n
    for key in (0, 1, G, 300, 301, 302, n - 3, n - 2, n - 1):
        assert sh.get_index(key) == one.get_index(key) == db.get_index(key)
        assert np.array_equal(sh.get_descriptor(key), one.get_descriptor(key))
        assert np.array_equal(sh.get_ringkey(key).view(np.uint32), one.get_ringkey(key).view(np.uint32))
        assert np.array_equal(sh.get_sectorkey(key).view(np.uint64), one.get_sectorkey(key).view(np.uint64))
    curs = sorted({c for c, _, _ in truth if c < n - 2} | {n - 1, n - 2, 150, 104, 103})
    hits = 0
    for cur in curs:
        a, b, o = sh.detect_intra(cur), one.detect_intra(cur), db.detect_intra(cur)
        assert a[:2] == b[:2] == o[:2] and _same_bits(a[2], b[2]) and a[2] == o[2]
        hits += a[0] >= 0
    assert hits >= 3
    for cur in curs[-12:]:                                  # the inter path keeps tree-period state: same call sequence
        a, b, o = sh.detect_inter(cur), one.detect_inter(cur), db.detect_inter(cur)
        assert a[0] == b[0] == o[0] and np.float32(a[1]) == np.float32(b[1]) == np.float32(o[1]) and _same_bits(a[2], b[2]) and a[2] == o[2]
    for cur in curs[-10:]:
        a, b, o = sh.detect_full(cur), one.detect_full(cur), db.detect_full(cur)
        assert a[:3] == b[:3] == o[:3] and _same_bits(a[3], b[3]) and a[3] == o[3]
        ia, da = sh.last_topk(k); ib, db_ = one.last_topk(k)
        assert np.array_equal(ia, ib) and np.array_equal(da.view(np.uint32), db_.view(np.uint32))
    # building blocks: global top-k with distances, explicit candidate lists (ragged over the shards), staged query
    for q in (n - 1, 417):
        ra = sh.topk_with_distance(q, 5, n - 120, 3); rb = one.topk_with_distance(q, 5, n - 120, 3)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1].view(np.uint32), rb[1].view(np.uint32))
        assert np.array_equal(ra[2].view(np.uint64), rb[2].view(np.uint64)) and np.array_equal(ra[3], rb[3]) and ra[4] == rb[4]
        cand = np.array([0, 5, 2, 2, n - 1, 333, -1, 17, 16 * G % n], dtype=np.int32)
        da, sa = sh.sc_distance_batch(q, cand=cand); dbb, sb = one.sc_distance_batch(q, cand=cand)
        assert np.array_equal(da.view(np.uint64), dbb.view(np.uint64)) and np.array_equal(sa, sb)
        da, sa = sh.sc_distance_batch(q, n=200); dbb, sb = one.sc_distance_batch(q, n=200)
        assert np.array_equal(da.view(np.uint64), dbb.view(np.uint64)) and np.array_equal(sa, sb)
    ext = synth_descriptors(1, R, S, seed=999)[0]
    sh.stage_query(ext); one.stage_query(ext)
    a = sh.detect_full_range(-1, 3, n - 7); b = one.detect_full_range(-1, 3, n - 7)
    assert a[:2] == b[:2] and _same_bits(a[2], b[2])
    # streams: several scans per launch, several launches in flight, ragged ranges, an empty range
    qs = np.array([n - 1 - i for i in range(21)], dtype=np.int32)
    lo = np.array([0, 3, 0, 50, 1, 0, 2] * 3, dtype=np.int32)
    hi = np.array([q - 100 for q in qs], dtype=np.int32); hi[5] = 0
    for spl, depth in ((1, 1), (2, 2), (4, 2), (3, 1)):
        a = sh.detect_full_stream(qs, lo, hi, spl, depth); b = one.detect_full_stream(qs, lo, hi, spl, depth)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint64), b[2].view(np.uint64))
    # a backlog in ascending order across three blocks of the front's stream form (runs of keyframes that follow each other on their
    # owner are staged with one copy per array), the staged external query in the middle of it, a keyframe asked for twice
    qb = np.concatenate([np.arange(n - 150, n - 80), [-1], np.arange(n - 80, n), [n - 3, n - 3]]).astype(np.int32)
    hb = np.where(qb >= 0, qb - 100, n - 7).astype(np.int32)
    a = sh.detect_full_stream(qb, 0, hb, 16, 2); b = one.detect_full_stream(qb, 0, hb, 16, 2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint64), b[2].view(np.uint64))
    tk = sh.detect_full_submit_many(qs[:6], lo[:6], hi[:6]); tk1 = one.detect_full_submit_many(qs[:6], lo[:6], hi[:6])
    for t, t1 in reversed(list(zip(tk, tk1))):              # collected in any order
        a = sh.detect_full_collect(t); b = one.detect_full_collect(t1)
        assert a[:2] == b[:2] and _same_bits(a[2], b[2])
    sh.close(); one.close()


@pytest.mark.parametrize("G", [2, 3])
def test_mirror_rows_of_recent_keyframes(G):
    """The other shards' copies of the newest keyframes (made when a keyframe is appended: sharded_front.hip, mirror rows) serve
    scattered queries without a copy at query time; keyframes that have left the mirror (more than 1024 newer ones on their owner)
    go through the staging rows.  Ingest one by one and in bulk, across the ring's wrap and across a growth of the arrays; every
    form of the pass equals one database bit for bit."""
    R, S = 64, 120
    n = 1024 * G + 700                                         # past the wrap of every owner's ring
    descs = synth_descriptors(n, R, S, seed=1013, revisit_frac=0.04)
    one = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=256)
    sh = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=256, devices=[0] * G, exchange=1)
    rs = np.random.RandomState(5)
    done = 0

    def check(count):
        # scattered queries: newest first, random recent ones, some old ones (staging rows), repeated keyframes
        qs = np.concatenate([np.arange(done - 1, max(done - 40, -1), -1), rs.randint(max(0, done - 900 * G), done, 70),
                             rs.randint(0, done, 30), [done - 1, done - 1]]).astype(np.int32)[:count]
        his = np.maximum(qs - 50, 0).astype(np.int32)
        a = one.detect_full_stream(qs, 0, his, 16, 2)
        b = sh.detect_full_stream(qs, 0, his, 16, 2)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint64), b[2].view(np.uint64))
        for q in (int(qs[0]), int(qs[45 % len(qs)])):           # blocking passes and the top-k take the same rows
            assert _same_full(one.detect_full_range(q, 0, max(q - 50, 0)), sh.detect_full_range(q, 0, max(q - 50, 0)))
            ia, da = one.ringkey_topk(q, 0, max(q - 50, 0), 5)[:2]
            ib, db_ = sh.ringkey_topk(q, 0, max(q - 50, 0), 5)[:2]
            assert np.array_equal(ia, ib) and np.array_equal(np.asarray(da).view(np.uint32), np.asarray(db_).view(np.uint32))

    for step in (300, 1, 1, 1, 500, 1, 1300, 1, 1, n):         # bulk appends of several sizes (one larger than a ring) and single ones
        m = min(step, n - done)
        if m <= 0:
            break
        if m == 1:
            one.save_from_wire(descs[done], 0, done); sh.save_from_wire(descs[done], 0, done)
        else:
            one.save_bulk(descs[done:done + m]); sh.save_bulk(descs[done:done + m])
        done += m
        check(142)
    assert done == n
    one.close(); sh.close()


def _same_full(a, b):
    return a[0] == b[0] and a[1] == b[1] and _same_bits(a[2], b[2])


def test_sharded_engine_grows_under_passes_in_flight():
    """appends that move a shard's arrays while another shard's pass still reads a staged copy of a keyframe"""
    R, S = 20, 60
    descs = synth_descriptors(900, R, S, seed=31, revisit_frac=0.05)
    one = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=16)
    sh = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=16, devices=[0, 0, 0], exchange=1)
    one.save_bulk(descs[:300]); sh.save_bulk(descs[:300])
    for i in range(300, 900, 50):
        t_sh = sh.detect_full_submit_many([i - 1, i - 2], [0, 0], [i - 101, i - 102])
        t_one = one.detect_full_submit_many([i - 1, i - 2], [0, 0], [i - 101, i - 102])
        sh.save_bulk(descs[i:i + 50]); one.save_bulk(descs[i:i + 50])            # regrows several times over the loop
        for a, b in zip(t_sh, t_one):
            ra, rb = sh.detect_full_collect(a), one.detect_full_collect(b)
            assert ra[:2] == rb[:2] and _same_bits(ra[2], rb[2])
    assert sh.get_size() == one.get_size() == 900
    for cur in (899, 700, 450):
        assert sh.detect_intra(cur) == one.detect_intra(cur)
    sh.close(); one.close()


def test_rccl_exchange_single_shard():
    """exchange = 2 drives the device-side reduction (pack kernel -> ncclAllReduce(uint64, min) x 2 -> select kernel)
    through RCCL; with one device it is the only configuration a one-GPU box can run, and it must agree with the
    host merge and with an unsharded engine"""
    R, S, n = 64, 120, 1500
    descs = synth_descriptors(n, R, S, seed=77, revisit_frac=0.05)
    one = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    rc = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n, devices=[0], exchange=2)
    assert rc.shard_info() == (1, 2)
    one.save_bulk(descs); rc.save_bulk(descs)
    qs = np.arange(n - 1, n - 41, -1, dtype=np.int32)
    hi = (qs - 100).astype(np.int32); hi[3] = 0
    a = rc.detect_full_stream(qs, 0, hi, 4, 2); b = one.detect_full_stream(qs, 0, hi, 4, 2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint64), b[2].view(np.uint64))
    for cur in (n - 1, n - 7):
        x, y = rc.detect_full(cur), one.detect_full(cur)
        assert x[:3] == y[:3] and _same_bits(x[3], y[3])
    with pytest.raises(Exception):
        ScanContextEngine(num_ring=R, num_sector=S, devices=[0, 0], exchange=2)     # RCCL needs one device per shard
    rc.close(); one.close()


@pytest.mark.parametrize("G", [2, 8])
def test_device_side_exchange_control_flow_with_several_ranks(G):
    """exchange = 2's control flow -- pack kernel per rank, grouped all-reduce(uint64, min) on the distance keys,
    select kernel, second all-reduce on (global index, shift), one D2H -- executed with G > 1 ranks.  RCCL refuses two ranks
    on one device, so the collective library is the tests' stand-in (tests/cpp/libmock_rccl.so, loaded through SCL_RCCL_LIB:
    element-wise min of the ranks' buffers on the host); everything around it is the product's code, unchanged.  The library is
    chosen once per process, so the body runs in a child interpreter (`_exchange_control_flow_body`)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    mock = os.path.join(here, "cpp", "libmock_rccl.so")
    assert os.path.exists(mock), "make builds tests/cpp/libmock_rccl.so"
    code = f"import sys; sys.path.insert(0, {here!r}); sys.path.insert(0, {os.path.dirname(here)!r}); import test_gpu_sharded as t; t._exchange_control_flow_body({G})"
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SCL_RCCL_LIB=mock), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]


def _exchange_control_flow_body(G):
    """Must equal the host merge (exchange = 1) and one unsharded database, bit for bit -- duplicated keyframes on different
    shards (equal distances: the lowest GLOBAL index wins, which is what the second reduction decides), ranges that leave some
    shards empty, a range with no candidate at all."""
    R, S, n = 64, 120, 1800
    descs = synth_descriptors(n, R, S, seed=78, revisit_frac=0.05)
    rs = np.random.RandomState(4)
    for i in range(40):                                                   # exact duplicates at other global indices: ties across shards
        a, b = rs.randint(0, n - 300, 2)
        descs[b] = descs[a]
    for q in range(n - 40, n, 5):                                         # queries that are rolled copies of a duplicated pair
        a = int(rs.randint(0, n - 300)); descs[(a + 1 + G // 2) % (n - 300)] = descs[a]
        descs[q] = np.roll(descs[a], int(rs.randint(0, S)), axis=1)
    one = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    host = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n, devices=[0] * G, exchange=1)
    dev = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n, devices=[0] * G, exchange=2)
    assert dev.shard_info() == (G, 2) and host.shard_info() == (G, 1)
    for e in (one, host, dev):
        e.save_bulk(descs)
    for cur in list(range(n - 1, n - 41, -1)):
        x, y, z = dev.detect_full(cur), host.detect_full(cur), one.detect_full(cur)
        assert x[:3] == y[:3] == z[:3] and _same_bits(x[3], y[3]) and _same_bits(x[3], z[3]), cur
    # launch groups of several scans (one exchange per group), ragged and empty ranges
    qs = [n - 1, n - 6, n - 11, n - 3]
    for his in ([n - 101, n - 106, n - 111, n - 103], [G - 1, 1, 0, 3], [0, 0, 0, 0], [n - 101, 0, 2, n - 103]):
        td = dev.detect_full_submit_many(qs, [0] * 4, his); th = host.detect_full_submit_many(qs, [0] * 4, his)
        to = one.detect_full_submit_many(qs, [0] * 4, his)
        for a, b, c in zip(td, th, to):
            ra, rb, rc_ = dev.detect_full_collect(a), host.detect_full_collect(b), one.detect_full_collect(c)
            assert ra[:2] == rb[:2] == rc_[:2] and _same_bits(ra[2], rb[2]) and _same_bits(ra[2], rc_[2]), (his, ra, rb, rc_)
    # several groups in flight before the first collect
    tick = [(dev.detect_full_submit_many([n - 1 - j, n - 2 - j], [0, 0], [n - 101 - j, n - 102 - j]),
             one.detect_full_submit_many([n - 1 - j, n - 2 - j], [0, 0], [n - 101 - j, n - 102 - j])) for j in range(0, 6, 2)]
    for td, to in tick:
        for a, c in zip(td, to):
            ra, rc_ = dev.detect_full_collect(a), one.detect_full_collect(c)
            assert ra[:2] == rc_[:2] and _same_bits(ra[2], rc_[2])
    dev.close(); host.close(); one.close()


def test_candidates_of_one_scan_verified_across_shards():
    """scl_icp_align_batch on a sharded engine deals the loop candidates to the shards"""
    tgts = [synth_structured_cloud(3000 + 100 * c, seed=60 + c) for c in range(5)]
    src = tgts[3][::2].copy(); src[:, 0] += 0.03
    one = ScanContextEngine()
    sh = ScanContextEngine(devices=[0, 0], exchange=1)
    a = sh.icp_align_batch(src, tgts); b = one.icp_align_batch(src, tgts)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert b[2][3]
    sh.close(); one.close()


@pytest.mark.parametrize("G", [2, 8])
def test_configs3_100k_keyframes_sharded_equals_single_database(G):
    """BASELINE configs[3] at full size: 100 000 keyframes, 64x120, over G shards vs one 100k-keyframe engine
    (3 GB + 3 GB of HBM): the stream of full-DB passes, the reference-faithful detection and the global top-k."""
    R, S, n, per = 64, 120, 100000, 12500
    one = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n + 64)
    sh = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n + 64, devices=[0] * G, exchange=1)
    for c in range(n // per):                                 # eight trajectory segments, as bench.py builds its shards
        seg = synth_descriptors(per, R, S, seed=1004 + 7919 * c, revisit_frac=0.0)
        if c == n // per - 1:
            rs = np.random.RandomState(9)
            for i in range(per - 100, per, 3):                # revisits of keyframes anywhere in the database
                src = int(rs.randint(0, per - 200)); seg[i] = np.roll(seg[src], int(rs.randint(0, S)), axis=1)
        one.save_bulk(seg); sh.save_bulk(seg)
    assert sh.get_size() == one.get_size() == n
    qs = np.arange(n - 1, n - 33, -1, dtype=np.int32)
    a = sh.detect_full_stream(qs, 0, qs - 100, 4, 2); b = one.detect_full_stream(qs, 0, qs - 100, 4, 2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint64), b[2].view(np.uint64))
    assert (a[2] < 1e-6).sum() >= 8                           # the planted revisits are found
    for cur in (n - 1, n - 2, n - 50):
        assert sh.detect_intra(cur) == one.detect_intra(cur)
        x, y = sh.detect_full(cur), one.detect_full(cur)
        assert x[:3] == y[:3] and _same_bits(x[3], y[3])
    if G == 2:
        # The checker at full size: two scans -- a planted revisit and an ordinary keyframe -- scored by
        # the CPU restatement against ALL eligible keyframes of the 100k database (every core of the host), arg-min as the
        # reference's loop takes it (strict <, ascending index), winner compared bit for bit with both engines.
        allv = one.get_descriptors(0, n)
        cfg = ob.make_config(R=R, S=S)
        odb = ob.OracleDB(cfg); odb.save_bulk(allv)
        del allv
        threads = len(os.sched_getaffinity(0))
        for cur in (n - 1, n - 2):
            hist = cur - 100
            dist, shift = odb.distance_batch_mt(cur, np.arange(hist, dtype=np.int32), True, threads)
            best = int(np.flatnonzero(dist == np.nanmin(dist))[0])
            x = sh.detect_full(cur)
            assert x[1] == best and x[2] == int(shift[best]) and _same_bits(x[3], dist[best]), (cur, x, best, dist[best])
            assert x[:3] == one.detect_full(cur)[:3]
        odb.close()
    sh.close(); one.close()


# ---- the multi-process path (scl_slam_amd/sharded.py) with the HIP engine as the shard scorer ------------------------

def _rank_main(rank, world, port, out_q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scl_slam_amd.sharded import FullScanStream, ShardedLoopDetector, local_count
    R, S, n = 64, 120, 4000
    descs = synth_descriptors(n, R, S, seed=1004, revisit_frac=0.03)
    eng = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)      # both ranks on the one device of this box
    det = ShardedLoopDetector(eng, rank, world, num_candidates=3)
    for i in range(n):
        det.save(descs[i], 0, i)
    curs = list(range(n - 1, n - 13, -1))
    res = [(det.detect_intra(c, descs[c]), det.detect_full(c, descs[c])) for c in curs]
    st = FullScanStream(eng, rank, world, depth=2, merge_every=4, scans_per_launch=1)
    for c in curs:
        eng.stage_query(descs[c])
        st.submit(-1, 0, local_count(c - 100, rank, world))
    staged = st.drain()
    # bench.py's layout for N > 1: every rank holds its shard of the database followed by ALL query keyframes, and hands the
    # scans to the native pipeline as arrays (FullScanStream.submit_many), winners merged by the packed-key all-reduces
    n_db, n_q = 3000, 12
    eng2 = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n_db // world + n_q + 8)
    eng2.save_bulk(descs[rank:n_db:world])
    n_local = eng2.get_size()
    eng2.save_bulk(descs[n - n_q:])
    st2 = FullScanStream(eng2, rank, world, depth=2, scans_per_launch=4, native_chunk=5)
    st2.submit_many(n_local + np.arange(n_q, dtype=np.int32), 0, n_local)
    arrays = st2.drain()
    out_q.put((rank, res, staged, arrays))
    dist.barrier(); dist.destroy_process_group()
    eng.close(); eng2.close()


def test_two_processes_share_the_database_over_gloo():
    """world_size 2, one process per shard (both on cuda:0 here), min all-reduces on packed keys over gloo: every rank
    must reach the single-database verdict for detect_intra / detect_full and for the batched stream"""
    import torch.multiprocessing as mp
    R, S, n = 64, 120, 4000
    descs = synth_descriptors(n, R, S, seed=1004, revisit_frac=0.03)
    one = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n)
    one.save_bulk(descs)
    curs = list(range(n - 1, n - 13, -1))
    want = [(one.detect_intra(c), one.detect_full(c)) for c in curs]
    one.close()
    n_db, n_q = 3000, 12
    one = ScanContextEngine(num_ring=R, num_sector=S, initial_capacity=n_db + n_q)
    one.save_bulk(descs[:n_db]); one.save_bulk(descs[n - n_q:])
    want_arrays = [one.detect_full_range(n_db + i, 0, n_db) for i in range(n_q)]
    one.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=300) for _ in range(2)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    for _, res, stream, arrays in got:
        for (intra, full), (w_intra, w_full), (d, g, shf) in zip(res, want, stream):
            assert intra == w_intra and full == w_full
            assert (g, shf) == (w_full[1], w_full[2]) and _same_bits(d, w_full[3])
        assert len(arrays) == len(want_arrays)
        for (d, g, shf), w in zip(arrays, want_arrays):
            assert (g, shf) == (w[0], w[1]) and _same_bits(d, w[2])


@pytest.mark.parametrize("G", [2, 3])
def test_points_entry_points_on_a_sharded_engine_equal_the_single_engine(G):
    """scl_make_and_save_many / scl_stream_from_points on a sharded front (keyframe g on shard g % G; the front goes keyframe by
    keyframe through the owners and searches group by group through its stream form): descriptors, winners, shifts and fp64 distances
    equal the one-GPU engine's, bit for bit."""
    from scl_slam_amd.synth import synth_descriptors, synth_scan
    R, S, n0, excl = 64, 120, 180, 20
    base = synth_descriptors(n0, R, S, seed=31)
    one = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=excl)
    sh = ScanContextEngine(num_ring=R, num_sector=S, num_exclude_recent=excl, initial_capacity=64, devices=[0] * G, exchange=1)
    one.save_bulk(base); sh.save_bulk(base)
    clouds = [synth_scan(4000 + 300 * (i % 4), seed=600 + i, stride_floats=4) for i in range(21)]
    for i in range(excl + 1, len(clouds), 3):
        clouds[i] = clouds[i - excl - 1]
    v1 = one.make_and_save_many(clouds[:5], robots=[1] * 5, indexs=list(range(5)))
    v2 = sh.make_and_save_many(clouds[:5], robots=[1] * 5, indexs=list(range(5)))
    assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32)) and sh.get_size() == n0 + 5 and sh.get_index(n0 + 3) == (1, 3)
    a = one.stream_from_points(clouds[5:], want_values=True)
    b = sh.stream_from_points(clouds[5:], want_values=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint64), b[2].view(np.uint64))
    assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32))
    assert (a[0] >= 0).sum() >= 1 and sh.get_size() == one.get_size() == n0 + len(clouds)
    one.close(); sh.close()
