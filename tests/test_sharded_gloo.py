"""Multi-rank path on CPU: world_size 2 over gloo.  The local scorer is the CPU checker wrapped
in the engine's interface (tests may use oracle/); what is under test is the sharding arithmetic,
the exchange and the merge -- they must reproduce the single-database result exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
from scl_slam_amd.sharded import ShardedLoopDetector, local_count
from scl_slam_amd.synth import synth_descriptors

R, S, K, N = 20, 60, 3, 420


class OracleShardEngine:
    """Engine-shaped wrapper over the CPU checker (one shard)."""

    def __init__(self, cfg):
        self.cfg, self.db, self.q = cfg, ob.OracleDB(cfg), None
        self.R, self.S = cfg.num_ring, cfg.num_sector

    def save_from_wire(self, values, robot=0, index=0):
        self.db.save_wire(values, robot, index)

    def stage_query(self, values):
        self.q = np.ascontiguousarray(values, dtype=np.float32).reshape(self.R, self.S)
        tmp = ob.OracleDB(self.cfg); tmp.save_wire(self.q)
        self.qkey = tmp.ringkey(0)

    def _dist(self, slot):
        return ob.distance(self.cfg, self.q, self._desc(slot), fast=True)

    def _desc(self, slot):
        cm = np.ctypeslib.as_array(self.db.L.sco_db_desc(self.db.h, slot), shape=(self.S, self.R))
        return cm.T.astype(np.float32)

    def topk_with_distance(self, query, lo, hi, k):
        assert query == -1
        idx, d2, found = ob.knn(self.db.ringkeys(hi)[lo:], self.qkey, k) if hi > lo else (np.full(k, -1, np.int32), np.zeros(k, np.float32), 0)
        dist_ = np.full(k, 1e7); shift = np.zeros(k, np.int32)
        for i in range(k):
            if idx[i] >= 0:
                idx[i] += lo
                dist_[i], shift[i] = self._dist(int(idx[i]))
        return idx, d2, dist_, shift, found

    def detect_full_submit(self, query, lo, hi):
        return self.detect_full_range(query, lo, hi)                # the checker has no queue: done at submit

    def detect_full_collect(self, ticket):
        return ticket

    def detect_full_range(self, query, lo, hi):
        best, bi, bs = 1e7, -1, 0
        for s in range(lo, hi):
            d, sh = self._dist(s)
            if d < best:
                best, bi, bs = d, s, sh
        return bi, bs, best


def test_full_scan_stream_orders_and_batches_single_process():
    """FullScanStream on one rank: every combination of launches in flight, scans per launch and merge batch
    returns the scans' results in submission order, equal to blocking calls (the checker stands in for the engine)."""
    from scl_slam_amd.sharded import FullScanStream
    n = 160
    descs = synth_descriptors(n, R, S, seed=77, revisit_frac=0.1)
    cfg = ob.make_config(R=R, S=S, k=K)
    eng = OracleShardEngine(cfg)
    for i in range(n):
        eng.save_from_wire(descs[i], 0, i)

    class SlotEngine:
        """adds the engine's submit_many and its limit of eight result slots"""
        def __init__(self, inner):
            self.inner, self.busy, self.q = inner, 0, None
        def stage(self, q):
            self.inner.stage_query(descs[q])
        def detect_full_submit(self, query, lo, hi):
            assert self.busy < 8; self.busy += 1
            self.stage(query)
            return self.inner.detect_full_range(-1, lo, hi)
        def detect_full_submit_many(self, qs, los, his):
            return [self.detect_full_submit(q, lo, hi) for q, lo, hi in zip(qs, los, his)]
        def detect_full_collect(self, t):
            self.busy -= 1
            return t

    queries = list(range(n - 1, n - 24, -1))
    blocking = []
    for q in queries:
        eng.stage_query(descs[q]); blocking.append(eng.detect_full_range(-1, 0, q - 30))
    for depth in (1, 2, 3):
        for per_launch in (1, 2, 3, 4):
            for merge_every in (1, 5, 16):
                st = FullScanStream(SlotEngine(eng), depth=depth, merge_every=merge_every, scans_per_launch=per_launch)
                for q in queries:
                    st.submit(q, 0, q - 30)
                res = st.drain()
                assert [(g, sh, d) for d, g, sh in res] == [(nn, sh, d) for nn, sh, d in blocking], (depth, per_launch, merge_every)


class OracleIcpEngine:
    """engine-shaped wrapper over the CPU ICP checker (tests may use oracle/)"""
    def icp_align_batch(self, src, tgts, params=None):
        import oracle_icp_binding as oi
        res = [oi.icp_align(src, t) for t in tgts]
        return (np.stack([r[0] for r in res]), np.array([r[1] for r in res], np.float32),
                np.array([r[2] for r in res]), np.array([r[3] for r in res], np.int32))


def _icp_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scl_slam_amd.sharded import verify_candidates_sharded
    from scl_slam_amd.synth import synth_structured_cloud
    tgts = [synth_structured_cloud(600 + 40 * c, seed=50 + c) for c in range(5)]
    src = tgts[2][::2].copy(); src[:, 0] += 0.02
    T, fit, conv, it = verify_candidates_sharded(OracleIcpEngine(), src, tgts, rank, world)
    out_q.put((rank, T.tobytes(), fit.tobytes(), conv.tolist(), it.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_candidate_verification_sharded_over_ranks():
    """5 candidates over 2 ranks: every rank ends with the table a single rank computes"""
    from scl_slam_amd.sharded import verify_candidates_sharded
    from scl_slam_amd.synth import synth_structured_cloud
    tgts = [synth_structured_cloud(600 + 40 * c, seed=50 + c) for c in range(5)]
    src = tgts[2][::2].copy(); src[:, 0] += 0.02
    T1, f1, c1, i1 = verify_candidates_sharded(OracleIcpEngine(), src, tgts)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_icp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=180) for _ in range(2)]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    for _, Tb, fb, conv, it in got:
        assert Tb == T1.tobytes() and fb == f1.tobytes() and conv == c1.tolist() and it == i1.tolist()
    assert c1[2] and i1[2] >= 1


def test_packed_winner_keys_order_and_ties():
    """the two-phase min all-reduce on packed keys == lexicographic (distance, global index) arg-min, including
    negative distances (1 - mean cosine can round below zero), exact ties and ranks without a candidate"""
    from scl_slam_amd.sharded import BIG_DIST, I64_MAX, dist_to_key, key_to_dist, pack_winner_keys, unpack_winner_keys
    d = np.array([-3e-16, -1e-300, 0.0, 5e-324, 1e-9, 0.1399999999999, 0.14, 1.0, 2.0, 1e7])
    k = dist_to_key(d)
    assert np.all(np.diff(k) > 0) and np.array_equal(key_to_dist(k).view(np.uint64), d.view(np.uint64))
    rs = np.random.RandomState(3)
    for world in (2, 3, 8):
        m = 64
        rec = np.zeros((world, m, 3))
        rec[:, :, 0] = rs.choice([0.0, 0.05, 0.05 + 1e-17, 0.3, -2e-16, 1.0 - 2 ** -53], size=(world, m))
        rec[:, :, 1] = rs.randint(0, 100000, size=(world, m)) * world + np.arange(world)[:, None]
        rec[:, :, 2] = rs.randint(0, 120, size=(world, m))
        rec[rs.random_sample((world, m)) < 0.2, 1] = -1.0
        rec[:, 0, 1] = -1.0                                              # a scan no rank has a candidate for
        keys = [pack_winner_keys(rec[r]) for r in range(world)]
        m1 = np.min(np.stack([k1 for k1, _ in keys]), axis=0)
        m2 = np.min(np.stack([np.where(k1 == m1, k2, I64_MAX) for k1, k2 in keys]), axis=0)
        got = unpack_winner_keys(m1, m2)
        for j in range(m):
            r = rec[:, j, :]; r = r[r[:, 1] >= 0]
            if len(r) == 0:
                assert got[j] == (BIG_DIST, -1, 0)
                continue
            b = r[np.lexsort((r[:, 1], r[:, 0]))[0]]
            assert got[j][1:] == (int(b[1]), int(b[2])) and np.float64(got[j][0]).view(np.uint64) == np.float64(b[0]).view(np.uint64)


def test_local_count():
    for world in (1, 2, 3, 8):
        for hi in range(0, 40):
            for r in range(world):
                assert local_count(hi, r, world) == len([g for g in range(hi) if g % world == r])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, curs, exchange, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    descs = synth_descriptors(N, R, S, seed=1001, revisit_frac=0.06)
    cfg = ob.make_config(R=R, S=S, k=K)
    det = ShardedLoopDetector(OracleShardEngine(cfg), rank, world, num_candidates=K, exchange=exchange)
    for i in range(N):
        det.save(descs[i], 0, i)
    res = []
    for cur in curs:
        res.append((cur, det.detect_intra(cur, descs[cur]), det.detect_full(cur, descs[cur])))
    # the same scans as a stream: winners of 4 scans per (asynchronous) all-gather, merged a batch later
    from scl_slam_amd.sharded import FullScanStream
    st = FullScanStream(det.engine, rank, world, depth=2, merge_every=4, exchange=exchange)
    for cur in curs:
        det.engine.stage_query(descs[cur])
        st.submit(-1, 0, local_count(cur - det.exclude, rank, world))
    stream = st.drain()
    out_q.put((rank, res, stream))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange", [(2, "allreduce"), (3, "allreduce"), (2, "allgather")])
def test_sharded_equals_single_database(world, exchange):
    descs, truth = synth_descriptors(N, R, S, seed=1001, revisit_frac=0.06, return_truth=True)
    curs = sorted({c for c, _, _ in truth} | {N - 1, 103, 104, 150})
    ref = ob.OracleDB(ob.make_config(R=R, S=S, k=K)); ref.save_bulk(descs)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, curs, exchange, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = [q.get(timeout=120) for _ in range(world)]
        results = {r: a for r, a, _ in got}
        streams = {r: b for r, _, b in got}
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    assert results[0] == results[1]                                  # every rank reaches the same verdict
    assert streams[0] == streams[1] and len(streams[0]) == len(curs)
    for (cur, _, full), (d, g, sh) in zip(results[0], streams[0]):   # batched exchange == per-scan exchange
        assert (g, sh, d) == (full[1], full[2], full[3])
    hits = 0
    for cur, intra, full in results[0]:
        o = ref.detect_intra(cur)
        assert (intra[0], intra[1]) == (o[0], o[1]) and intra[2] == o[2]
        of = ref.detect_full(cur)
        assert full == (of[0], of[1], of[2], of[3])
        hits += intra[0] >= 0
    assert hits >= 3
