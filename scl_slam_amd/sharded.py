"""Keyframe database sharded by keyframe index across the GPUs of one node.

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Global keyframe ``g`` lives on rank ``g % G`` in local slot
``g // G`` (round robin keeps the shards balanced while the database grows append-only).
Every (query, keyframe) pair is independent, so scoring needs no data-path collective; the
only exchange is the reduction of the per-rank winners:

* reference-faithful mode (``detect_intra``, descriptor.h:1613-1674): every rank returns its
  local ring-key top-k *with* their SC distance/shift, one all-gather of k records per rank,
  then every rank merges to the global top-k (ascending ring distance, ties -> lowest global
  index) and applies the reference's candidate loop including its float narrowing
  (descriptor.h:1645-1659);
* full-DB mode (``detect_full``, ``FullScanStream``): the min all-reduce north_star names, on
  packed 64-bit keys.  fp64 distance + index + shift do not fit one 64-bit key without rounding
  the distance (which would change near-tie winners), so the exact reduction takes two 8-byte
  ``all_reduce(MIN)`` per scan: (1) the order-preserving integer image of the fp64 distance,
  (2) ``global_index << 16 | shift`` contributed only by ranks whose local minimum equals the
  global one (everyone else sends INT64_MAX) -> ties go to the lowest global index, exactly the
  single-database rule.  ``exchange="allgather"`` keeps the round-1 all-gather + host merge as a
  fallback to measure the collective's cost against.

Messages are <= k * 32 bytes per rank: latency bound, xGMI bandwidth is irrelevant here.
For a stream of scans (``FullScanStream``) the winners of ``merge_every`` scans travel in ONE
asynchronous all-gather that is awaited a batch later, so the exchange (and the CU the RCCL
kernel needs while the SC-distance kernel owns every CU) never sits between two scans.
The search-range rule ``[0, cur - NUM_EXCLUDE_RECENT)`` (descriptor.h:1627) is applied on
GLOBAL indices: rank r may use local slots ``l`` with ``l * G + r < cur - exclude``.
"""
import numpy as np

BIG_DIST = 10000000.0
I64_MAX = np.iinfo(np.int64).max


def dist_to_key(d):
    """fp64 distances -> int64 keys with the same order (IEEE bits; negative values have their
    magnitude bits flipped).  Self-inverse up to the view: ``key_to_dist(dist_to_key(d)) == d``."""
    b = np.ascontiguousarray(d, dtype=np.float64).view(np.int64)
    return np.where(b >= 0, b, b ^ I64_MAX)


def key_to_dist(k):
    k = np.ascontiguousarray(k, dtype=np.int64)
    return np.where(k >= 0, k, k ^ I64_MAX).view(np.float64)


def pack_winner_keys(rec):
    """rec: (m, 3) float64 rows (dist, global index or -1, shift) -> (key1, key2) int64 arrays.
    key1 orders by distance (INT64_MAX = this rank has no candidate), key2 = index << 16 | shift."""
    rec = np.asarray(rec, dtype=np.float64).reshape(-1, 3)
    have = rec[:, 1] >= 0
    k1 = np.where(have, dist_to_key(rec[:, 0]), I64_MAX)
    k2 = np.where(have, (rec[:, 1].astype(np.int64) << 16) | (rec[:, 2].astype(np.int64) & 0xFFFF), I64_MAX)
    return k1, k2


def unpack_winner_keys(m1, m2):
    """reduced keys -> list of (dist, global index, shift); no candidate anywhere -> (BIG_DIST, -1, 0)"""
    m1 = np.asarray(m1, dtype=np.int64); m2 = np.asarray(m2, dtype=np.int64)
    none = (m1 == I64_MAX) | (m2 == I64_MAX)
    d = np.where(none, BIG_DIST, key_to_dist(np.where(m1 == I64_MAX, 0, m1)))
    g = np.where(none, -1, m2 >> 16)
    sh = np.where(none, 0, m2 & 0xFFFF)
    return list(zip(d.tolist(), g.tolist(), sh.tolist()))


def local_count(global_hi, rank, world):
    """number of local slots l with l*world + rank < global_hi"""
    if global_hi <= rank:
        return 0
    return (global_hi - rank + world - 1) // world


class ShardedLoopDetector:
    def __init__(self, engine, rank=0, world=1, group=None, device="cpu", num_candidates=3,
                 num_exclude_recent=100, dist_thres=0.14, exchange="allreduce"):
        self.engine, self.rank, self.world, self.group = engine, rank, world, group
        self.device = device
        self.exchange = exchange
        self.k, self.exclude, self.thres = num_candidates, num_exclude_recent, dist_thres
        self.n_global = 0
        self.index_map = []                  # replicated (robot, index) map, descriptor.h:1758-1761

    # ---- ingest: every rank sees every descriptor (the ROS topic is a broadcast), keeps its share
    def save(self, values, robot=0, index=0):
        g = self.n_global
        if g % self.world == self.rank:
            self.engine.save_from_wire(values, robot, index)
        self.index_map.append((robot, index))
        self.n_global += 1
        return g

    def get_index(self, g):
        return self.index_map[g]

    def get_size(self):
        return self.n_global

    # ---- collectives -----------------------------------------------------------------
    def _all_gather(self, rec):
        """rec: float64 array (m, c) -> (world*m, c), same on every rank"""
        if self.world == 1:
            return rec
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(rec)).to(self.device)
        out = torch.empty((self.world * t.shape[0], t.shape[1]), dtype=t.dtype, device=self.device)
        dist.all_gather_into_tensor(out, t, group=self.group)      # concatenated along dim 0, rank order
        return out.cpu().numpy()

    def _all_reduce_min(self, keys):
        """int64 array -> element-wise minimum over the ranks (RCCL / gloo ``all_reduce(MIN)``)"""
        if self.world == 1:
            return keys
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.array(keys, dtype=np.int64, copy=True)).to(self.device)   # copy: the reduction is in place
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return t.cpu().numpy()

    # ---- detection ---------------------------------------------------------------------
    def detect_intra(self, cur, values_cur):
        """detectIntraLoopClosureID on the sharded database; `values_cur` = descriptor of keyframe
        `cur` (every rank has it: it arrived on the broadcast).  Returns (loop_id, shift, dist)."""
        k = self.k
        if cur < self.exclude + k + 1:                                   # descriptor.h:1620-1623
            return -1, 0.0, BIG_DIST
        hi = local_count(cur - self.exclude, self.rank, self.world)
        self.engine.stage_query(values_cur)
        idx, d2, dist_, shift, found = self.engine.topk_with_distance(-1, 0, hi, k)
        rec = np.full((k, 4), -1.0, dtype=np.float64)
        for i in range(k):
            if idx[i] >= 0:
                rec[i] = (float(d2[i]), float(int(idx[i]) * self.world + self.rank), dist_[i], float(shift[i]))
        allr = self._all_gather(rec)
        allr = allr[allr[:, 1] >= 0]
        order = np.lexsort((allr[:, 1], allr[:, 0]))[:k]                 # ascending ring distance, ties -> lowest index
        min_dis = np.float32(BIG_DIST)                                   # descriptor.h:1637 (a float)
        min_index, min_bias = -1, 0
        for r in allr[order]:                                            # descriptor.h:1645-1659
            if r[2] < float(min_dis):
                min_dis = np.float32(r[2])
                min_index, min_bias = int(r[1]), int(r[3])
        if float(min_dis) < self.thres:                                  # descriptor.h:1662
            return min_index, float(min_bias), float(min_dis)
        return -1, 0.0, float(min_dis)

    def detect_full(self, cur, values_cur):
        """full-DB mode: (loop_id, nn_idx, shift, dist) over every eligible keyframe of every shard."""
        hi = local_count(cur - self.exclude, self.rank, self.world)
        self.engine.stage_query(values_cur)
        nn, sh, d = self.engine.detect_full_range(-1, 0, hi)
        rec = np.array([[d, float(nn * self.world + self.rank) if nn >= 0 else -1.0, float(sh)]], dtype=np.float64)
        if self.exchange == "allreduce":
            k1, k2 = pack_winner_keys(rec)
            m1 = self._all_reduce_min(k1)
            m2 = self._all_reduce_min(np.where(k1 == m1, k2, I64_MAX))
            d, g, sh = unpack_winner_keys(m1, m2)[0]
            if g < 0:
                return -1, -1, 0, BIG_DIST
            return (g if d < self.thres else -1), g, sh, d
        allr = self._all_gather(rec)
        allr = allr[allr[:, 1] >= 0]
        if len(allr) == 0:
            return -1, -1, 0, BIG_DIST
        best = allr[np.lexsort((allr[:, 1], allr[:, 0]))[0]]
        d, g, sh = float(best[0]), int(best[1]), int(best[2])
        return (g if d < self.thres else -1), g, sh, d


class FullScanStream:
    """Full-DB detection for a stream of scans on a sharded database.

    ``submit(query, lo, hi)`` enqueues one scan's pass over this rank's shard
    (``scl_detect_full_submit``; a few passes are kept in flight).  Local winners are
    collected in submission order; every ``merge_every`` scans their records go into one
    asynchronous all-gather, merged one batch later (global arg-min, ties -> lowest global
    index).  ``drain()`` flushes; ``results`` holds (dist, global_idx, shift) per scan, in
    order, identical on every rank.
    """

    def __init__(self, engine, rank=0, world=1, group=None, device="cpu", depth=2, merge_every=16, scans_per_launch=1,
                 always_exchange=False, native_chunk=0, exchange="allreduce"):
        self.engine, self.rank, self.world, self.group, self.device = engine, rank, world, group, device
        self.exchange = exchange                            # "allreduce": two min all-reduces on packed keys; "allgather": host merge
        self.stage1 = self.stage2 = None                    # all-reduce pipeline: batch b in phase 1, batch b-1 in phase 2
        self.always_exchange = always_exchange              # run the collective even for world == 1 (exercises the backend)
        # native_chunk > 0: database-resident queries are handed to scl_detect_full_stream in chunks of that many
        # scans (the submit / collect pipeline runs in the engine, the Python loop pays once per chunk); each
        # chunk's winners form one exchange batch
        self.native_chunk = native_chunk if hasattr(engine, "detect_full_stream") else 0
        self.depth, self.merge_every = max(1, depth), max(1, merge_every)
        self.per_launch = max(1, scans_per_launch)          # database-resident queries sharing one kernel launch
        self.inflight, self.batch, self.pending, self.results = [], [], None, []
        self.held = []

    def submit(self, query, lo, hi):
        if self.native_chunk > 0 and query >= 0:
            self.held.append((query, lo, hi))
            if len(self.held) >= self.native_chunk:
                self._run_native()
            return
        if self.per_launch > 1 and query >= 0:
            self.held.append((query, lo, hi))
            if len(self.held) >= self.per_launch:
                self._launch_held()
            return
        self._launch_held()
        self._make_room(1)
        self.inflight.append(self.engine.detect_full_submit(query, lo, hi))

    def _run_native(self):
        if not self.held:
            return
        q, lo, hi = zip(*self.held)
        self.held = []
        self._native_call(np.asarray(q, dtype=np.int32), np.asarray(lo, dtype=np.int32), np.asarray(hi, dtype=np.int32))

    def _native_call(self, q, lo, hi):
        nn, sh, d = self.engine.detect_full_stream(q, lo, hi, self.per_launch, self.depth)
        if self.world == 1 and not self.always_exchange and not self.batch and self.pending is None and self.stage1 is None and self.stage2 is None:
            # one shard, nothing in flight: the engine's winners are the result (what _merge would return)
            ok = nn >= 0
            self.results.extend(zip(np.where(ok, d, BIG_DIST).tolist(), np.where(ok, nn, -1).tolist(), np.where(ok, sh, 0).tolist()))
            return
        g = np.where(nn >= 0, nn.astype(np.float64) * self.world + self.rank, -1.0)
        self._exchange(np.stack([np.asarray(d, dtype=np.float64), g, sh.astype(np.float64)], axis=1))

    def submit_many(self, queries, lo, hi):
        """Array form of ``submit`` for database-resident queries (``lo`` / ``hi`` scalars or arrays): the scans go to
        the engine's native pipeline ``native_chunk`` at a time, with no per-scan work in Python."""
        q = np.ascontiguousarray(queries, dtype=np.int32)
        lo = np.ascontiguousarray(np.broadcast_to(np.asarray(lo, dtype=np.int32), q.shape))
        hi = np.ascontiguousarray(np.broadcast_to(np.asarray(hi, dtype=np.int32), q.shape))
        if self.native_chunk <= 0 or (q < 0).any():
            for a, b, c in zip(q.tolist(), lo.tolist(), hi.tolist()):
                self.submit(a, b, c)
            return
        self._launch_held()
        for s0 in range(0, len(q), self.native_chunk):
            s1 = min(len(q), s0 + self.native_chunk)
            self._native_call(q[s0:s1], lo[s0:s1], hi[s0:s1])

    def _launch_held(self):
        if self.native_chunk > 0:
            self._run_native()
            return
        if not self.held:
            return
        q, lo, hi = zip(*self.held)
        self.held = []
        self._make_room(len(q))
        self.inflight.extend(self.engine.detect_full_submit_many(q, lo, hi))

    def _make_room(self, incoming):
        # at most `depth` launches in flight, the new one included (the engine has 8 result slots)
        limit = min(self.depth * self.per_launch, 8)
        while self.inflight and len(self.inflight) + incoming > limit:
            self._collect_one()

    def _collect_one(self):
        nn, sh, d = self.engine.detect_full_collect(self.inflight.pop(0))
        self.batch.append((d, float(nn * self.world + self.rank) if nn >= 0 else -1.0, float(sh)))
        if len(self.batch) >= self.merge_every:
            self._exchange()

    def _exchange(self, rec=None):
        if rec is None:
            if not self.batch:
                return
            rec = np.array(self.batch, dtype=np.float64)
            self.batch = []
        elif self.batch:                                    # per-scan submissions collected so far go first (order)
            self._exchange()
        if self.world == 1 and not self.always_exchange:
            self._merge(rec[None])
            return
        import torch
        import torch.distributed as dist
        if self.exchange == "allreduce":
            # phase 1 of this batch starts now; phase 1 of the previous batch is awaited and its phase 2 started;
            # phase 2 of the batch before that is awaited and delivered.  Nothing blocks on a collective that was
            # enqueued in this call.
            k1, k2 = pack_winner_keys(rec)
            both = torch.from_numpy(np.stack([k1, k2])).to(self.device)       # one copy to the device for both phases
            t1 = both[0].clone()                                              # reduced in place; both[0] keeps this rank's keys
            w1 = dist.all_reduce(t1, op=dist.ReduceOp.MIN, group=self.group, async_op=True)
            prev1, self.stage1 = self.stage1, (w1, t1, both)
            self._advance(prev1)
            return
        t = torch.from_numpy(rec).to(self.device)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=self.device)
        work = dist.all_gather_into_tensor(out.view(-1, t.shape[1]), t, group=self.group, async_op=True)
        prev, self.pending = self.pending, (work, out, t)
        self._finish(prev)

    def _advance(self, prev1):
        """prev1 = a batch whose phase 1 (distance keys) is in flight: await it, start its phase 2 (index | shift of
        the ranks that hold the minimum; masked on the device, no trip to the host between the phases); the batch that
        was in phase 2 is awaited and delivered first (order)."""
        import torch
        import torch.distributed as dist
        prev2, self.stage2 = self.stage2, None
        if prev2 is not None:
            w2, t2, m1 = prev2
            w2.wait()
            both = torch.stack([m1, t2]).cpu().numpy()                          # one copy back for both reduced keys
            self.results.extend(unpack_winner_keys(both[0], both[1]))
        if prev1 is not None:
            w1, t1, both = prev1
            w1.wait()
            t2 = torch.where(both[0] == t1, both[1], torch.full_like(both[1], I64_MAX))
            w2 = dist.all_reduce(t2, op=dist.ReduceOp.MIN, group=self.group, async_op=True)
            self.stage2 = (w2, t2, t1)

    def _finish(self, pending):
        if pending is None:
            return
        work, out, _ = pending
        work.wait()
        self._merge(out.cpu().numpy())

    def _merge(self, allr):
        """allr: (world, m, 3) records (dist, global index or -1, shift): per scan the smallest distance, ties to the
        lowest global index (vectorised: no per-scan work in Python beyond building the tuples)"""
        allr = np.asarray(allr, dtype=np.float64)
        valid = allr[:, :, 1] >= 0
        d = np.where(valid, allr[:, :, 0], np.inf)
        dmin = d.min(axis=0)
        cand = valid & (d == dmin[None, :])
        g = np.where(cand, allr[:, :, 1], np.inf)
        gmin = g.min(axis=0)
        win = np.argmax(cand & (g == gmin[None, :]), axis=0)
        cols = np.arange(allr.shape[1])
        any_valid = valid.any(axis=0)
        out_d = np.where(any_valid, allr[win, cols, 0], BIG_DIST)
        out_g = np.where(any_valid, allr[win, cols, 1], -1.0).astype(np.int64)
        out_s = np.where(any_valid, allr[win, cols, 2], 0.0).astype(np.int64)
        self.results.extend(zip(out_d.tolist(), out_g.tolist(), out_s.tolist()))

    def drain(self):
        self._launch_held()
        while self.inflight:
            self._collect_one()
        self._exchange()
        prev, self.pending = self.pending, None
        self._finish(prev)
        while self.stage1 is not None or self.stage2 is not None:      # flush the all-reduce pipeline
            prev1, self.stage1 = self.stage1, None
            self._advance(prev1)
        return self.results


def verify_candidates_sharded(engine, src, tgts, rank=0, world=1, group=None, device="cpu", params=None):
    """Geometric verification of the loop candidates of one scan across the GPUs of a node:
    the alignments are independent, so rank r verifies candidates r, r + G, r + 2G, ... with
    ``icp_align_batch`` and one all-gather of 19 floats per candidate (4x4 transform, fitness, converged,
    iterations) gives every rank the whole table.  Returns (T[m,4,4], fitness[m], converged[m], iterations[m])."""
    m = len(tgts)
    mine = list(range(rank, m, world))
    per = (m + world - 1) // world
    rec = np.zeros((per, 19), dtype=np.float64)
    rec[:, 17] = -1.0                                                  # padding rows: converged = -1
    if mine:
        T, fit, conv, it = engine.icp_align_batch(src, [tgts[c] for c in mine], params)
        for k in range(len(mine)):
            rec[k, :16] = np.asarray(T[k], dtype=np.float64).reshape(16)
            rec[k, 16], rec[k, 17], rec[k, 18] = float(fit[k]), float(bool(conv[k])), float(it[k])
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(rec).to(device)
        out = torch.empty((world * per, 19), dtype=t.dtype, device=device)
        dist.all_gather_into_tensor(out, t, group=group)
        allr = out.cpu().numpy().reshape(world, per, 19)
    else:
        allr = rec[None]
    Tm = np.zeros((m, 4, 4), np.float32); fit = np.zeros(m, np.float32); conv = np.zeros(m, bool); it = np.zeros(m, np.int32)
    for c in range(m):
        r = allr[c % world, c // world]
        Tm[c] = r[:16].reshape(4, 4); fit[c] = r[16]; conv[c] = r[17] > 0.5; it[c] = int(r[18])
    return Tm, fit, conv, it
