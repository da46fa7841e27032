"""The reference calls detect* from the loop-closure thread without a lock while the LIO thread and the ROS
spinner append under mtxSC (DM.h:1001-1003, 625-628, 1078): the engine must serialise that itself."""
import threading

import numpy as np
import pytest

from scl_slam_amd import ScanContextEngine
from scl_slam_amd.synth import synth_descriptors

pytestmark = pytest.mark.gpu


def test_append_and_detect_from_two_threads():
    R, S, n0, n1 = 20, 60, 400, 900
    descs = synth_descriptors(n1, R, S, seed=5, revisit_frac=0.1)
    e = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=100, initial_capacity=64)
    try:
        e.save_bulk(descs[:n0])
        curs = list(range(250, n0))
        seen, errors = {}, []

        def appender():
            try:
                for i in range(n0, n1):                      # capacity doubles several times on the way
                    e.save_from_wire(descs[i], 0, i)
            except Exception as ex:                          # noqa: BLE001
                errors.append(ex)

        def detector():
            try:
                for _ in range(3):
                    for c in curs:
                        r = (e.detect_intra(c), e.detect_full_range(c, 0, c - 100))
                        assert seen.setdefault(c, r) == r   # a keyframe's verdict does not depend on later appends
            except Exception as ex:                          # noqa: BLE001
                errors.append(ex)

        ta, td = threading.Thread(target=appender), threading.Thread(target=detector)
        ta.start(); td.start(); ta.join(); td.join()
        assert not errors, errors
        assert e.get_size() == n1
        for c in curs[::7]:                                  # and equals the verdict on the final database
            assert seen[c] == (e.detect_intra(c), e.detect_full_range(c, 0, c - 100))
    finally:
        e.close()


def test_appends_get_in_between_the_chunks_of_a_long_stream_call():
    """VERDICT r3 #6 / missing 6: the reference appends under mtxSC (DM.h:1001-1003, 625-628) while the loop-closure thread detects
    unlocked (DM.h:1078).  A stream call of thousands of scans used to hold the engine's one mutex from start to end; now it keeps
    the lock on the pass buffers but gives the database lock up while it waits for a chunk, so appends land between its chunks --
    capacity doublings included -- and the call still scores exactly the database it was started on."""
    import time
    import oracle_binding as ob
    R, S, n0, extra = 64, 120, 3000, 260
    descs = synth_descriptors(n0 + extra, R, S, seed=11, revisit_frac=0.05)
    e = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=100, initial_capacity=n0 + 40)   # doubles on the way
    db = ob.OracleDB(ob.make_config(R=R, S=S))
    try:
        e.save_bulk(descs[:n0]); db.save_bulk(descs[:n0])
        scans = 8192
        qs = (n0 - 100 + (np.arange(scans) % 100)).astype(np.int32)
        his = (qs - 100).astype(np.int32)                       # D.h:1627: [0, cur - NUM_EXCLUDE_RECENT)
        alone = e.detect_full_stream(qs, 0, his, 16, 2)        # the same call with nobody else around
        for i in (0, 37, 99):                                   # ... which is the checker's verdict
            d_ref, s_ref = db.distance_batch(int(qs[i]), cand=np.arange(0, int(his[i]), dtype=np.int32))
            b = int(np.argmin(d_ref))
            assert (alone[0][i], alone[1][i]) == (b, s_ref[b]) and alone[2][i].view(np.uint64) == d_ref[b].view(np.uint64)
        errors, done_at, started = [], [], threading.Event()
        span = {}

        def appender():
            try:
                started.wait()
                for i in range(n0, n0 + extra):
                    e.save_from_wire(descs[i], 0, i)
                    done_at.append(time.perf_counter())
            except Exception as ex:                              # noqa: BLE001
                errors.append(ex)

        def streamer():
            try:
                started.set()
                span["t0"] = time.perf_counter()
                span["res"] = e.detect_full_stream(qs, 0, his, 16, 2)
                span["t1"] = time.perf_counter()
            except Exception as ex:                              # noqa: BLE001
                errors.append(ex)

        ta, ts = threading.Thread(target=appender), threading.Thread(target=streamer)
        ta.start(); ts.start(); ts.join(); ta.join()
        assert not errors, errors
        got = span["res"]
        assert np.array_equal(got[0], alone[0]) and np.array_equal(got[1], alone[1]) and np.array_equal(got[2].view(np.uint64), alone[2].view(np.uint64)), \
            "the stream call must score the database it was started on, whatever is appended meanwhile"
        inside = sum(1 for t in done_at if span["t0"] < t < span["t1"])
        assert inside >= 3, f"appends completed inside the stream call's {1e3 * (span['t1'] - span['t0']):.1f} ms: {inside} (they used to wait for its end)"
        # the appended keyframes are there, and what is detected for them is the checker's verdict
        assert e.get_size() == n0 + extra
        db.save_bulk(descs[n0:])
        for cur in (n0 + 5, n0 + extra - 1):
            g = e.detect_full(cur); o = db.detect_full(cur)
            assert g[:3] == o[:3] and np.float64(g[3]).view(np.uint64) == np.float64(o[3]).view(np.uint64), (cur, g, o)
    finally:
        e.close(); db.close()


def test_staged_queries_survive_a_capacity_doubling_between_the_chunks_of_a_stream_call():
    """a staged query (id -1) lives in a row BEHIND the database slots (index cap + j), so its slot number changes when an
    append doubles the capacity while a stream call waits for a chunk: the next chunk's alignment was enqueued with the old slot, its
    products are built with the new one.  The stream's scans alternate between the staged query and a keyframe holding the SAME
    descriptor; with appends doubling the capacity twice inside the call every staged scan must still equal its keyframe twin."""
    R, S, n0, extra = 64, 120, 1500, 1700
    descs = synth_descriptors(n0 + extra, R, S, seed=12, revisit_frac=0.05)
    e = ScanContextEngine(num_ring=R, num_sector=S, num_candidates=3, num_exclude_recent=100, initial_capacity=n0 + 8)   # 1508 -> 3016 -> 6032
    try:
        e.save_bulk(descs[:n0])
        twin = n0 - 7
        e.stage_query(descs[twin])
        scans = 6144
        qs = np.where(np.arange(scans) % 2 == 0, -1, twin).astype(np.int32)
        his = np.full(scans, n0 - 100, dtype=np.int32)
        alone = e.detect_full_stream(qs, 0, his, 16, 2)
        assert np.array_equal(alone[0][0::2], alone[0][1::2]) and np.array_equal(alone[2][0::2].view(np.uint64), alone[2][1::2].view(np.uint64))
        errors, started, span = [], threading.Event(), {}

        def appender():
            try:
                started.wait()
                for i in range(n0, n0 + extra):
                    e.save_from_wire(descs[i], 0, i)
            except Exception as ex:                              # noqa: BLE001
                errors.append(ex)

        def streamer():
            try:
                started.set()
                span["res"] = e.detect_full_stream(qs, 0, his, 16, 2)
            except Exception as ex:                              # noqa: BLE001
                errors.append(ex)

        ta, ts = threading.Thread(target=appender), threading.Thread(target=streamer)
        ta.start(); ts.start(); ts.join(); ta.join()
        assert not errors, errors
        got = span["res"]
        assert np.array_equal(got[0], alone[0]) and np.array_equal(got[1], alone[1]) and np.array_equal(got[2].view(np.uint64), alone[2].view(np.uint64))
        assert e.get_size() == n0 + extra
        # the staged query is still where the engine says it is
        assert e.detect_full_range(-1, 0, n0 - 100) == e.detect_full_range(twin, 0, n0 - 100)
    finally:
        e.close()
