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
