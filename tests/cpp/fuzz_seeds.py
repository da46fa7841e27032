#!/usr/bin/env python3
"""Seed corpus for tests/cpp/fuzz_host.cpp: one well-formed input per target (first byte = target), built with `struct` the way
tests/test_messages.py builds its hand-made wire bytes, so the fuzzer starts from inputs that reach the deep branches."""
import os
import struct
import sys

out = sys.argv[1]
os.makedirs(out, exist_ok=True)


def header(seq=7, sec=1, nsec=2, frame=b"odom"):
    return struct.pack("<III", seq, sec, nsec) + struct.pack("<I", len(frame)) + frame


def transform():
    return struct.pack("<7d", 1.0, 2.0, 3.0, 0.0, 0.0, 0.0, 1.0)


vals = struct.pack("<6f", *range(6))
gd = header() + struct.pack("<i", 5) + transform() + transform() + struct.pack("<I", 6) + vals
li = header() + struct.pack("<4i", 0, 1, 10, 20) + struct.pack("<f", 0.25) + transform()
fields = b""
for name, off in ((b"x", 0), (b"y", 4), (b"z", 8), (b"intensity", 16)):
    fields += struct.pack("<I", len(name)) + name + struct.pack("<IBI", off, 7, 1)
pts = bytes(range(64))
cloud = header() + struct.pack("<II", 1, 2) + struct.pack("<I", 4) + fields + struct.pack("<B", 0) + struct.pack("<II", 32, 64) + struct.pack("<I", 64) + pts + struct.pack("<B", 1)
req = struct.pack("<4i", 3, 9, 0, 1) + cloud
resp = struct.pack("<B", 1) + transform()
cells = 24
db = b"SCLDB\x00\x00\x01" + struct.pack("<4i", 1, 4, 6, 2) + struct.pack("<4i", 1, 3, 2, 0) + struct.pack(f"<{2 * cells}f", *range(2 * cells)) + struct.pack("<4i", 0, 0, 1, 5)
for i, b in enumerate((gd, li, req, resp, db)):
    open(os.path.join(out, f"seed{i}"), "wb").write(bytes([i]) + b)
