#!/usr/bin/env python3
"""The 80x180 full-database pass of bench.py's secondary on its own (for rocprofv3 --kernel-trace --stats)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
print(json.dumps(bench.secondary_80x180(0)))
