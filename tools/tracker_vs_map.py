#!/usr/bin/env python
"""Times the tracker against map-sized targets (bench.py's `legs.tracker_vs_map`, stand-alone): `python tools/tracker_vs_map.py [K ...]`."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    sizes = tuple(int(a) for a in sys.argv[1:]) or (8280, 100_000, 1_000_000)
    print(json.dumps(bench.tracker_vs_map_leg(sizes), indent=1))
