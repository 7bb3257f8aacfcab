"""python tools/box_probe.py [reps]: the box probe of bench.py on its own (include/scfgp_hip.h: scfgp_box_probe)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    b = bench.box_probe(0); b.pop('what', None)
    print(json.dumps(b))
