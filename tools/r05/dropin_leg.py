#!/usr/bin/env python3
"""bench.py's `dropin` leg alone (the reference program, and the same program relinked with the csadp drop-in in both modes)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

sets = tuple(sys.argv[1:]) or ("Primates", "Mammals", "Set3")
print(json.dumps(bench.dropin_leg(sets), indent=1))
