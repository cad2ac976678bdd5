#!/usr/bin/env python3
"""the batched decoder shape of bench.py on its own (for rocprofv3): decode_bench.py [streams=64] [reps=4]"""
import importlib, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import _cabi as A
pkg = importlib.import_module("digital-subband-video-1_amd")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
print(json.dumps(bench.decode_bench(pkg, A, 0, S, R)))
