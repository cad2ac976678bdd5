#!/usr/bin/env python3
"""the bench line's intra-only shape (config 2: 1080p, every picture an I picture) by itself, for a kernel trace: tools/ab/intra_prof.sh"""
import importlib, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import _cabi as A
pkg = importlib.import_module("digital-subband-video-1_amd")
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 64
r = bench.shape_bench(pkg, A, 0, 1920, 1080, A.SUBSAMP_420, streams, 12, 4, 0x10800002, 0, qp=85, gop=0, rc_mode_cli=1)
print(json.dumps(r))
