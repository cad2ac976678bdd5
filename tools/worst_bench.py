#!/usr/bin/env python3
"""the bench line's worst-case shape by itself (clip style 4: a third of all P blocks intra), for a kernel trace: tools/ab/worst_prof.sh"""
import importlib, json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import _cabi as A
pkg = importlib.import_module("digital-subband-video-1_amd")
gops = int(sys.argv[1]) if len(sys.argv) > 1 else 160
style = int(sys.argv[2]) if len(sys.argv) > 2 else 4
r = bench.shape_bench(pkg, A, 0, 1920, 1080, A.SUBSAMP_420, gops, 12, 4, 0x10800003, 0, style=style, qp=85, gop=12, rc_mode_cli=1, scd=0)
print(json.dumps(r))
