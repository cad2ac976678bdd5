import sys,importlib,os
sys.path.insert(0,"tests"); sys.path.insert(0,".")
import bench, _cabi as A
pkg=importlib.import_module("digital-subband-video-1_amd")
r=bench.shape_bench(pkg, A, 0, 3840, 2160, 0x0, 2, 30, 4, 0x21600005, 0, qp=85, gop=30, rc_mode_cli=0, kbps=20000)
print(os.environ.get("DSV1_NO_FETCH_FAST","fast"), r["Mpix_s"], r["ms_per_step"])
