import os, sys, ctypes as C, importlib, subprocess
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import numpy as np, _cabi as A
pkg=importlib.import_module("digital-subband-video-1_amd")
L=pkg.lib()
for style,seed in ((0,0x10800003),(5,0x10800333),(1,0x1080abc),(2,0x1080abd)):
    W,H,FMT=1920,1080,0x5
    clip=A.gen_clip(W,H,FMT,seed,12,style=style)
    b=pkg.Batch(pkg.make_encoder_cfg(W,H,FMT,qp=85,gop=12,rc_mode_cli=1),1,12)
    z=(C.c_ulonglong*8)(); L.dsvg_hme_counts(z); before=list(z)
    b.encode(clip.reshape(1,12,-1)); b.close()
    L.dsvg_hme_counts(z); d=[z[i]-before[i] for i in range(8)]
    print("style %d: blocks with candidates %d: window %d, none valid %d, too wide %d | +-1 searches %d: from window %d, only zero vector %d" % (style,d[0],d[1],d[2],d[3],d[4],d[5],d[6]))
