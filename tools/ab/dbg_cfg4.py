import sys, importlib; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, _cabi as A
import bench
pkg = importlib.import_module("digital-subband-video-1_amd")
for (w,h) in ((352,288),(3840,2160)):
    fmt,gop,n=0x5,12,3
    clip=A.gen_clip(w,h,fmt,0x21600004,gop,style=0)
    cli=dict(qp=85,gop=gop,rc_mode_cli=1,scd=0)
    batch_in=np.empty((n,gop,A.frame_bytes(w,h,fmt)),dtype=np.uint8); batch_in[:]=clip
    b=pkg.Batch(pkg.make_encoder_cfg(w,h,fmt,**cli),n,gop)
    d=b.upload(batch_in)
    for s in range(n): b.set_fnum(s,s*gop)
    first=[bytes(o) for o in b.encode(d,on_device=True)]
    b.close()
    joined=pkg.concat_gops(first)
    fresh,kind=bench.ref_encode(pkg,A,clip,w,h,fmt,**cli)
    want=bench.joined_gops(A,fresh,n,gop,eos=True)
    print(w,h,kind,len(joined),len(want),joined==want, first[0]==fresh)
    pj=A.split_packets(joined); pw=A.split_packets(want)
    print(len(pj),len(pw))
    for i,(a,c) in enumerate(zip(pj,pw)):
        if a!=c:
            dd=[k for k in range(min(len(a),len(c))) if a[k]!=c[k]][:10]
            print(i,len(a),len(c),a[:20].hex(),c[:20].hex(),dd); break
