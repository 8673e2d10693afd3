import sys, numpy as np
sys.path.insert(0, '.')
import path_tracer_ocaml_amd as P
from oracle import oracle as O
w,h,spp=600,300,32
d=O.desc_shirley(w,h); o=O.Scene(d.ptr,d); g=P.Scene(d.ptr,0,keepalive=d)
A=d.arrays()
rng=np.random.default_rng(3); n=20000
xs,ys,ps=rng.integers(0,w,n),rng.integers(0,h,n),rng.integers(0,spp,n)
for depth in (1,2,3,8):
    c,_=o.trace_samples(w,h,spp,depth,xs,ys,ps); gg,_=g.trace_samples(w,h,spp,depth,xs,ys,ps)
    bad=(c.view(np.uint64)!=gg.view(np.uint64)).any(axis=1)
    print('depth',depth,'bad',bad.sum())
    if bad.sum() and depth<=3:
        idx=np.nonzero(bad)[0][:12]
        for i in idx:
            print('  sample',i,xs[i],ys[i],ps[i],'cpu',c[i],'gpu',gg[i])
        # first-hit material kinds of bad samples
        cam=A['camera']
        alpha=np.zeros(2+2*depth); O.lib().orc_lds_alpha(2+2*depth,alpha.ctypes.data_as(O.dp))
        break
