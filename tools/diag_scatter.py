import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import path_tracer_ocaml_amd as P
from oracle import oracle as O
w,h,spp,depth=600,300,32,8
d=O.desc_shirley(w,h); o=O.Scene(d.ptr,d); g=P.Scene(d.ptr,0,keepalive=d)
rng=np.random.default_rng(3); n=20000
xs,ys,ps=[np.ascontiguousarray(a,dtype=np.int32) for a in (rng.integers(0,w,n),rng.integers(0,h,n),rng.integers(0,spp,n))]
ip=O.ip; dp=O.dp
def run_cpu():
    ray=np.zeros((n,6)); att=np.zeros((n,3)); alive=np.zeros(n,dtype=np.int32); info=np.zeros((n,3),dtype=np.int32)
    L=O.lib(); L.orc_debug_first_scatter.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_int,C.c_int,C.c_int64,ip,ip,ip,dp,dp,ip,ip]
    L.orc_debug_first_scatter(o._h,w,h,spp,depth,n,xs.ctypes.data_as(ip),ys.ctypes.data_as(ip),ps.ctypes.data_as(ip),ray.ctypes.data_as(dp),att.ctypes.data_as(dp),alive.ctypes.data_as(ip),info.ctypes.data_as(ip))
    return ray,att,alive,info
def run_gpu():
    ray=np.zeros((n,6)); att=np.zeros((n,3)); alive=np.zeros(n,dtype=np.int32)
    L=P.lib(); from path_tracer_ocaml_amd import abi
    L.ptx_debug_first_scatter.argtypes=[C.c_void_p,C.POINTER(abi.RenderParams),C.c_int64,ip,ip,ip,dp,dp,ip]
    p=P.render_params(w,h,spp,depth)
    rc=L.ptx_debug_first_scatter(g._h,C.byref(p),n,xs.ctypes.data_as(ip),ys.ctypes.data_as(ip),ps.ctypes.data_as(ip),ray.ctypes.data_as(dp),att.ctypes.data_as(dp),alive.ctypes.data_as(ip))
    assert rc==0,P.last_error()
    return ray,att,alive
cr,ca,cal,info=run_cpu(); gr,ga,gal=run_gpu()
print('alive mismatch',(cal!=gal).sum())
both=(cal==1)&(gal==1)
badray=both&((cr.view(np.uint64)!=gr.view(np.uint64)).any(axis=1))
badatt=both&((ca.view(np.uint64)!=ga.view(np.uint64)).any(axis=1))
print('bad ray',badray.sum(),'bad attn',badatt.sum())
for i in np.nonzero(badray|(cal!=gal))[0][:10]:
    print(i,xs[i],ys[i],ps[i],'info prim/mat/scatter',info[i],'alive',cal[i],gal[i]); print('  cpu',cr[i]); print('  gpu',gr[i])
import collections
print(collections.Counter(map(tuple,info[badray|(cal!=gal)][:,1:])))
