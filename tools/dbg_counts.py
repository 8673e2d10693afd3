import sys; sys.path.insert(0,'.')
import torch, path_tracer_ocaml_amd as P
from path_tracer_ocaml_amd import host as H
w,h,spp,depth=1920,1080,8,8
hs=H.shirley_spheres(w,h); sc=P.Scene(hs.ptr,0,keepalive=hs)
raw=torch.zeros((h,w,3),dtype=torch.float64,device='cuda')
for d in (1,8):
    p=P.render_params(w,h,spp,d,count_work=True)
    st=sc.render_raw_device(p,raw.data_ptr())
    print('depth',d,{k:st[k] for k in ('segments','nodes_tested','prims_tested')})
