import sys, os, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csn_amd import _lib
L=_lib; L.build(); lib=L.lib()
lib.csn_set_math_mode(1)
d,T=64,64
E=1;H=1;nb=1;D=d;N=T;Tp=64
q=torch.zeros((E,D,N),device='cuda'); q[:,0,:]=1.0      # only d=0 contributes: S[key][q] = k[0][key]
k=torch.zeros((E,D,N),device='cuda'); k[0,0,:]=torch.arange(N,device='cuda').float()
v=torch.ones((E,D,N),device='cuda')
ctx=torch.zeros((E,D,N),device='cuda'); lse=torch.zeros((E,H,N),device='cuda')
sc=torch.zeros((E,H,nb,T,Tp),device='cuda')
lib.csn_block_attn_fwd_f32(q.data_ptr(),k.data_ptr(),v.data_ptr(),D*N,D*N,None,None,N,ctx.data_ptr(),D*N,sc.data_ptr(),lse.data_ptr(),E,H,d,T,nb,Tp,8.0,0.0,0,0,0,torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(sc[0,0,0,:,0].tolist())
# now d=1 row only
k=torch.zeros((E,D,N),device='cuda'); k[0,5,:]=torch.arange(N,device='cuda').float(); q=torch.zeros((E,D,N),device='cuda'); q[:,5,:]=1.0
lib.csn_block_attn_fwd_f32(q.data_ptr(),k.data_ptr(),v.data_ptr(),D*N,D*N,None,None,N,ctx.data_ptr(),D*N,sc.data_ptr(),lse.data_ptr(),E,H,d,T,nb,Tp,8.0,0.0,0,0,0,torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(sc[0,0,0,:,0].tolist())
