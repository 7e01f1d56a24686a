import sys, os, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csn_amd import _lib
L=_lib; L.build(); lib=L.lib()
lib.csn_set_math_mode(1)
def run(d,T,nb,H=1,E=1,seed=0):
    rng=np.random.default_rng(seed)
    D=H*d; N=T*nb; Tp=(T+31)//32*32
    q=torch.from_numpy(rng.standard_normal((E,D,N)).astype(np.float32)).cuda()*0.3
    k=torch.from_numpy(rng.standard_normal((E,D,N)).astype(np.float32)).cuda()*0.3
    v=torch.from_numpy(rng.standard_normal((E,D,N)).astype(np.float32)).cuda()
    ctx=torch.zeros((E,D,N),device='cuda'); lse=torch.zeros((E,H,N),device='cuda')
    sc=torch.zeros((E,H,nb,T,Tp),device='cuda')
    rc=lib.csn_block_attn_fwd_f32(q.data_ptr(),k.data_ptr(),v.data_ptr(),D*N,D*N,None,None,N,ctx.data_ptr(),D*N,sc.data_ptr(),lse.data_ptr(),E,H,d,T,nb,Tp,8.0,0.0,0,0,0,torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    nan_ctx=torch.isnan(ctx); nan_sc=torch.isnan(sc[...,:T]); nan_lse=torch.isnan(lse)
    print(f"d={d} T={T} nb={nb} H={H} rc={rc} nan ctx {nan_ctx.float().mean().item():.3f} sc {nan_sc.float().mean().item():.3f} lse {nan_lse.float().mean().item():.3f}")
    if nan_ctx.any():
        idx=nan_ctx.nonzero()
        print('  first nan ctx idx', idx[:5].tolist(), 'rows with nan', sorted(set(idx[:,1].tolist()))[:20], 'cols', sorted(set(idx[:,2].tolist()))[:20])
for d,T in [(64,100),(64,128),(64,32),(32,100),(128,100),(64,500)]:
    run(d,T,1)

def run2(d,T):
    rng=np.random.default_rng(0)
    E=1;H=1;nb=1;D=d;N=T;Tp=(T+31)//32*32
    q=torch.ones((E,D,N),device='cuda')*0.1
    k=torch.ones((E,D,N),device='cuda')*0.1
    v=torch.ones((E,D,N),device='cuda')
    ctx=torch.zeros((E,D,N),device='cuda'); lse=torch.zeros((E,H,N),device='cuda')
    sc=torch.zeros((E,H,nb,T,Tp),device='cuda')
    lib.csn_block_attn_fwd_f32(q.data_ptr(),k.data_ptr(),v.data_ptr(),D*N,D*N,None,None,N,ctx.data_ptr(),D*N,sc.data_ptr(),lse.data_ptr(),E,H,d,T,nb,Tp,8.0,0.0,0,0,0,torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s=sc[0,0,0,:,:T]
    bad=(s-0.01*d).abs()>1e-3
    bad=bad|torch.isnan(s)
    keys=sorted(set(bad.nonzero()[:,0].tolist())); qs=sorted(set(bad.nonzero()[:,1].tolist()))
    print(f"d={d} T={T}: bad keys {keys[:40]} ... n={len(keys)}; bad q n={len(qs)} first {qs[:10]}; sample vals {s[keys[0] if keys else 0,:4].tolist()}")
run2(64,64); run2(64,100); run2(128,64)
