import sys, os, ctypes as C
import numpy as np, torch
ROOT="/root/repo"
sys.path[:0]=[os.path.join(ROOT,"gl-abc-mcmc_amd"), os.path.join(ROOT,"tests")]
import oracle_lib
from helpers import descriptors, bits
from test_hip_parity import _fast_run
oracle = oracle_lib.load()
model, local, glob = descriptors(dict(epsilon=0.05, local=("gauss",[0,0],[0.35,0.35]), **{"global":("gauss",[0,0],[1,1])}))
n,T,seed,gf,N,d = 4096,40,20261004,0.9,5,2
rng=np.random.default_rng(7*N+d)
theta0=rng.standard_normal((n,d)).astype(np.float32)
y0=(np.abs(theta0)+0.2236068*rng.standard_normal((n,d))).astype(np.float32)
hist,chains,_,(tu,tr,tz)=_fast_run(model,local,glob,theta0,y0,T,seed,gf,N,chain0=77)
print("zero rows in tz:", (tz==0).all(axis=-1).sum(), "of", tz.shape[0]*tz.shape[1]*tz.shape[2], "per candidate:", (tz==0).all(axis=-1).sum(axis=(0,1)))
print("tu range", tu.min(), tu.max(), "tr range", tr.min(), tr.max(), "tz std", tz.std())
hc=oracle_lib.HostChains(theta0,y0,chain0=77); hh=np.zeros((T,d,n),np.float32)
run,keep=oracle_lib.make_run(seed=seed,step0=1,n_steps=T,gf=gf,batch=N,history=hh,tape=(tu,tr,tz,N))
cs=hc.struct()
assert oracle.oracle_init_weights(C.byref(model),C.byref(glob),C.byref(cs))==0
assert oracle.oracle_glmcmc_steps(C.byref(model),C.byref(local),C.byref(glob),C.byref(cs),C.byref(run))==0
differ=(bits(hist)!=bits(hh)).any(axis=(0,1)); fl=np.flatnonzero(differ)
print("diverged", len(fl))
first=[int(np.flatnonzero((bits(hist[:,:,c])!=bits(hh[:,:,c])).any(axis=1))[0]) for c in fl]
print("first divergence step histogram", np.bincount(first, minlength=T))
for c in fl[:5]:
    t0=int(np.flatnonzero((bits(hist[:,:,c])!=bits(hh[:,:,c])).any(axis=1))[0])
    prev = hist[t0-1,:,c] if t0>0 else theta0[c]
    print("chain",c,"t0",t0,"prev",prev,"kernel",hist[t0,:,c],"oracle",hh[t0,:,c],"branch u",tu[c,t0,0],"global" if tu[c,t0,0]<gf else "local","u_res",tr[c,t0],"u_acc",tu[c,t0,1])
    th = tz[c,t0,:,:2]; print("   candidates theta'", th.tolist())
