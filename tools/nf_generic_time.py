import sys, time
sys.path[:0]=['gl-abc-mcmc_amd']
import torch, glabcmcmc_amd as g
from glabcmcmc_amd.examples.Mixture import Mixture_set
from glabcmcmc_amd.examples.UserModel import TorchMixture
n=65536
lp = g.DiagGaussian(2, torch.zeros(1, 2), torch.log(torch.tensor([0.35, 0.35])))
th0=torch.zeros(n,2); 
for name,m in (("fused",Mixture_set(0.05)),("generic-torch",TorchMixture(2,0.05))):
    y0=m.generate_samples(th0) if name=="fused" else TorchMixture(2,0.05).generate_samples(th0)
    for rep in range(2):
        torch.cuda.synchronize(); t=time.perf_counter()
        out=g.GLMCMC_NF(m, 101, th0, y0, lp, None, 0.9, 20, 5, None, 1, num_layers=8, seed=1, verbose=False, return_device=True)
        torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(name, "%.1f ms per iteration"%(dt*10), "%.3g chain-steps/s"%(n*100/dt))
