"""NF kernel throughput (forward / sample) for a few row counts: python tools/nf_shapes.py  (needs an MI355X)"""
import ctypes as C, sys, time
sys.path.insert(0, "gl-abc-mcmc_amd")
import torch
from glabcmcmc_amd import _capi
from glabcmcmc_amd.flows import RealNVP
torch.manual_seed(0)
flow = RealNVP(8).cuda()
blob = flow.packed_params(); f = flow.descriptor(blob); lib = _capi.lib()
for rows in (65536, 327680, 655360):
    z = torch.empty(2, rows, device="cuda"); lq = torch.empty(rows, device="cuda")
    for inverse in (False, True):
        def go():
            if inverse: _capi.check(lib.glabc_nf_log_prob(C.byref(f), z.data_ptr(), rows, lq.data_ptr(), None), "lp")
            else: _capi.check(lib.glabc_nf_sample(C.byref(f), None, 1, 0, rows, z.data_ptr(), lq.data_ptr(), None), "s")
        go(); go(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): go()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("rows %8d %s  %.3f ms  %.1f TFLOP/s (MFMA)" % (rows, "inverse" if inverse else "forward", dt * 1e3, 2 * 128 * 128 * 8 * rows / dt / 1e12))
