"""NF coupling kernel against the number of 64-row pairs a workgroup holds: python tools/nf_pairs_sweep.py  (needs an MI355X)

One workgroup per CU, 12 wavefronts = 3 per SIMD.  P pairs per workgroup are P / 4 per SIMD; the question is how much of the
gap to the matrix peak is the tail of a coupling in which a SIMD has fewer than three wavefronts left with work (P = 20: 2 | 2 | 1
pairs on a SIMD's wavefronts) and how much is there at every P."""
import ctypes as C, sys, time
sys.path.insert(0, "gl-abc-mcmc_amd")
import torch
from glabcmcmc_amd import _capi
from glabcmcmc_amd.flows import RealNVP
torch.manual_seed(0)
flow = RealNVP(8).cuda()
blob = flow.packed_params(); f = flow.descriptor(blob); lib = _capi.lib()
pairs = [int(a) for a in sys.argv[1:]] or [4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48, 60]
for P in pairs:
    rows = 256 * 64 * P
    z = torch.empty(2, rows, device="cuda"); lq = torch.empty(rows, device="cuda")
    for inverse in (False, True):
        def go():
            if inverse: _capi.check(lib.glabc_nf_log_prob(C.byref(f), z.data_ptr(), rows, lq.data_ptr(), None), "lp")
            else: _capi.check(lib.glabc_nf_sample(C.byref(f), None, 1, 0, rows, z.data_ptr(), lq.data_ptr(), None), "s")
        go(); go(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): go()
        e1.record(); torch.cuda.synchronize(); dt = e0.elapsed_time(e1) / 10 * 1e-3
        print("pairs/wg %3d rows %8d %s  %.3f ms  %.1f TFLOP/s (MFMA)  %.2f us per pair-coupling per SIMD" % (
            P, rows, "inverse" if inverse else "forward", dt * 1e3, 2 * 128 * 128 * 8 * rows / dt / 1e12, dt * 1e6 / 8 / ((P + 3) // 4)), flush=True)
