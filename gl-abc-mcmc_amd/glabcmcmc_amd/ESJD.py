"""esjd -- Expected Square Jump Distance (reference: ESJD.py:2-25).

``esjd(data)`` with ``data`` of shape (N, D) returns a NumPy 0-d float32 like the
reference; (N, C, D) returns an array of C values (one per chain).  The reduction
runs in the gfx950 kernel behind ``glabc_esjd`` whatever device ``data`` lives on
(a CPU tensor is staged to the GPU); there is no CPU implementation.
"""
import ctypes as C

import torch

from . import _capi, engine


def esjd_per_chain(history_cm):
    """history_cm: float32 CUDA tensor [n_rows][d][n_chains] (chain-major) -> float32 [n_chains]"""
    n_rows, d, n = history_cm.shape
    h = history_cm.contiguous()
    out = torch.empty(n, dtype=torch.float32, device=h.device)
    stream = torch.cuda.current_stream(h.device).cuda_stream
    with torch.cuda.device(h.device):
        _capi.check(_capi.lib().glabc_esjd(h.data_ptr(), n_rows, d, n, n, out.data_ptr(), C.c_void_p(stream)),
                    "glabc_esjd")
    return out


def esjd(data):
    data = torch.as_tensor(data)
    dev = data.device if data.is_cuda else engine.require_device(None)
    x = data.detach().to(device=dev, dtype=torch.float32)
    if x.dim() == 2:
        return esjd_per_chain(x.unsqueeze(-1))[0].cpu().numpy()
    if x.dim() == 3:
        return esjd_per_chain(x.permute(0, 2, 1)).cpu().numpy()
    raise ValueError("esjd expects (N, D) or (N, C, D)")
