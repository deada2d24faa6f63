"""Exact checkpoint / resume of a batched run.

The reference has no resume: its only persistence is the CSV dump of Theta_Re (GLMCMC.py:105-111).  Here a run's
complete state is small and explicit -- the chain-major state arrays, the per-chain streaming sums, and the
position in the counter-based random stream (seed, next iteration index, global id of chain 0) -- so a resumed
run continues bit for bit: iteration i of chain c always draws Philox(seed; c, i, slot).
"""
import torch

from . import _capi, engine

_FIELDS = ("theta", "y", "log_w", "flags", "n_moves", "theta64", "y64", "log_w64", "grad")
_MOMENTS = ("sum_theta", "sum_outer", "sum_jump")


def save(path, chains, seed, next_step, moments=None, extra=None):
    """Write chains (engine.ChainBatch), the Philox position and optional engine.Moments to `path` (torch.save)."""
    state = {"version": 2, "stream_layout": _capi.STREAM_LAYOUT, "seed": int(seed), "next_step": int(next_step), "chain0": chains.chain0,
             "n": chains.n, "d": chains.d, "yd": chains.yd, "extra": extra or {}}
    for f in _FIELDS:
        t = getattr(chains, f, None)
        state[f] = None if t is None else t.detach().cpu()
    if moments is not None:
        state["moments"] = {f: getattr(moments, f).detach().cpu() for f in _MOMENTS}
        state["moments"]["steps"] = moments.steps
    torch.save(state, path)


def load(path, device=None):
    """-> (chains, seed, next_step, moments or None, extra)"""
    dev = engine.require_device(device)
    state = torch.load(path, map_location="cpu", weights_only=True)      # tensors, numbers, strings, dicts only
    layout = state.get("stream_layout", 1)                               # version-1 files predate the tag
    if layout != _capi.STREAM_LAYOUT:
        raise RuntimeError("%s was written on random-stream layout %d, this build draws layout %d (include/glabc.h "
                           "GLABC_STREAM_LAYOUT): the resumed chains would not continue the run that was saved"
                           % (path, layout, _capi.STREAM_LAYOUT))
    chains = engine.ChainBatch.__new__(engine.ChainBatch)
    chains.device, chains.n, chains.d, chains.yd, chains.chain0 = dev, state["n"], state["d"], state["yd"], state["chain0"]
    for f in _FIELDS:
        t = state[f]
        setattr(chains, f, None if t is None else t.to(dev))
    moments = None
    if state.get("moments") is not None:
        moments = engine.Moments.__new__(engine.Moments)
        moments.n, moments.d, moments.steps = state["n"], state["d"], state["moments"]["steps"]
        for f in _MOMENTS:
            setattr(moments, f, state["moments"][f].to(dev))
    return chains, state["seed"], state["next_step"], moments, state["extra"]
