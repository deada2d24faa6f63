"""Runs longer than HBM: the chain history streamed to a .npy file while the sampler keeps running
("history chain-major from HBM -> pinned host -> CSV/NPY in a side thread").

Two device blocks of K iterations and two pinned host blocks rotate: while the fused kernel fills block b on the
compute stream, block 1-b travels over PCIe on a copy stream and a writer thread moves the previous pinned block into
the output file.  The numbers are those of one uninterrupted run (the Philox counter carries the iteration
index), whatever K is.  Layout of the file: float32 [n_steps][d][C] -- the kernels' chain-major rows, row t = iteration
step0 + t.
"""
import queue
import threading

import numpy as np
import torch

from . import engine


def stream_history(entry, model_desc, local_desc, global_desc, chains, n_steps, seed, global_frequency, batch_size, path,
                   step0=1, block=2000, moments=None):
    """Advance `chains` by n_steps iterations of `entry` ('glabc_glmcmc_steps' / 'glabc_globalmcmc_steps'), writing every
    history row to the .npy file `path`.  Returns the (n_steps, d, C) shape written."""
    dev = chains.device
    d, n = chains.d, chains.n
    k = int(min(block, n_steps))
    shape = (int(n_steps), d, n)
    # blocks arrive in order: a .npy header and plain sequential writes (first-touching a memmap is 2x slower)
    out = open(path, "wb")
    np.lib.format.write_array_header_1_0(out, {"descr": "<f4", "fortran_order": False, "shape": shape})
    dev_blocks = [torch.empty(k, d, n, dtype=torch.float32, device=dev) for _ in range(2)]
    host_blocks = [torch.empty(k, d, n, dtype=torch.float32, pin_memory=True) for _ in range(2)]
    copy_stream = torch.cuda.Stream(dev)
    filled = [torch.cuda.Event(), torch.cuda.Event()]          # compute finished writing dev_blocks[b]
    copied = [torch.cuda.Event(), torch.cuda.Event()]          # dev_blocks[b] has reached host_blocks[b]
    work = queue.Queue(maxsize=1)
    host_free = [threading.Event(), threading.Event()]
    for e in host_free:
        e.set()
    errors = []

    def writer():
        while True:
            item = work.get()
            if item is None:
                return
            b, t0, rows = item
            try:
                copied[b].synchronize()
                out.write(memoryview(host_blocks[b][:rows].numpy()).cast("B"))
            except Exception as exc:                           # surfaced by the main thread after the loop
                errors.append(exc)
            finally:
                host_free[b].set()

    th = threading.Thread(target=writer, daemon=True)
    th.start()
    compute = torch.cuda.current_stream(dev)
    done, i = 0, 0
    try:
        while done < n_steps:
            b = i & 1
            rows = min(k, n_steps - done)
            compute.wait_event(copied[b])                      # the copy that last read dev_blocks[b] is over
            engine.run_steps(entry, model_desc, local_desc, global_desc, chains, rows, step0 + done, seed, global_frequency,
                             batch_size, history=dev_blocks[b], moments=moments, steps_per_launch=rows)
            filled[b].record(compute)
            host_free[b].wait()                                # the writer is done with host_blocks[b]
            host_free[b].clear()
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(filled[b])
                host_blocks[b][:rows].copy_(dev_blocks[b][:rows], non_blocking=True)
                copied[b].record(copy_stream)
            work.put((b, done, rows))
            done += rows
            i += 1
    finally:
        # also when a launch raises: stop the writer thread and close the file
        work.put(None)
        th.join()
        out.close()
    if errors:
        raise errors[0]
    return shape
