"""Host-side pieces the sampler functions share: argument normalisation, the
reference's CSV side effect and its end-of-run summary print."""
import csv

import torch

from . import engine


def prepare(ABCset, Initial_theta, Initial_y, device, chain0):
    dev = engine.require_device(device)
    theta0 = torch.as_tensor(Initial_theta, dtype=torch.float32).detach().cpu()
    single = theta0.dim() == 1 or theta0.shape[0] == 1
    chains = engine.ChainBatch(theta0, torch.as_tensor(Initial_y).detach().cpu(), dev, chain0=chain0)
    if chains.d != ABCset.theta_dim:
        raise ValueError("Initial_theta has %d columns, Model.theta_dim is %d" % (chains.d, ABCset.theta_dim))
    return dev, chains, single


def allocate_history(num_ite, chains, record_history):
    """Theta_Re in chain-major layout [num_ite][d][C]; row 0 = Initial_theta (GLMCMC.py:56-57)."""
    if not record_history:
        return None
    hist = torch.empty(num_ite, chains.d, chains.n, dtype=torch.float32, device=chains.device)
    hist[0].copy_(chains.theta)
    return hist


class HostMirror:
    """Theta_Re on its way to host memory WHILE the sampler runs (the reference returns a CPU tensor, GLMCMC.py:137).  A large
    history leaves the device at PCIe speed (57 GB/s into pinned memory) -- several times the time the
    kernels need to produce it -- so the copy must not wait for the end of the run: after every launch its rows go to a pinned
    buffer on a second stream while the next launch computes.  The fused wrappers use it by default for histories of 16 MiB and
    more that are returned to the host (many chains); engine.run_steps / run_glmala_steps call rows_done()."""
    MIN_ELEMS = 1 << 22
    LAUNCH_BYTES = 64 << 20                                        # rows per launch: about this much history, so that copies and
                                                                   # kernels alternate often enough to overlap
    def __init__(self, hist):
        self.hist = hist                                           # [num_ite][d][C] on the device
        self.host = torch.empty(hist.shape, dtype=hist.dtype, pin_memory=True)
        self.stream = torch.cuda.Stream(hist.device)
        self.rows = 0                                              # rows already handed to the copy stream
        self.rows_done(1)                                          # row 0 = Initial_theta

    @staticmethod
    def wanted(hist, single, return_device):
        return hist is not None and not single and not return_device and hist.numel() >= HostMirror.MIN_ELEMS

    def steps_per_launch(self, requested):
        if requested:
            return requested
        return max(8, self.LAUNCH_BYTES // (4 * self.hist.shape[1] * self.hist.shape[2]))

    def rows_done(self, upto):
        """rows [self.rows, upto) of the history have been enqueued on the current stream: copy them behind it"""
        if upto <= self.rows:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.hist.device))
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            self.host[self.rows:upto].copy_(self.hist[self.rows:upto], non_blocking=True)
        self.rows = upto

    def result(self):
        self.rows_done(self.hist.shape[0])
        self.stream.synchronize()
        return self.host


def finish(hist, chains, single, filelocation, csv_variant, verbose, return_device, mirror=None):
    """What the reference does after its loop: CSV rows, summary print, return Theta_Re."""
    if hist is None:
        return None
    if mirror is not None:                                         # the rows are (almost) in host memory already
        host = mirror.result().permute(0, 2, 1)                    # (num_ite, C, d) view of the chain-major host buffer
        if filelocation is not None:
            write_csv(host.reshape(host.shape[0], -1), filelocation, csv_variant)
        return host
    if single:
        Theta_Re = hist[:, :, 0].cpu()                             # (num_ite, d) float32 CPU, as the reference returns
        if filelocation is not None:
            write_csv(Theta_Re, filelocation, csv_variant)
        if verbose:
            print_summary(Theta_Re)
        return Theta_Re.to(chains.device) if return_device else Theta_Re
    out = hist.permute(0, 2, 1)                                    # (num_ite, C, d) view of the chain-major buffer
    if filelocation is not None:
        write_csv(out.reshape(out.shape[0], -1).cpu(), filelocation, csv_variant)
    if return_device:
        return out
    if hist.numel() < (1 << 22):
        return out.cpu()
    # large histories: one contiguous DMA into pinned memory (57 GB/s against 5-8 GB/s for a pageable, permuting
    # copy -- DESIGN.md section 5); the result is the (num_ite, C, d) view of that chain-major host buffer
    host = torch.empty(hist.shape, dtype=hist.dtype, pin_memory=True)
    host.copy_(hist, non_blocking=True)
    torch.cuda.synchronize(hist.device)
    return host.permute(0, 2, 1)


def write_csv(Theta_Re, filelocation, variant):
    """The reference's progress dump: header row = theta0, then blocks of rows appended
    every 10 000 iterations and at the end.  variant 'glmcmc' is GLMCMC.py:105-111
    (k = (i-1)//10000); variant 'global' is GlobalMCMC.py:70-76 == GLMALA.py:201-207 ==
    GLMCMC_NFs.py:153-159 (k = i//10000, start (k-1)*10000+1), which re-writes the previous
    block at the tail -- reproduced, since downstream scripts read the file as written.  variant 'aglmcmc' is
    AGLMCMC.py:275-288: blocks of 10 000 rows while the loop runs, the tail after it -- and nothing at all for the last
    block when num_ite - 1 is a multiple of 10 000 (neither clause fires; reproduced as well)."""
    num_ite = Theta_Re.shape[0]
    rows = Theta_Re.numpy()
    with open(filelocation, "w", newline="", encoding="utf-8") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        if variant == "aglmcmc":
            for i in range(1, num_ite):
                if i % 10000 == 0 and i + 1 < num_ite:                          # AGLMCMC.py:276-280
                    w.writerows(rows[max(1, (i // 10000 - 1) * 10000 + 1):i + 1])
            i = num_ite - 1
            if i >= 1 and i % 10000 != 0:                                       # AGLMCMC.py:283-288
                w.writerows(rows[max(1, (i // 10000) * 10000 + 1):i + 1])
            return
        for i in range(1, num_ite):
            if i % 10000 == 0 or i == num_ite - 1:
                if variant == "glmcmc":
                    start = max(1, ((i - 1) // 10000) * 10000 + 1)
                else:
                    start = max(1, (i // 10000 - 1) * 10000 + 1)
                for j in range(start, i + 1):
                    w.writerow(rows[j])


def print_summary(Theta_Re):
    """GLMCMC.py:113-136: per-coordinate mean, variance and 1.96-sigma interval."""
    means = torch.mean(Theta_Re, dim=0)
    variances = torch.var(Theta_Re, dim=0)
    for i in range(Theta_Re.size(1)):
        mean = means[i].item()
        margin = 1.96 * torch.std(Theta_Re[:, i])
        print(f"Theta_Re {i + 1}:")
        print(f"  Mean: {mean:.4f}")
        print(f"  Variance: {variances[i].item():.4f}")
        print(f"  95% Confidence Interval: {(mean - margin, mean + margin)}")
