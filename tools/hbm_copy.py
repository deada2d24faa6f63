"""Attainable HBM copy bandwidth on the box (SURVEY.md 8d asks for the roofline fraction against it as well)."""
import time, torch
n = 1 << 30                                   # 4 GiB per buffer
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.ones(n, dtype=torch.float32, device="cuda")
for _ in range(3): a.copy_(b)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): a.copy_(b)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("device-to-device copy of %.1f GiB: %.3f ms, %.0f GB/s (read + write)" % (n * 4 / 2**30, dt * 1e3, 2 * n * 4 / dt / 1e9))
a.fill_(0.0); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): a.fill_(1.0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("fill of %.1f GiB: %.3f ms, %.0f GB/s (write only)" % (n * 4 / 2**30, dt * 1e3, n * 4 / dt / 1e9))
