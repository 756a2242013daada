"""Practical HBM bandwidth of the box: device-to-device copy of 4 GiB (read + write counted), best of 10, with torch on the default stream."""
import torch
n = 1 << 30
a = torch.empty(n, dtype=torch.int32, device="cuda")
b = torch.empty_like(a)
a.fill_(1)
best = 0.0
for _ in range(10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); b.copy_(a); e.record(); torch.cuda.synchronize()
    best = max(best, 2 * 4 * n / (s.elapsed_time(e) * 1e-3) / 1e12)
print(f"d2d copy: {best:.2f} TB/s (read + write)")
