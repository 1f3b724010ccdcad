"""Diagnostic (not a test): what a plain device-to-device copy reaches on this box with the evaluation kernel's traffic (read 0.8 GB, write
0.8 GB per launch) -- the practical ceiling for a 1:1 read/write stream, next to the 8 TB/s spec.  python tools/copy_roof.py"""
import torch
n = (1 << 18) * 379
a = torch.randn(n, dtype=torch.float64, device="cuda:0"); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("copy of %.2f GB (read + write %.2f GB): %.3f ms, %.0f GB/s" % (n * 8 / 1e9, 2 * n * 8 / 1e9, ms, 2 * n * 8 / ms / 1e6))
c = torch.empty_like(a)
for _ in range(3): torch.add(a, b, out=c)
torch.cuda.synchronize(); e0.record()
for _ in range(10): torch.add(a, b, out=c)
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 10
print("a + b -> c (2 reads, 1 write, %.2f GB): %.3f ms, %.0f GB/s" % (3 * n * 8 / 1e9, ms, 3 * n * 8 / ms / 1e6))
for _ in range(3): s = a.sum()
torch.cuda.synchronize(); e0.record()
for _ in range(10): s = a.sum()
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 10
print("sum(a) (read only, %.2f GB): %.3f ms, %.0f GB/s" % (n * 8 / 1e9, ms, n * 8 / ms / 1e6))
