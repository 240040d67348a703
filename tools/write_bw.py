#!/usr/bin/env python3
"""Write-only HBM bandwidth of this GPU (torch fill of a 4.7 GB buffer): the ceiling for the dynamics kernel, whose
traffic is 94 % stores."""
import torch
n = 4681 * 1024 * 1024 // 8
x = torch.empty(n, dtype=torch.float64, device="cuda:0")
for _ in range(3):
    x.fill_(1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    x.fill_(2.0)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("fill %.2f GB in %.3f ms: %.0f GB/s" % (n * 8 / 1e9, ms, n * 8 / ms / 1e6))
y = torch.empty_like(x)
for _ in range(2):
    y.copy_(x)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    y.copy_(x)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("copy %.2f GB read + %.2f GB written in %.3f ms: %.0f GB/s" % (n * 8 / 1e9, n * 8 / 1e9, ms, 2 * n * 8 / ms / 1e6))
