#!/usr/bin/env python3
"""Streaming write / copy bandwidth of the box (torch fill_ / copy_ on 2.4 GB), for comparison with the trainer's activation stores."""
import torch
x = torch.empty(600 * 1024 * 1024, dtype=torch.float32, device='cuda')
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = x.numel() * 4 / 1e9
ms = t(lambda: x.fill_(1.0)); print(f'fill_  {gb:.2f} GB: {ms:.3f} ms = {gb / ms:.2f} TB/s written')
ms = t(lambda: x.zero_()); print(f'zero_  {gb:.2f} GB: {ms:.3f} ms = {gb / ms:.2f} TB/s written')
ms = t(lambda: y.copy_(x)); print(f'copy_  {gb:.2f} GB: {ms:.3f} ms = {2 * gb / ms:.2f} TB/s read + written')
ms = t(lambda: x.sum()); print(f'sum    {gb:.2f} GB: {ms:.3f} ms = {gb / ms:.2f} TB/s read')
# pieces of rows: the trainer's activation stores write [rows, 256] fp32 buffers in column blocks
R = x.numel() // 256
v = x.view(R, 256)
for w in (16, 32, 64, 128):
    ms = t(lambda: [v[:, c:c + w].fill_(1.0) for c in range(0, 256, w)], n=3)
    print(f'fill_ of {256 // w} column blocks of {w} floats ({4 * w} B pieces at a 1 KiB stride): {ms:.3f} ms = {gb / ms:.2f} TB/s written')
