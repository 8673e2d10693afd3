#!/usr/bin/env python3
"""What a 1080p f64 framebuffer (49.8 MB) costs to bring to pinned host memory: one hipMemcpyAsync against 2 / 4 pieces on as many
streams (SDMA engines in parallel) -- the floor under ptx_render's host-framebuffer step.  usage: tools/d2h_rate.py"""
import time
import torch

n = 1920 * 1080 * 3
src = torch.randn(n, dtype=torch.float64, device="cuda")
dst = torch.empty(n, dtype=torch.float64).pin_memory()
for parts in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    step = (n + parts - 1) // parts

    def go():
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                dst[k * step:(k + 1) * step].copy_(src[k * step:(k + 1) * step], non_blocking=True)
        torch.cuda.synchronize()

    for _ in range(3):
        go()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        go()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("%d piece(s): %.3f ms  %.1f GB/s" % (parts, ms, n * 8 / ms * 1e-6))
