#!/usr/bin/env python
"""Latency of a cross-stream event wait on this ROCm stack: stream A runs a ~200 us kernel and records an event, stream B (idle)
waits for it and runs a tiny kernel; printed: time from A's event to B's first kernel start (HIP events), for both directions and
for B = torch's current stream.  Run with and without HSA_ENABLE_INTERRUPT=0 / GPU_MAX_HW_QUEUES.

    python tools/micro/stream_wait_probe.py
"""
import os
import statistics
import sys

import torch


def probe(a, b, n=200):
    x = torch.randn(64 << 20, device="cuda")      # ~200 us of HBM traffic per pass
    y = torch.empty_like(x)
    tiny = torch.zeros(64, device="cuda")
    gaps = []
    for _ in range(n):
        e1 = torch.cuda.Event(enable_timing=True)
        e2 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(a):
            torch.mul(x, 1.0001, out=y)
            e1.record(a)
        b.wait_event(e1)
        with torch.cuda.stream(b):
            tiny.add_(1.0)
            e2.record(b)
        torch.cuda.synchronize()
        gaps.append(e1.elapsed_time(e2) * 1e3)
    gaps.sort()
    return gaps[len(gaps) // 10], statistics.median(gaps), gaps[-len(gaps) // 10]


def main():
    torch.cuda.set_device(0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    print("env:", {k: os.environ.get(k) for k in ("HSA_ENABLE_INTERRUPT", "GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG")})
    for name, a, b in (("side -> side", s1, s2), ("side -> current", s1, cur), ("current -> side", cur, s1)):
        p10, med, p90 = probe(a, b)
        print(f"{name:18s}: event on A -> end of B's tiny kernel  p10 {p10:7.1f}  median {med:7.1f}  p90 {p90:7.1f} us")


if __name__ == "__main__":
    main()
