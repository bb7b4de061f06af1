#!/usr/bin/env python
"""Is a step host-bound?  Host time to ENQUEUE one step against the time until the device has finished it, for the three bench legs.

    python tools/host_probe.py
"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch


def main():
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for name, args in (("cfg 2 (r18 fp32 8x512^2)", ("segmentation", "resnet18", "fp32", 8, 512)),
                       ("cfg 3 (adversarial bf16)", ("adversarial", "resnet18", "bf16", 8, 512)),
                       ("cfg 5 (r50 bf16 8x768^2)", ("segmentation", "resnet50", "bf16", 8, 768))):
        step, model, trainer = bench.build_leg(*args, 23, dev, 0, 1, False)
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        hs, al = [], []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            hs.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
            al.append((time.perf_counter() - t0) * 1e3)
        hs.sort(); al.sort()
        # steady state: steps back to back without a sync in between
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        t_host = (time.perf_counter() - t0) / 20 * 1e3
        torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{name}: one step from an idle device: host enqueue {hs[5]:.2f} ms, finished {al[5]:.2f} ms | 20 steps back to back: "
              f"host {t_host:.2f} ms per step, device done {t_dev:.2f} ms per step")
        del step, model, trainer
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
