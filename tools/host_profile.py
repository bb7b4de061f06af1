#!/usr/bin/env python
"""cProfile of the HOST side of a step (where the ~20 us per launch go):  python tools/host_profile.py [adversarial|segmentation] [bf16|fp32]"""
import cProfile
import os
import pstats
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch


def main():
    import bench
    wl = sys.argv[1] if len(sys.argv) > 1 else "adversarial"
    dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    step, model, trainer = bench.build_leg(wl, "resnet18", dt, 8, 512, 23, dev, 0, 1, False)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
