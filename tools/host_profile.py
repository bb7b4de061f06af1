#!/usr/bin/env python
"""cProfile of the HOST side of a step (where the ~20 us per launch go), the reverse plans included (they run inside the autograd
engine's thread: profiled there).   python tools/host_profile.py [adversarial|segmentation] [bf16|fp32]"""
import cProfile
import os
import pstats
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch


def main():
    import bench
    from uda_aerial_semantic_segmentation_research_amd import discriminator, unet
    wl = sys.argv[1] if len(sys.argv) > 1 else "adversarial"
    dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    step, model, trainer = bench.build_leg(wl, "resnet18", dt, 8, 512, 23, dev, 0, 1, False)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    bw = cProfile.Profile()
    orig = unet.Unet._backward_plan

    def prof_bwd(self, *a, **k):
        bw.enable()
        try:
            return orig(self, *a, **k)
        finally:
            bw.disable()
    unet.Unet._backward_plan = prof_bwd
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
    print("==== the calling thread (forward, losses, optimizers; run_backward = the reverse plans + autograd)")
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    print("==== Unet._backward_plan (inside the autograd engine)")
    pstats.Stats(bw).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
