"""Train step with the batch handed over from HOST memory each step (what a DataLoader-driven epoch does): fp32 NCHW images
+ int64 masks through pinned buffers, and uint8 images + uint8 masks through data.prepare_batch.  Prints images/s for each."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from uda_aerial_semantic_segmentation_research_amd import _lib, data
    from uda_aerial_semantic_segmentation_research_amd.optim import FusedAdam
    from uda_aerial_semantic_segmentation_research_amd.train import SegmentationTrainer
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    _lib.require_gpu()
    torch.manual_seed(0)
    net = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23)
    tr = SegmentationTrainer(net, torch.device("cuda"))
    net.train()
    opt = FusedAdam(net.parameters(), lr=1e-4)
    n, hw, steps = 8, 512, 20
    x_h = torch.randn(n, 3, hw, hw).pin_memory()
    y_h = torch.randint(0, 23, (n, hw, hw)).pin_memory()
    xu_h = torch.randint(0, 256, (n, hw, hw, 3), dtype=torch.uint8).pin_memory()
    yu_h = torch.randint(0, 23, (n, hw, hw), dtype=torch.uint8).pin_memory()
    x_d, y_d = x_h.cuda(), y_h.cuda()

    def resident():
        tr.train_step(x_d, y_d, opt)

    def host_fp32():
        tr.train_step(x_h.cuda(non_blocking=True), y_h.cuda(non_blocking=True), opt)

    def host_u8():
        x, y = data.prepare_batch(xu_h, yu_h, data.random_d4_codes(n))
        tr.train_step(x, y, opt)

    for name, fn in (("inputs resident in HBM", resident), ("fp32 images + int64 masks from pinned host memory", host_fp32),
                     ("uint8 images + masks from pinned host memory via prepare_batch (D4 + normalise on device)", host_u8)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name:92s} {n * steps / dt:8.1f} images/s  {1e3 * dt / steps:7.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
