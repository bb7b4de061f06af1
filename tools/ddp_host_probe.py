#!/usr/bin/env python
"""Host-side cost of the RCCL calls in the data-parallel step, one GPU (world 1, UDASEG_DDP_REHEARSE semantics): how long the host
spends inside each dist.all_reduce enqueue, and whether the host or the device is ahead when Adam is enqueued.

    python tools/ddp_host_probe.py
"""
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist


def main():
    import bench
    from uda_aerial_semantic_segmentation_research_amd import ddp
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    os.environ["UDASEG_DDP_REHEARSE"] = "1"
    step, model, trainer = bench.build_leg("segmentation", "resnet18", "fp32", 8, 512, 23, dev, 0, 1, True)
    calls = []
    orig = ddp.average_

    def timed(flat, world, group=None):
        t0 = time.perf_counter()
        orig(flat, world, group)
        calls.append((flat.numel() * 4 / 2 ** 20, (time.perf_counter() - t0) * 1e6))
    ddp.average_ = timed
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    rows = []
    for _ in range(10):
        calls.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t_host = (time.perf_counter() - t0) * 1e3           # host time to ENQUEUE the whole step
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) * 1e3            # until the device has finished it
        rows.append((t_host, t_all, list(calls)))
    for t_host, t_all, cs in rows[-4:]:
        print(f"host enqueue {t_host:6.2f} ms, step finished {t_all:6.2f} ms (host ahead by {t_all - t_host:5.2f} ms at the end); all_reduce calls: "
              + ", ".join(f"{mb:.0f} MiB {us:.0f} us" for mb, us in cs))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
