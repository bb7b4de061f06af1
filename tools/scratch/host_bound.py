import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import bench
dev = torch.device("cuda", 0)
for wl, enc, dt, size in (("segmentation", "resnet18", "fp32", 512), ("segmentation", "resnet18", "bf16", 512),
                          ("adversarial", "resnet18", "bf16", 512), ("segmentation", "resnet50", "bf16", 768)):
    step, model, trainer = bench.build_leg(wl, enc, dt, 8, size, 23, dev, 0, 1, False)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{wl} {enc} {dt} {size}: host enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step", flush=True)
    del step, model, trainer
    torch.cuda.empty_cache()
