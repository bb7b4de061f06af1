import os, sys, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import importlib.util, torch
spec = importlib.util.spec_from_file_location("bench", os.path.join(sys.path[0], "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
mode = sys.argv[1]
def leg(name, *a):
    r = b.also_leg(name, *a, 23, dev)
    print(mode, name, r["value"], r["ms_per_step"], flush=True)
def quick(workload, enc, dt, size):
    step, model, trainer = b.build_leg(workload, enc, dt, 8, size, 23, dev, 0, 1, False)
    dtm, ev, loss = b.timed_region(step, 10, 6, 1, dev, False)
    print(mode, "quick", workload, enc, dt, size, round(8 * 10 / dtm, 1), flush=True)
    return step, model, trainer
if mode == "cfg5_only":
    leg("cfg5", "segmentation", "resnet50", "bf16", 8, 768)
elif mode == "r18fp32_then_cfg5":
    s = quick("segmentation", "resnet18", "fp32", 512); del s; torch.cuda.empty_cache()
    leg("cfg5", "segmentation", "resnet50", "bf16", 8, 768)
elif mode == "cfg3_then_cfg5":
    leg("cfg3", "adversarial", "resnet18", "bf16", 8, 512)
    leg("cfg5", "segmentation", "resnet50", "bf16", 8, 768)
elif mode == "cfg5_noroof_twice":
    for i in range(2):
        s = quick("segmentation", "resnet50", "bf16", 768); del s; torch.cuda.empty_cache()
elif mode == "cfg5_roof_then_cfg5":
    leg("cfg5a", "segmentation", "resnet50", "bf16", 8, 768)
    leg("cfg5b", "segmentation", "resnet50", "bf16", 8, 768)
