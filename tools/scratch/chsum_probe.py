import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from uda_aerial_semantic_segmentation_research_amd import kernels as K
dev = torch.device("cuda", 0)
step, model, trainer = bench.build_leg("adversarial", "resnet18", sys.argv[1] if len(sys.argv) > 1 else "bf16", 8, 512, 23, dev, 0, 1, False)
for _ in range(3): step()
torch.cuda.synchronize()
orig = K.channel_sum
log = []
def cs(x, out, accumulate=False, st=None):
    torch.cuda.synchronize()
    t = time.perf_counter()
    orig(x, out, accumulate, st)
    torch.cuda.synchronize()
    log.append((tuple(x.shape), x.dtype, (time.perf_counter() - t) * 1e6))
K.channel_sum = cs
import uda_aerial_semantic_segmentation_research_amd.engine as E
step()
for l in log: print(l)
