"""Diagnostic: per-parameter gradient error of the HIP path and of the fp32 CPU oracle, both against an fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.unet_ref import UnetRef
from oracle.adversarial_ref import synthetic_batch
from uda_aerial_semantic_segmentation_research_amd.unet import Unet
from uda_aerial_semantic_segmentation_research_amd.losses import CrossEntropyLoss


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


for name, (n, h, w) in [("resnet18", (2, 64, 64)), ("resnet18", (2, 128, 128)), ("resnet18", (4, 256, 256)),
                        ("resnet50", (2, 128, 128))]:
    torch.manual_seed(1234)
    ref = UnetRef(name, classes=23).train()
    ref64 = UnetRef(name, classes=23).double().train()
    ref64.load_state_dict(ref.state_dict())
    net = Unet(name, classes=23)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().train()
    x, y, _ = synthetic_batch(n, h, w, seed=0)
    l32 = torch.nn.functional.cross_entropy(ref(x), y); l32.backward()
    l64 = torch.nn.functional.cross_entropy(ref64(x.double()), y); l64.backward()
    lg = CrossEntropyLoss()(net(x.cuda()), y.cuda()); lg.backward()
    print(f"== {name} {n}x{h}x{w}: loss cpu32 {l32.item():.7f} cpu64 {l64.item():.7f} gpu {lg.item():.7f}")
    g32 = dict(ref.named_parameters()); g64 = dict(ref64.named_parameters())
    rows = []
    for k, p in net.named_parameters():
        rows.append((k, rel(p.grad, g64[k].grad), rel(g32[k].grad, g64[k].grad), rel(p.grad, g32[k].grad)))
    rows.sort(key=lambda r: -r[1])
    print("   worst 8 (gpu-vs-f64, cpu32-vs-f64, gpu-vs-cpu32):")
    for r in rows[:8]:
        print(f"   {r[0]:48s} {r[1]:.2e} {r[2]:.2e} {r[3]:.2e}")
    import statistics
    print("   median gpu-vs-f64 %.2e  median cpu32-vs-f64 %.2e" % (statistics.median(r[1] for r in rows), statistics.median(r[2] for r in rows)))
