"""Stand-alone time of the five per-step weight-packing launches of the r18 Unet (fp32): python tools/pack_time.py"""
import sys, torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uda_aerial_semantic_segmentation_research_amd import kernels as K
from uda_aerial_semantic_segmentation_research_amd.unet import Unet
net = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23).to("cuda").train()
net.ensure_arena()
def t(f, it=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
print("pack_frag fwd %.1f us" % t(lambda: K.pack_frag_batched(net._arena, None, net._frag_arena, net._frag_fwd_table)))
print("pack_up   fwd %.1f us" % t(lambda: K.pack_up_batched(net._arena, None, net._frag_arena, net._up_fwd_table)))
print("pack_dgrad    %.1f us" % t(lambda: K.pack_dgrad_batched(net._arena, net._wt_arena, net._wt_table)))
print("pack_frag bwd %.1f us" % t(lambda: K.pack_frag_batched(None, net._wt_arena, net._frag_arena, net._frag_bwd_table)))
print("pack_up   bwd %.1f us" % t(lambda: K.pack_up_batched(None, net._wt_arena, net._frag_arena, net._up_bwd_table)))
