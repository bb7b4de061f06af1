"""N > 1 path on CPU: world_size-2 gloo processes drive the gradient reducer exactly as the backward plan does
(descending "ready" offsets) and check bucket order, averaging, and parameter broadcast.  No kernels involved."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakePlan:
    def __init__(self, garena):
        self.garena = garena


class _FakeNet:
    grad_ready_hook = None


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uda_aerial_semantic_segmentation_research_amd.ddp import GradAllReducer
    total = 1000
    net = _FakeNet()
    red = GradAllReducer(net, bucket_bytes=4 * 256, tail_bytes=0)
    g = torch.arange(total, dtype=torch.float32) * (rank + 1)
    P = _FakePlan(g)
    for off in (900, 640, 512, 100, 0):         # the plan reports finished offsets back-to-front
        net.grad_ready_hook(P, off)
    n_before_finish = len(red.launched)
    red.finish()
    want = torch.arange(total, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok = torch.allclose(g, want) and red.launched == [(744, 1000), (488, 744), (232, 488), (0, 232)] and n_before_finish == 4
    # second backward reuses the reducer
    g2 = torch.full((total,), float(rank))
    P2 = _FakePlan(g2)
    net.grad_ready_hook(P2, 800)
    red.finish()
    ok = ok and torch.allclose(g2, torch.full((total,), (world - 1) / 2))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_reducer_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29517, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_bucket_ranges_cover_arena_back_to_front():
    from uda_aerial_semantic_segmentation_research_amd.ddp import bucket_ranges
    r = bucket_ranges(14335040, (32 << 20) // 4)
    assert r[0][1] == 14335040 and r[-1][0] == 0
    assert all(a2 == b1 for (a1, b1), (a2, b2) in zip(r[1:], r[:-1])) or all(r[i][0] == r[i + 1][1] for i in range(len(r) - 1))
    assert sum(b - a for a, b in r) == 14335040


def test_bucket_ranges_short_tail_for_the_last_gradients():
    from uda_aerial_semantic_segmentation_research_amd.ddp import bucket_ranges
    total, big, tail = 14335040, (32 << 20) // 4, (4 << 20) // 4
    r = bucket_ranges(total, big, tail)
    assert r[-1] == (0, tail) and r[0][1] == total
    assert all(r[i][0] == r[i + 1][1] for i in range(len(r) - 1)) and sum(b - a for a, b in r) == total
    assert all(b - a <= big for a, b in r)
    assert bucket_ranges(100, 64, 1000) == [(0, 100)] and bucket_ranges(100, 64, 0) == [(36, 100), (0, 36)]


# ---- world 8 with the REAL network's arena, offsets and bucket plan: what the driver's 8-GPU run does, on gloo
def _worker8(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uda_aerial_semantic_segmentation_research_amd.ddp import GradAllReducer
    from uda_aerial_semantic_segmentation_research_amd.engine import ConvP
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    net = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23)        # CPU: the arena layout only, no kernel runs
    total = net._arena.numel()
    idx = net.entry_index()
    # the offsets Unet._backward_plan reports, in its order: head, decoder blocks last to first, encoder blocks last to first, 0
    offs = [idx[(id(net.segmentation_head[0]), "weight")][0]]
    blocks = [b for st in net.encoder.stages() for b in st] + list(net.decoder.blocks)
    for blk in reversed(blocks):
        first = next(m for m in blk.modules() if isinstance(m, ConvP))
        offs.append(idx[(id(first), "weight")][0])
    offs.append(0)
    red = GradAllReducer(net, bucket_bytes=32 << 20)                               # bench.py's plan: 32 MiB buckets, 4 MiB tail

    class P:
        pass
    plan = P()
    plan.garena = torch.full((total,), float(rank + 1))
    launched_at = []
    for o in offs:
        net.grad_ready_hook(plan, o)
        launched_at.append(len(red.launched))
    red.finish()
    mib = [round((b - a) * 4 / 2 ** 20) for a, b in red.launched]
    ok = torch.all(plan.garena == (world + 1) / 2).item()                          # mean of 1..world, every element, exactly
    ok = ok and red.launched[0][1] == total and red.launched[-1] == (0, (4 << 20) // 4) and mib == [32, 19, 4]
    ok = ok and all(red.launched[i][0] == red.launched[i + 1][1] for i in range(len(red.launched) - 1))
    ok = ok and offs == sorted(offs, reverse=True) and launched_at[-1] == 3
    # the 32 MiB bucket leaves while the encoder's backward still runs (before the report of the first encoder stage)
    first_enc = offs.index(idx[(id(next(m for m in blocks[0].modules() if isinstance(m, ConvP))), "weight")][0])
    ok = ok and launched_at[first_enc - 1] >= 1
    ret[rank] = (bool(ok), mib, total)
    dist.destroy_process_group()


def test_reducer_world8_gloo_real_arena_and_bucket_plan():
    """Eight gloo ranks, the r18-Unet's real 14.3 M-element gradient arena, the offsets its backward plan reports and the 32 / 19 /
    4 MiB bucket plan of bench.py --gpus 8: buckets tile the arena back to front, leave as soon as complete, every element ends at
    the mean over ranks."""
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(world, 29531, ret), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world)), dict(ret)
    assert ret[0][2] == 14335040
