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
