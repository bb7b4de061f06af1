"""Single-node data parallelism for the arena-backed networks: one process per GPU, RCCL all-reduce of the flat gradient
arena over xGMI, overlapped with the rest of backward on a side HIP stream.

The reference is single-process: this layer is new.  Image batches are sharded across ranks, BatchNorm
statistics stay local (per-GPU batch 8 reproduces the single-GPU semantics; torch-DDP ``broadcast_buffers=False``
equivalent), gradients are averaged.  Because a network's gradients are ONE contiguous arena filled back-to-front by
the backward plan, a bucket is just an arena slice: the plan reports "everything at offsets >= o is final" after each
block and the reducer launches the buckets that became complete -- large (default 32 MiB) collectives, plus one short
(4 MiB) slice for the gradients that finish last, sized for the
per-link-bound xGMI ring rather than many small ones.
"""
import torch
import torch.distributed as dist


def bucket_ranges(total, bucket_elems, tail_elems=0):
    """Arena slices [a, b) walking from the END of the arena (first gradients to be ready) to the front.

    ``tail_elems`` > 0 makes the front-most slice at most that long: it holds the gradients that finish last (stem and the
    first encoder stages), its all-reduce cannot overlap with any backward work, so it is kept short; everything behind it
    goes out in large slices while the expensive high-resolution encoder layers are still computing."""
    out, b = [], total
    tail = min(max(tail_elems, 0), total)
    while b > tail:
        a = max(tail, b - bucket_elems)
        out.append((a, b))
        b = a
    if tail > 0:
        out.append((0, tail))
    return out


FORCE = False   # tests / 1-GPU rehearsal: issue the collectives even when world == 1
_COMM_STREAMS = {}


def average_(flat, world, group=None):
    """In-place mean over ranks (AVG where the backend has it, else SUM + scale: gloo).

    Enqueued with ``async_op=False``: with NCCL/RCCL that does NOT block the host -- it makes the CURRENT stream wait for
    the collective (which runs on the backend's own stream), so ordering a consumer after the current stream is enough.
    (With ``async_op=True`` and no ``work.wait()`` the calling stream would not be ordered after the collective.)"""
    if world == 1 and not (FORCE and dist.is_initialized()):
        return
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)


class GradAllReducer:
    """Attach to a network (``Unet``): ``net.grad_ready_hook`` drives bucket launches during backward;
    ``finish()`` (called by the trainer before ``optimizer.step()``) makes the compute stream wait for them."""

    def __init__(self, net, bucket_bytes=32 << 20, group=None, tail_bytes=4 << 20):
        self.net = net
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.tail_elems = max(0, tail_bytes // 4)
        self.comm_stream = None
        self._plan = None
        self._pending = []
        self._garena = None
        self.launched = []        # (a, b) of the last backward, for tests
        net.grad_ready_hook = self._on_ready

    def _begin(self, P):
        self._plan = P
        self._garena = P.garena
        self._pending = bucket_ranges(P.garena.numel(), self.bucket_elems, self.tail_elems)
        self.launched = []
        if P.garena.is_cuda and self.comm_stream is None:
            dev = P.garena.device          # one all-reduce stream per device (see engine._SIDE_STREAMS for why not one per reducer)
            self.comm_stream = _COMM_STREAMS.get(dev) or _COMM_STREAMS.setdefault(dev, torch.cuda.Stream(device=dev))

    def _launch(self, a, b):
        g = self._garena[a:b]
        self.launched.append((a, b))
        if self.world == 1 and not FORCE:
            return
        if g.is_cuda:
            # gradients come from the backward's main stream AND its weight-gradient side stream: wait on both, without
            # stalling either of them
            if hasattr(self._plan, "ready_events"):
                evs = self._plan.ready_events()
            else:
                evs = [torch.cuda.Event()]
                evs[0].record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                for ev in evs:
                    self.comm_stream.wait_event(ev)
                average_(g, self.world, self.group)
        else:
            average_(g, self.world, self.group)

    def _on_ready(self, P, offset):
        if P is not self._plan:
            self._begin(P)
        while self._pending and self._pending[0][0] >= offset:
            self._launch(*self._pending.pop(0))

    def finish(self):
        """All buckets out, compute stream ordered after the collectives."""
        while self._pending:
            self._launch(*self._pending.pop(0))
        if self.comm_stream is not None and (self.world > 1 or FORCE):
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._plan = None

    @staticmethod
    def allreduce_now(net, group=None):
        """Average a network's whole gradient arena on the current stream (small networks: the discriminator)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        ga = getattr(net, "_grad_arena", None)
        if (world > 1 or FORCE) and ga is not None:
            average_(ga, world, group)


def broadcast_parameters(net, src=0, group=None):
    """Rank ``src``'s parameters and BatchNorm buffers to every rank (one collective per arena)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    net.ensure_arena()
    dist.broadcast(net._arena, src, group=group)
    dist.broadcast(net._buf_arena, src, group=group)
    dist.broadcast(net._nbt, src, group=group)
