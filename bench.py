#!/usr/bin/env python
"""bench.py -- training images/sec at 512x512 on N MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
           bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ResNet-18 encoder + Unet decoder, source-only cross entropy, batch 8 x 3 x 512 x 512
per GPU, fp32, synthetic N(0,1) images / uniform labels, seeded random-init weights.  A "step" is the reference's timed
region src/models/train.py:340-344: zero_grad -> forward -> CrossEntropy -> backward -> (gradient all-reduce) ->
Adam step.  Weak scaling: per-GPU batch fixed; `value` = images processed by all ranks / max-over-ranks wall time of
EXACTLY --steps steps between two barrier + synchronize pairs.

One JSON line on rank 0:
  * `step_ms`     -- HIP-event duration of every timed step (median / p10 / p90), next to the wall-clock `ms_per_step`;
  * `sustained`   -- the same step repeated for ~2.5 s after the timed region (not part of `value`): a 20-step region lasts
                     a quarter of a second, too short for a sampling monitor to see the GPU busy or for the clock to settle;
  * `roofline`    -- dominant conv kernel symbol, HIP-event timed on the launch stream in a separate single-stream leg;
  * `cpu_baseline`-- the CPU oracle timed on the host cores (N=1 only);
  * `also`        -- short legs of BASELINE configs 3 and 5 (bf16 storage: adversarial iteration 8+8 x 512^2; r50 at 768^2),
                     each with its own ms_per_step and dominant-kernel roofline (N=1 only; informational).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see uda_aerial_semantic_segmentation_research_amd/__init__.py
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch
import torch.distributed as dist

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: ~2.5 PF dense bf16 (never the 2:1-sparsity figure)
F32X3_MFMA_PRODUCTS = 6             # bf16 MFMA products per fp32 product in the three-term-split kernels (i + j <= 2 of 3 x 3)
ARITHMETIC_FP32 = ("fp32 tensors, weights, gradients and accumulation.  Stride-1 3x3 convolutions (forward, data and weight "
                   "gradient where the channel counts allow) evaluate each fp32 product on the bf16 matrix pipe from an EXACT "
                   "three-term split of both operands (x = bf16(x) + bf16(x - x0) + bf16(x - x0 - x1); the six products with "
                   "i + j <= 2, fp32 accumulate; what is left out is <= 2^-23 of a product = one fp32 ulp; measured against f64 "
                   "next to the fp32-MFMA kernels in tests/test_gpu_f32x3.py).  The strided, 1x1 and 4x4 layers of more than 32 "
                   "channels take the same split inside the shared implicit-GEMM kernels (round 4); the 7x7 stem's weight gradient, "
                   "<= 32-channel shared-source launches and the small full-resolution weight gradients stay on "
                   "v_mfma_f32_32x32x2_f32 / 16x16x4_f32.  The bf16 MFMA adder truncates toward minus infinity (measured): every "
                   "split kernel runs its K loop + - - + (or alternates accumulators) so that the bias cancels.  "
                   "UDASEG_F32_SPLIT=0 runs everything on the fp32-MFMA kernels (also-leg 'fp32-MFMA kernels only').")
WORKLOAD_TEXT = {
    "segmentation": "source-only CE train step (zero_grad,fwd,CE,bwd,allreduce,Adam)",
    "adversarial": "adversarial iteration (D step on 8 source + 8 target images, then segmenter step: CE + lambda*BCE)",
    "inference": "validation step (eval forward with folded BN, CE, device-side confusion matrix)",
}
# conv FLOPs per SOURCE image, fwd + dgrad + wgrad (no dgrad for the stem / D's first conv)
CONV_GFLOP_PER_IMAGE = {("segmentation", "resnet18", 512): 133.30, ("segmentation", "resnet50", 512): 258.93,
                        ("segmentation", "resnet50", 768): 582.60, ("segmentation", "resnet34", 512): 191.29,
                        ("adversarial", "resnet18", 512): 251.68}


def synthetic(n, h, w, classes, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 3, h, w, generator=g)
    g = torch.Generator().manual_seed(seed + 1)
    y = torch.randint(0, classes, (n, h, w), generator=g, dtype=torch.int64)
    return x.to(device), y.to(device)


def pmc_traffic(kernel, prefer=None, run_symbols=None):
    """(HBM-side bytes per launch of `kernel`, source file) from the newest committed PMC reduction (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py).  The PMC
    passes cannot run inside this script (counter collection needs its own rocprofv3 process): the value is a committed
    measurement of the same kernel symbol, (None, None) when no committed file knows the symbol.
    run_symbols: the convolution symbols THIS run launched -- a file collected on another build (one whose kernel list does not
    hold every symbol of the run that makes up >= 2 % of the conv time) is refused: its numbers describe other kernels."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True)
    if prefer:      # the same symbol runs other shapes in another configuration: take the file collected on THIS workload first
        files.sort(key=lambda f: prefer not in os.path.basename(f))
    for f in files:
        try:
            ks = json.load(open(f))["kernels"]
        except Exception:
            continue
        k = ks.get(kernel)
        if not k:
            continue
        if run_symbols and any(sym not in ks for sym in run_symbols):
            continue
        return k["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
    return None, None


def host_cores():
    """The cores this process may actually use: os.cpu_count() clipped by the affinity mask and by the cgroup's CPU quota (the
    GPU box is a slice of a large host: more threads than granted cores only adds contention)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _cpulist(txt):
    out = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def pin_to_gpu_numa_node(device_index, local_rank, local_world):
    """N > 1: each rank's launcher thread (Python + ctypes + hipLaunchKernel: ~4 ms of host time per 8 ms step) runs on cores of ITS
    GPU's NUMA node -- the node's allowed cores are dealt out among the ranks whose GPUs sit on that node, so eight launchers do not
    pile onto the same cores or launch across the socket link.  The PCI address comes from torch.cuda.get_device_properties (no
    context is created); without NUMA information the allowed cores are split evenly among the local ranks.  Returns a dict for the
    JSON line ({"cores": n, "numa_node": k | None}) or None when pinning is not possible (no sched_setaffinity)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    node, cores = None, None
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node >= 0:
            on_node = [c for c in _cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) if c in allowed]
            # ranks sharing this node: those whose device has the same numa_node
            peers = []
            for r in range(local_world):
                try:
                    q = torch.cuda.get_device_properties(r)
                    b2 = f"{q.pci_domain_id:04x}:{q.pci_bus_id:02x}:{q.pci_device_id:02x}.0"
                    if int(open(f"/sys/bus/pci/devices/{b2}/numa_node").read()) == node:
                        peers.append(r)
                except Exception:
                    peers.append(r)
            if on_node and local_rank in peers:
                k, m = peers.index(local_rank), len(peers)
                per = max(1, len(on_node) // m)
                cores = on_node[k * per:(k + 1) * per] or on_node
        else:
            node = None
    except Exception:
        node = None
    if not cores:
        per = max(1, len(allowed) // max(local_world, 1))
        cores = allowed[local_rank * per:(local_rank + 1) * per] or allowed
    try:
        os.sched_setaffinity(0, cores)
        torch.set_num_threads(max(1, min(len(cores), 4)))
    except OSError:
        return None
    return {"cores": len(cores), "numa_node": node}


def cpu_baseline(encoder, classes, hw, workload="segmentation", batch=8, warmups=2, timed=5, budget_s=30.0):
    """The CPU oracle (pure-torch restatement of the reference path) on this box's host cores, as BASELINE.md section 3 plans
    it: the SAME batch as the GPU leg, fp32, torch.set_num_threads(os.cpu_count()), 2 warm-ups, median of >= 5 timed steps
    (time.perf_counter).  The sample is bounded: if the warm-ups show that the plan would exceed `budget_s`, fewer steps are
    timed (never fewer than 3) and `sample` says so."""
    from oracle.unet_ref import UnetRef
    from oracle.adversarial_ref import (AdversarialLossRef, DomainDiscriminatorRef, adversarial_step, segmentation_step,
                                        synthetic_batch)
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    model = UnetRef(encoder, classes=classes).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    x, y, xt = synthetic_batch(batch, hw, hw, classes=classes, seed=0)
    if workload == "adversarial":
        D = DomainDiscriminatorRef(3).train()
        d_opt = torch.optim.Adam(D.parameters(), lr=1e-4)
        adv = AdversarialLossRef(0.001)

        def one():
            adversarial_step(model, D, adv, opt, d_opt, x, y, xt)
        what = "oracle/adversarial_ref.py adversarial_step (D step + segmenter step)"
    else:
        def one():
            segmentation_step(model, opt, x, y)
        what = "oracle/adversarial_ref.py segmentation_step"
    t0 = time.perf_counter()
    for _ in range(warmups):
        one()                                      # warm-up (oneDNN primitive creation, allocator)
    warm = (time.perf_counter() - t0) / max(warmups, 1)
    n_t = timed
    if warm * (warmups + timed) > budget_s:
        n_t = max(3, int(budget_s / warm) - warmups)
    ts = []
    for _ in range(n_t):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    med = percentile(ts, 0.5)
    return {"value": round(batch / med, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"median of {n_t} steps of batch {batch}x3x{hw}x{hw} fp32 after {warmups} warm-ups ({med:.2f} s per step), "
                      f"torch {torch.__version__} CPU with {cores} threads, {what} on UnetRef({encoder}) + torch.optim.Adam"}


def build_leg(workload, encoder, dtype, batch, size, classes, dev, rank, world, rehearse):
    """Model + trainer + synthetic batch of one benchmark leg -> (step callable, model, trainer)."""
    from uda_aerial_semantic_segmentation_research_amd.ddp import GradAllReducer, broadcast_parameters
    from uda_aerial_semantic_segmentation_research_amd.optim import FusedAdam
    from uda_aerial_semantic_segmentation_research_amd.train import SegmentationTrainer
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    torch.manual_seed(1234)
    model = Unet(encoder_name=encoder, encoder_weights=None, in_channels=3, classes=classes,
                 compute_dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)
    if workload == "adversarial":
        from uda_aerial_semantic_segmentation_research_amd.adversarial_trainer import AdversarialTrainer
        trainer = AdversarialTrainer(model, dev, lambda_adv=0.001)
        trainer.discriminator.train()
    else:
        trainer = SegmentationTrainer(model, dev)
    model.train()
    model.ensure_arena()
    if world > 1 or rehearse:
        from uda_aerial_semantic_segmentation_research_amd import ddp as _ddp
        _ddp.FORCE = os.environ.get("UDASEG_DDP_REHEARSE") == "1"       # "2": hooks and streams only, no collectives
        if world > 1:
            broadcast_parameters(model)
        else:
            dist.broadcast(model._arena, 0)
        if os.environ.get("UDASEG_DDP_REHEARSE") != "3":               # "3": process group only
            trainer.grad_reducer = GradAllReducer(model, bucket_bytes=int(os.environ.get("UDASEG_DDP_BUCKET_MB", "32")) << 20)
            if workload == "adversarial":
                broadcast_parameters(trainer.discriminator)
                trainer.d_grad_reducer = GradAllReducer          # its allreduce_now(): one collective over D's gradient arena
    opt = FusedAdam(model.parameters(), lr=1e-4)
    x, y = synthetic(batch, size, size, classes, seed=100 * rank, device=dev)
    if workload == "inference":
        # validate()'s inner loop (train.py:398-408): eval forward + CE + metrics, BatchNorm folded into the convs
        model.eval()
        from uda_aerial_semantic_segmentation_research_amd.metrics import confusion_matrix

        def step():
            with torch.no_grad():
                out = model(x)
                loss = trainer.criterion(out, y)
                confusion_matrix(out, y, classes)
            return loss
    elif workload == "adversarial":
        xt, _ = synthetic(batch, size, size, classes, seed=100 * rank + 2, device=dev)
        trainer.discriminator_optimizer = FusedAdam(trainer.discriminator.parameters(), lr=1e-4)

        def step():
            return trainer.adversarial_step(x, y, xt, opt, update_metrics=False)[3]
    else:
        def step():
            return trainer.train_step(x, y, opt)[0]
    return step, model, trainer


# every convolution kernel symbol of the library, classed EXPLICITLY by the matrix pipe it runs on (a heuristic on the
# template-argument count mispriced launches silently whenever a kernel's template list changed; an unknown symbol now raises)
SPLIT_PREFIXES = ("conv3x3_f32x3_kernel<", "conv3x3_f32x3_ws_kernel<", "conv_wgrad_h2_kernel<3,", "conv_wgrad_x3_kernel<",
                  "conv_up_fwd_f32x3_kernel<", "conv_up_dgrad_f32x3_kernel<", "conv_wgrad_up_kernel<", "conv3x3_n16_f32x3_kernel<", "conv_stem_f32x3_kernel")
NATIVE_PREFIXES = ("conv_wgrad_kernel<", "conv3x3_small_", "conv_wgrad_bf16_kernel<", "conv_wgrad_h2_kernel<1,", "conv_halo_bf16_kernel<",
                   "conv1x1_stream_bf16_kernel<", "conv1x1_gemm_bf16_kernel", "conv_halo_s2", "conv_small", "conv2d_folded")
# conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, BF16, UNI, UP[, X3]>: the three-term instantiations carry an EIGHTH argument "true"
IGEMM_TEMPLATE_ARGS = (7, 8)


def pipe_of(symbol, dtype):
    """(bf16-MFMA products per counted product, peak TFLOP/s of the pipe the kernel symbol runs on).  The fp32 three-term-split
    kernels evaluate every fp32 product as SIX bf16 MFMA products: the pipe that bounds them is the bf16 one and the work it does
    is 6 x the algorithmic FLOPs."""
    native = (1, BF16_MFMA_PEAK_TFLOPS if dtype == "bf16" else FP32_MFMA_PEAK_TFLOPS)
    if symbol.startswith("conv_igemm_kernel<"):
        args = [a.strip() for a in symbol[symbol.index("<") + 1:symbol.rindex(">")].split(",")]
        if args[0] == "*":                  # legacy lump of the bf16 instantiations (fixed table, unused since round 2)
            return native
        if len(args) not in IGEMM_TEMPLATE_ARGS or any(a not in ("true", "false") for a in args[4:]):
            raise RuntimeError(f"bench.py: conv_igemm_kernel's template list changed ({symbol}): update IGEMM_TEMPLATE_ARGS / pipe_of")
        return (F32X3_MFMA_PRODUCTS, BF16_MFMA_PEAK_TFLOPS) if (len(args) == 8 and args[7] == "true") else native
    if symbol.startswith(SPLIT_PREFIXES):
        return F32X3_MFMA_PRODUCTS, BF16_MFMA_PEAK_TFLOPS
    if symbol.startswith(NATIVE_PREFIXES):
        return native
    raise RuntimeError(f"bench.py: convolution kernel symbol {symbol!r} is not classed by matrix pipe (SPLIT_PREFIXES / NATIVE_PREFIXES)")


def percentile(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, max(0, int(round(q * (len(v) - 1)))))]


def timed_region(step, steps, warmup, world, dev, rehearse):
    """W untimed steps, then EXACTLY `steps` steps between barrier + synchronize pairs.  Returns (max-over-ranks wall
    seconds, per-step HIP-event milliseconds, last loss)."""
    for _ in range(warmup):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        loss = step()
        b.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1 or rehearse:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, [a.elapsed_time(b) for a, b in evs], loss


def roofline_leg(step, model, trainer, dtype, psteps=3, layer_table=False, pmc_tag=None):
    """Separate leg: HIP events around every conv-kernel launch on the launch stream (C-ABI udaseg_prof_*), single stream --
    with the weight gradients on their side stream concurrent kernels stretch each other's durations and a per-kernel rate
    would under-state the kernel (the timed region keeps the overlap).  No collective may be issued here."""
    from uda_aerial_semantic_segmentation_research_amd import engine as _engine
    from uda_aerial_semantic_segmentation_research_amd import kernels as K
    was_side = _engine.SIDE_STREAM_WGRAD
    _engine.SIDE_STREAM_WGRAD = False
    trainer.grad_reducer = None
    model.grad_ready_hook = None
    if hasattr(trainer, "d_grad_reducer"):
        trainer.d_grad_reducer = None
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    # single-stream step time without any event in the stream (for the "other" share of the split below) -- taken BEFORE the
    # profiled steps: releasing their thousands of events afterwards stalled the next launches now and then (45 instead of 11 ms)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(psteps):
        step()
    e1.record()
    torch.cuda.synchronize()
    serial_ms = e0.elapsed_time(e1) / psteps
    K.prof_reset()
    K.prof_enable(True)
    for _ in range(psteps):
        step()
    torch.cuda.synchronize()
    K.prof_enable(False)
    _engine.SIDE_STREAM_WGRAD = was_side
    # HBM-side view of the same launches: algorithmic bytes (gathered tensor + output + weights, each once) over the summed
    # launch time -- at bf16 rates most of these convolutions are bound by HBM, not by the matrix cores
    es = 2 if dtype == "bf16" else 4
    conv_bytes, conv_ms_api = 0.0, 0.0
    agg = {}
    for fam in (0, 1):
        for ms, fl, kind, d in K.prof_records(fam):
            n_, hi_, wi_, ci_, ho_, wo_, co_, kh_, kw_ = d[:9]
            x_b, y_b, w_b = n_ * hi_ * wi_ * ci_ * es, n_ * ho_ * wo_ * co_ * es, co_ * kh_ * kw_ * ci_ * (4 if kind == 2 else es)
            conv_bytes += x_b + y_b + w_b
            conv_ms_api += ms
            a = agg.setdefault((kind, d), [0.0, 0.0, 0])
            a[0] += ms
            a[1] += fl
            a[2] += 1
    if layer_table:
        print("kind  n  hi  wi   ci   co k s |  calls/step  ms/call   GFLOP   TFLOP/s   alg MB   TB/s   ms/step", file=sys.stderr)
        for (kind, d), (ms, fl, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            by = d[0] * d[1] * d[2] * d[3] * es + d[0] * d[4] * d[5] * d[6] * es + d[6] * d[7] * d[8] * d[3] * (4 if kind == 2 else es)
            print(f"{('fwd', 'dgrad', 'wgrad')[kind]:5s} {d[0]:2d} {d[1]:3d} {d[2]:3d} {d[3]:4d} {d[6]:4d} {d[7]} {d[9]} | "
                  f"{cnt / psteps:5.1f} {ms / cnt:9.4f} {fl / cnt / 1e9:8.2f} {fl / ms / 1e9:8.1f} {by / 1e6:8.1f} "
                  f"{by * cnt / ms / 1e9:6.2f} {ms / psteps:8.3f}", file=sys.stderr)
    allk = [k for k in K.prof_kernels() if k[3] > 0]
    K.prof_reset()
    kern = [k for k in allk if k[0].startswith("conv")]            # MFMA kernels: k[2] = FLOPs
    hbm_k = [k for k in allk if not k[0].startswith("conv")]       # bandwidth kernels (BatchNorm passes): k[2] = algorithmic bytes
    # BASELINE cfg 5 asks for the split "HBM-bound vs MFMA-bound": every conv call is classed by its arithmetic intensity
    # (FLOPs / algorithmic bytes) against the machine balance peak / 6.3 TB/s (measured copy rate, MI355X_MICROARCH.md)
    balance = (BF16_MFMA_PEAK_TFLOPS if dtype == "bf16" else FP32_MFMA_PEAK_TFLOPS) * 1e12 / 6.3e12
    split = {"mfma": [0.0, 0.0, 0.0, 0], "hbm": [0.0, 0.0, 0.0, 0]}
    for (kind, d), (ms, fl, cnt) in agg.items():
        n_, hi_, wi_, ci_, ho_, wo_, co_, kh_, kw_ = d[:9]
        by = cnt * (n_ * hi_ * wi_ * ci_ * es + n_ * ho_ * wo_ * co_ * es + co_ * kh_ * kw_ * ci_ * (4 if kind == 2 else es))
        cls = split["mfma" if fl / max(by, 1) >= balance else "hbm"]
        cls[0] += ms
        cls[1] += fl
        cls[2] += by
        cls[3] += cnt
    api_ms = max(split["mfma"][0] + split["hbm"][0], 1e-9)
    kern_ms = sum(k[1] for k in kern)                               # kernel-level events: free of the API-level event overhead
    bn_ms, bn_bytes = sum(k[1] for k in hbm_k), sum(k[2] for k in hbm_k)
    time_split = {
        "serial_step_ms": round(serial_ms, 3),
        "note": "single-stream leg; conv calls classed by FLOPs / algorithmic bytes against the machine balance "
                f"({balance:.0f} FLOP/B = dense MFMA peak / 6.3 TB/s); class times are the API-level event times scaled to the "
                "kernel-level conv total (the API-level pair carries ~8 us of event overhead per call)",
        "conv_mfma_bound": {"ms": round(split["mfma"][0] / api_ms * kern_ms / psteps, 3), "calls": split["mfma"][3] // psteps,
                            "tflops": round(split["mfma"][1] / max(split["mfma"][0] / api_ms * kern_ms, 1e-9) / 1e9, 1)},
        "conv_hbm_bound": {"ms": round(split["hbm"][0] / api_ms * kern_ms / psteps, 3), "calls": split["hbm"][3] // psteps,
                           "TB_per_s": round(split["hbm"][2] / max(split["hbm"][0] / api_ms * kern_ms, 1e-9) / 1e9, 3),
                           "tflops": round(split["hbm"][1] / max(split["hbm"][0] / api_ms * kern_ms, 1e-9) / 1e9, 1)},
        "batchnorm_passes": {"ms": round(bn_ms / psteps, 3), "TB_per_s": round(bn_bytes / max(bn_ms, 1e-9) / 1e9, 3),
                             "by_kernel": {k[0]: {"ms": round(k[1] / psteps, 3), "TB_per_s": round(k[2] / max(k[1], 1e-9) / 1e9, 3),
                                                  "launches": k[3] // psteps} for k in hbm_k}},
        "other_ms": round(serial_ms - (kern_ms + bn_ms) / psteps, 3)}
    kern.sort(key=lambda k: -k[1])
    dom = kern[0]                                   # the kernel symbol with the most device time
    algorithmic = dom[2] / (dom[1] * 1e-3) / 1e12
    conv_ms = sum(k[1] for k in kern) / psteps
    peak = BF16_MFMA_PEAK_TFLOPS if dtype == "bf16" else FP32_MFMA_PEAK_TFLOPS
    mult, peak = pipe_of(dom[0], dtype)
    split = mult > 1
    # the three-term-split kernels evaluate every fp32 product as SIX bf16 MFMA products (csrc/conv_halo_f32x3.hip): the pipe that
    # bounds them is the bf16 one and the work it does is 6 x the algorithmic FLOPs -- priced against the dense bf16 peak
    achieved = algorithmic * mult
    # pipe-relative utilisation of the whole single-stream conv time: every kernel's pipe work against ITS pipe's dense peak
    pipe_ms = sum(k[2] * pipe_of(k[0], dtype)[0] / (pipe_of(k[0], dtype)[1] * 1e12) * 1e3 for k in kern) / psteps
    conv_total = max(sum(k[1] for k in kern), 1e-9)
    traffic, traffic_src = pmc_traffic(dom[0], pmc_tag, run_symbols=[k[0] for k in kern if k[1] >= 0.02 * conv_total])
    return {"bound": "mfma", "kernel": dom[0], "achieved": round(achieved, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "algorithmic_tflops": round(algorithmic, 2),
            "pipe": ("bf16 MFMA, 6 products per fp32 product (exact three-term operand split, fp32 accumulation): achieved = 6 x "
                     "algorithmic FLOPs / launch time, peak = dense bf16") if split else
                    ("bf16 MFMA" if dtype == "bf16" else "fp32 MFMA"),
            "traffic": traffic,
            "traffic_source": (traffic_src + " (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this symbol; "
                               "counters cannot be collected from inside the timed process)") if traffic_src else None,
            "launches_per_step": dom[3] // psteps, "avg_launch_us": round(1e3 * dom[1] / dom[3], 2),
            "gflop_per_launch": round(dom[2] / dom[3] / 1e9, 3),
            "ms_per_step": round(dom[1] / psteps, 3),
            "hbm_view": {"algorithmic_GB_per_step": round(conv_bytes / psteps / 1e9, 3),
                         "achieved_TB_per_s": round(conv_bytes / (conv_ms_api * 1e-3) / 1e12, 3) if conv_ms_api else None,
                         "frac_of_8TBps": round(conv_bytes / (conv_ms_api * 1e-3) / 8e12, 4) if conv_ms_api else None,
                         "note": "all conv launches: (gathered tensor + output + weights, each counted once) / summed launch time"},
            "time_split": time_split,
            "pipe_ms_per_step": round(pipe_ms, 4),
            "all_conv_kernels": {"ms_per_step": round(conv_ms, 3),
                                 "achieved": round(sum(k[2] for k in kern) / psteps / (conv_ms * 1e-3) / 1e12, 2),
                                 "flops_note": "per-kernel GEMM FLOPs count PHYSICAL channels (image 3->4, logits 23->24): the "
                                               "stem and head rows are overstated by 33 % / 4 %; the headline "
                                               "conv_tflops_per_gpu uses the logical 133.30 GFLOP per image",
                                 "by_kernel": {k[0]: {"ms_per_step": round(k[1] / psteps, 3),
                                                      "tflops": round(k[2] / (k[1] * 1e-3) / 1e12, 1),
                                                      **({"bf16_pipe_tflops": round(F32X3_MFMA_PRODUCTS * k[2] / (k[1] * 1e-3) / 1e12, 1)}
                                                         if pipe_of(k[0], dtype)[0] > 1 else {}),
                                                      "launches_per_step": k[3] // psteps} for k in kern}}}


def also_leg(name, workload, encoder, dtype, batch, size, classes, dev, steps=10, warmup=6, cpu=False, split=True):
    """A short informational leg of another BASELINE config in the same process (N=1 only).  split=False: the fp32 network
    built without the three-term-split kernels (what UDASEG_F32_SPLIT=0 gives)."""
    from uda_aerial_semantic_segmentation_research_amd import engine as _engine, kernels as _K
    was_split = _engine.USE_F32_SPLIT
    _engine.USE_F32_SPLIT = was_split and split
    if not split:
        _K.set_f32_split(0)            # the shared-source kernels' own three-term mode (conv_igemm X3, conv_wgrad_x3) too
    try:
        step, model, trainer = build_leg(workload, encoder, dtype, batch, size, classes, dev, 0, 1, False)
        dt, ev_ms, loss = timed_region(step, steps, warmup, 1, dev, False)
        roof = roofline_leg(step, model, trainer, dtype, psteps=2,
                            pmc_tag="cfg3" if workload == "adversarial" else ("cfg5" if dtype == "bf16" else "fp32"))
    finally:
        _engine.USE_F32_SPLIT = was_split
        if not split:
            _K.set_f32_split(-1)
    value = batch * steps / dt
    gf = CONV_GFLOP_PER_IMAGE.get((workload, encoder, size))
    peak = BF16_MFMA_PEAK_TFLOPS if dtype == "bf16" else FP32_MFMA_PEAK_TFLOPS
    out = {"config": name, "workload": f"{encoder}-Unet {WORKLOAD_TEXT[workload]}, batch {batch}x3x{size}x{size}, {dtype}",
           "value": round(value, 2), "unit": "images/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * dt / steps, 3), "step_ms_median": round(percentile(ev_ms, 0.5), 3),
           "final_loss": round(float(loss.item()), 5),
           "conv_mfma_util": round(value * gf / 1e3 / peak, 4) if gf else None,
           "conv_pipe_util": round(roof["pipe_ms_per_step"] / (1e3 * dt / steps), 4),
           "roofline": {k: roof[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                             "launches_per_step", "avg_launch_us", "ms_per_step", "hbm_view")},
           "time_split": roof["time_split"],
           "conv_kernels": roof["all_conv_kernels"]["by_kernel"],
           "all_conv_kernels_ms_per_step": roof["all_conv_kernels"]["ms_per_step"]}
    if cpu:
        out["cpu_baseline"] = cpu_baseline(encoder, classes, size, workload=workload, batch=batch, warmups=1, timed=3,
                                           budget_s=25.0)
    del step, model, trainer
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--encoder", default="resnet18")
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--classes", type=int, default=23)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the short cfg 3 / cfg 5 legs")
    ap.add_argument("--no-sustain", action="store_true", help="skip the ~2.5 s sustained leg after the timed region")
    ap.add_argument("--layer-table", action="store_true", help="print TFLOP/s per conv shape (stderr)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="fp32 = BASELINE configs[1] (headline); bf16 = bf16 storage / MFMA with fp32 master (configs 3, 5), "
                         "informational")
    ap.add_argument("--workload", default="segmentation", choices=["segmentation", "adversarial", "inference"],
                    help="segmentation = BASELINE configs[1] (headline); adversarial = configs[2]'s iteration "
                         "(adversarial_trainer.py:85-114) in fp32, reported for information")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` by itself: start the N ranks as a CHILD process before anything here touches the GPU
        # (plain subprocess, never an exec), relay its output and exit with its code.
        import subprocess
        port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with python -m torch.distributed.run "
                         f"--nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ... (or plain `python bench.py --gpus N`)")
    # UDASEG_BENCH_SHARE_GPU=1 + UDASEG_BENCH_BACKEND=gloo: rehearsal of the N>1 control flow on a one-GPU box (all ranks on
    # device 0, collectives through gloo); the real run is one rank per GPU over RCCL
    share = os.environ.get("UDASEG_BENCH_SHARE_GPU", "0") == "1"
    backend = os.environ.get("UDASEG_BENCH_BACKEND", "nccl")
    local = 0 if share else local
    pinned = pin_to_gpu_numa_node(local, int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) \
        if world > 1 else None
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rehearse = os.environ.get("UDASEG_DDP_REHEARSE", "0") in ("1", "2", "3")   # run the N>1 code path (NCCL, side stream) at world 1
    if world > 1 or rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    headline = (args.workload == "segmentation" and args.dtype == "fp32" and args.size == 512 and args.encoder == "resnet18"
                and args.batch == 8)
    step, model, trainer = build_leg(args.workload, args.encoder, args.dtype, args.batch, args.size, args.classes, dev, rank,
                                     world, rehearse)
    dt, ev_ms, loss = timed_region(step, args.steps, args.warmup, world, dev, rehearse)
    final_loss = float(loss.item())

    sustained = None
    if not args.no_sustain and world == 1:
        # the timed region is a fraction of a second: keep stepping (outside `value`) so that a sampling monitor sees the GPU
        # busy and the rate is also known with the clock settled
        per = dt / args.steps
        n_sus = max(10, min(2000, int(2.5 / max(per, 1e-4))))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        torch.cuda.synchronize()
        ds = time.perf_counter() - t0
        sustained = {"steps": n_sus, "seconds": round(ds, 3), "ms_per_step": round(1e3 * ds / n_sus, 3),
                     "images_per_s": round(args.batch * n_sus / ds, 2)}

    roofline = None
    store = dist.distributed_c10d._get_default_store() if (world > 1 and dist.is_initialized()) else None
    if not args.no_roofline and rank == 0:
        # only rank 0 runs this leg.  The other ranks wait on the HOST (a key of the rendezvous store), not inside a collective: an
        # NCCL barrier would park a spinning kernel on seven GPUs for the seconds this takes
        tag = "cfg3" if args.workload == "adversarial" else ("cfg5" if args.encoder == "resnet50" else "fp32")
        try:
            roofline = roofline_leg(step, model, trainer, args.dtype, layer_table=args.layer_table, pmc_tag=tag)
        finally:
            if store is not None:
                store.set("udaseg_bench_roofline_done", "1")
    elif store is not None and not args.no_roofline:
        import datetime
        store.wait(["udaseg_bench_roofline_done"], datetime.timedelta(seconds=600))
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / dt
        gf = CONV_GFLOP_PER_IMAGE.get((args.workload, args.encoder, args.size))
        peak = BF16_MFMA_PEAK_TFLOPS if args.dtype == "bf16" else FP32_MFMA_PEAK_TFLOPS
        out = {
            "metric": "training images/sec at 512x512" if headline
            else f"{args.workload} images/sec at {args.size}x{args.size} (informational, {args.encoder}, {args.dtype})",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.encoder}-Unet " + WORKLOAD_TEXT[args.workload] + ", "
                                   f"batch {args.batch}x3x{args.size}x{args.size} per GPU, {args.classes} classes, random init",
                       "global_batch": args.batch * world, "image": f"{args.size}x{args.size}", "parallelism": f"dp{world}",
                       "final_loss": round(final_loss, 5),
                       "host_cores_per_rank": (pinned or {}).get("cores") if world > 1 else host_cores(),
                       "launcher_numa_node": (pinned or {}).get("numa_node") if world > 1 else None,
                       "arithmetic": ARITHMETIC_FP32 if args.dtype == "fp32" else
                       "bf16 storage of activations / weight copies, bf16 MFMA with fp32 accumulation, fp32 master weights, "
                       "statistics, gradients and optimizer state",
                       # ALGORITHMIC conv FLOP rate per GPU (133.30 GFLOP per image), TFLOP/s
                       "conv_tflops_per_gpu": round(value * gf / 1e3 / world, 2) if gf else None,
                       # north_star's "conv MFMA util %", stated against the pipes the kernels actually run on: per step, every conv
                       # kernel's pipe work (6 bf16 products per fp32 product for the three-term-split kernels, FLOPs as they are
                       # for the fp32-MFMA ones) divided by ITS pipe's dense peak (2.5 PFLOP/s bf16, 157.3 TFLOP/s fp32), summed, over
                       # the measured step time.  (Round 3's conv_mfma_util_per_gpu divided by the fp32 peak a pipe most FLOPs no
                       # longer touch.)
                       "conv_pipe_util_per_gpu": round(roofline["pipe_ms_per_step"] / (1e3 * dt / args.steps), 4) if roofline else None},
            "step_ms": {"median": round(percentile(ev_ms, 0.5), 3), "p10": round(percentile(ev_ms, 0.1), 3),
                        "p90": round(percentile(ev_ms, 0.9), 3), "timer": "hipEvent pair per step on the compute stream"},
            "sustained": sustained,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.encoder, args.classes, args.size)
        if world == 1 and headline and not args.no_also and not rehearse:
            del step, model, trainer
            torch.cuda.empty_cache()
            also = [
                also_leg("BASELINE cfg 2 on the fp32-MFMA kernels only (UDASEG_F32_SPLIT=0)", "segmentation", "resnet18", "fp32", 8, 512,
                         args.classes, dev, split=False),
                also_leg("BASELINE cfg 3", "adversarial", "resnet18", "bf16", 8, 512, args.classes, dev,
                         cpu=not args.no_cpu_baseline),
                also_leg("BASELINE cfg 5 (per-GPU work)", "segmentation", "resnet50", "bf16", 8, 768, args.classes, dev),
            ]
            out["also"] = also
            # the driver's record keeps the SCALAR keys of `config` only (round 4's nested copy was dropped): one flat key per figure
            for name, a in zip(("fp32_mfma_only", "cfg3", "cfg5"), also):
                out["config"][f"{name}_images_per_s"] = a["value"]
                out["config"][f"{name}_ms_per_step"] = a["ms_per_step"]
                out["config"][f"{name}_frac"] = a["roofline"]["frac"]
                out["config"][f"{name}_dominant_kernel"] = a["roofline"]["kernel"]
                out["config"][f"{name}_conv_pipe_util"] = a["conv_pipe_util"]
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
