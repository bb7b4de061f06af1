"""Shared machinery of the HIP-backed networks: parameter arenas, layer holders, conv+BN+activation plans.

Design (MI355X-first):
* every parameter of a network lives in ONE flat fp32 arena in HBM (conv weights physically OHWI, channel counts
  padded to multiples of 4); the ``nn.Parameter`` objects the reference's code sees (OIHW shapes, smp / reference
  ``state_dict`` keys) are strided views into it.  Gradients are produced into a mirror arena, so the optimizer is one
  fused pass and the data-parallel all-reduce works on large contiguous buckets;
* a network's forward/backward is an explicit plan of C-ABI kernel launches behind a single ``autograd.Function``
  (no per-op autograd graph, no per-op allocator traffic beyond activations).
"""
import math
import os
import weakref

import torch
import torch.nn as nn

from . import kernels as K
from ._lib import ACT_LEAKY, ACT_NONE

ALIGN = 64  # floats; every arena entry starts on a 256-byte boundary
# Weight gradients run on a side HIP stream (see Plan.begin_backward).  UDASEG_SERIAL=1 keeps everything on one stream:
# per-kernel durations are then free of overlap slow-down (bench.py's roofline leg and the committed rocprof stats use it).
SIDE_STREAM_WGRAD = os.environ.get("UDASEG_SERIAL", "0") != "1"
# BatchNorm-backward reductions of single-consumer activations in the consumer's data-gradient epilogue (UDASEG_FUSE_BN_REDUCE=0:
# always the stand-alone reduce kernel; tests flip the module attribute to cross-check)
FUSE_BN_REDUCE = os.environ.get("UDASEG_FUSE_BN_REDUCE", "1") != "0"
# The backward pass needs the weights in three more packings (the data-gradient transpose, its fragment packing, the phase packing:
# 20 + 27 + 14 us in front of the first data gradient).  They depend on the weights alone, which are final once the optimizer has
# stepped: UDASEG_PREPACK=2 (default) issues all three on the side stream at the START of the forward, where they run beside the stem
# and the first encoder blocks (matrix-pipe-bound) -- cfg 2 1026.9 -> 1033.5 images/s over four alternations, the bf16 legs unchanged
# (profiles/r05_prepack.txt).  History: round 2 measured the transpose alone there at -0.4 % (a 0.1 ms element-wise pass then); this
# round's first attempt put the three beside the LOSS kernels, which are HBM-bound like they are: neutral.  =1: the transpose only
# (+0.1 %); =0: everything on the main stream in front of the backward (A/B).
PREPACK_DGRAD = os.environ.get("UDASEG_PREPACK", "2") in ("1", "2")
PREPACK_ALL = os.environ.get("UDASEG_PREPACK", "2") == "2"       # also the backward's fragment / phase packings
# bf16 storage: the stride-1 3x3 / 1x1 convolutions run on the bf16-first kernels (halo staged once, fragment-packed weights);
# UDASEG_FRAG=0 keeps them on the shared implicit-GEMM source (A/B, cross-check; tests flip the module attribute)
USE_FRAG_KERNELS = os.environ.get("UDASEG_FRAG", "1") != "0"
# "auto": a layer takes them where the library's measured heuristic prefers them (udaseg_conv_frag_preferred); "always": wherever
# they are supported (UDASEG_FRAG=2; network-level tests at small sizes, where "auto" would leave most layers on the old kernel)
FRAG_POLICY = "always" if os.environ.get("UDASEG_FRAG") == "2" else "auto"
# fp32 storage: the stride-1 3x3 convolutions run on the bf16 matrix pipe with every operand split exactly into three bf16
# terms (csrc/conv_halo_f32x3.hip: six products per operand pair, fp32 accumulation, one unit in the last place of a PRODUCT
# left out).  UDASEG_F32_SPLIT=0 keeps them on the fp32-MFMA kernels (A/B; tests flip the module attribute to cross-check).
USE_F32_SPLIT = os.environ.get("UDASEG_F32_SPLIT", "1") != "0"
# fp32 storage (round 5): the up-sampled half of a decoder conv1's fused input runs as four 2x2 phase convolutions of the
# half-resolution tensor with pre-summed weights (csrc/conv_up_f32x3.hip: 4 taps instead of 9, forward and data gradient; the skip
# half is a plain 3x3 convolution of its own).  UDASEG_UP_PHASE=0: the nine-tap gather over the virtual concatenation (A/B; tests
# flip the module attribute to cross-check)
USE_UP_PHASE = os.environ.get("UDASEG_UP_PHASE", "1") != "0"
# ... and the weight gradient (conv_wgrad_up_kernel: 16 phase-tap correlations at the half resolution).  Built, graded against f64
# (tests/test_gpu_up.py) and OFF by default: inside the overlapped step it is neutral (same box, images/s: off 983.7 / 996.7, on
# 984.8 - 995.2 over two tile shapes and three block counts, profiles/r05_up_phase.txt) -- it does 4/9 of the nine-tap kernel's matrix
# work but stages the same bytes per pixel for it, and the weight gradients share the chip with the main stream's chain, so what
# counts is CU-time, not FLOPs.  UDASEG_UP_PHASE_WGRAD=1 switches it on.
USE_UP_PHASE_WGRAD = os.environ.get("UDASEG_UP_PHASE_WGRAD", "0") == "1"
# fp32 storage (round 5): 3x3 layers that produce exactly 16 channels (decoder block 4 conv2 forward / data gradient, the head's data
# gradient) on the sixteen-wide matrix tile (csrc/conv_n16_f32x3.hip).  UDASEG_N16=0: the 32-row tile (A/B, cross-check)
USE_N16 = os.environ.get("UDASEG_N16", "1") != "0"
# fp32 storage (round 5): the 7x7 / stride 2 stem forward on its own kernel (csrc/conv_stem_f32x3.hip).  UDASEG_STEM=0: the shared
# implicit-GEMM source (A/B, cross-check)
USE_STEM = os.environ.get("UDASEG_STEM", "1") != "0"
# fp32 storage, phase form: the skip half of a decoder conv1 is a convolution of an ENCODER feature alone -- it is launched on the side
# stream as soon as that feature exists and runs beside the rest of the encoder / the earlier decoder blocks (the forward has no other
# side-stream work; its many small BatchNorm launches leave most of the chip idle).  UDASEG_PRELAUNCH_SKIP=0: on the main stream, in
# program order (A/B)
PRELAUNCH_SKIP = os.environ.get("UDASEG_PRELAUNCH_SKIP", "1") != "0"


# bf16 storage: BatchNorm + activation of a layer whose ONLY consumer is a convolution on the bf16-first kernels is not written at
# all -- the consumer applies it while it stages its input (UDASEG_FUSE_BN_APPLY=0: always the stand-alone bn_apply pass)
FUSE_BN_APPLY = os.environ.get("UDASEG_FUSE_BN_APPLY", "0") != "0"
# Round 4: OFF by default.  The small-GEMM 1x1 kernel (csrc/conv_halo_bf16.hip: conv1x1_gemm_bf16_kernel) fills LDS by DMA and cannot
# transform what it gathers, and r50's 1x1 consumers are exactly its layers: cfg 5 462.1 images/s with the unwritten activations
# (their consumers then stay on the streaming kernel), 465.1 without (profiles/r04_gemm_1x1.txt).  "2": only in front of 1x1 consumers.  The weight gradient of a 3x3 consumer gathers every input element once per tap
# (in different blocks), so the transform is evaluated nine times per element there and costs more than the pass it saves
# (measured, profiles/r03_bn_fusion_ab.txt: cfg 3 / cfg 5 images/s off 987.8 / 395.3, everywhere ("1") 990.7 / 397.9, 1x1
# consumers only 993.5 / 399.3; wgrad 256 -> 256 at 48^2 92 -> 136 us with the transform in its gather).
FUSE_BN_APPLY_1X1_ONLY = os.environ.get("UDASEG_FUSE_BN_APPLY", "0") == "2"
# fp32 storage (round 4): the same for the <= 32-channel full-resolution decoder layers, whose stand-alone bn_apply pass moves
# 268 MB per layer at 8 x 512^2: the consumer's forward (one-role split kernel) and weight gradient (small-channel direct kernel)
# apply act(fma(y, scale, shift)) while they stage; the producer's BatchNorm backward re-evaluates its mask from y anyway.
# UDASEG_FUSE_BN_APPLY_F32=0: always the stand-alone pass (A/B, cross-check; tests flip the module attribute)
# Default: consumers of <= 32 channels on both sides only.  The kernels take every stride-1 3x3 geometry ("2": everywhere), but on
# the >= 64-channel layers the transform in the weight gradient's staging costs more than the 8-33 MB passes it saves (same box,
# images/s: off 957.8, <= 32 channels 976.0, everywhere 964.5; profiles/r04_bn_unwritten_fp32.txt).
FUSE_BN_APPLY_F32 = os.environ.get("UDASEG_FUSE_BN_APPLY_F32", "1") != "0"
FUSE_BN_APPLY_F32_SMALL_ONLY = os.environ.get("UDASEG_FUSE_BN_APPLY_F32", "1") != "2"
# ... and through the next decoder block's up-sampling (block output -> conv1 of a block without a skip input): built and tested,
# off by default -- the 67 MB pass it saves is paid back in the consumer's staging (964.4 with, 966.3 without, 952.9 all off)
FUSE_BN_APPLY_F32_UP = os.environ.get("UDASEG_FUSE_BN_APPLY_F32_UP", "0") == "1"
# >= 64-channel layers: the consumer's forward (wave-specialised kernel) can write the activation out while it stages it, so that
# its weight gradient reads a plain tensor (UDASEG_FUSE_BN_APPLY_F32_WRITE=1).  Built, bit-exact (tests), and OFF: 950.8 against 958.0
# images/s without it (same box, three alternations) -- the transform + stores in the loader waves cost more than the eleven
# launch-floor-bound bn_apply passes they replace.
FUSE_BN_APPLY_F32_WRITE = os.environ.get("UDASEG_FUSE_BN_APPLY_F32_WRITE", "0") == "1"


# bf16 storage: weight gradients of the stride-1 3x3 layers with channel counts that are multiples of 64 on the halo-resident
# kernel (few long-lived blocks, x halo and dy staged once per tile); UDASEG_WGRAD_HALO=0: per-tap split-K kernel everywhere
USE_WGRAD_HALO = os.environ.get("UDASEG_WGRAD_HALO", "1") != "0"


# HIP priority of that stream (0 = default, -1 = high): UDASEG_SIDE_PRIORITY (measured on the fp32 step: 857 / 854 images/s, no effect)
SIDE_STREAM_PRIORITY = int(os.environ.get("UDASEG_SIDE_PRIORITY", "0"))
_SIDE_STREAMS = {}   # device -> the one side HIP stream the weight gradients of every network on that device run on
_ARENA_OWNERS = {}   # parameter-arena storage pointer -> weakref of the ArenaModule that owns it


def arena_owner(storage_ptr):
    """The live network whose parameter arena starts at ``storage_ptr`` (optim.FusedAdam asks before it updates a whole
    storage in one pass), or None."""
    ref = _ARENA_OWNERS.get(storage_ptr)
    net = ref() if ref is not None else None
    if net is None or net._arena is None or net._arena.untyped_storage().data_ptr() != storage_ptr:
        _ARENA_OWNERS.pop(storage_ptr, None)
        return None
    return net


_PADDED_INPUTS = {}   # data pointer -> weakref of a channel-padded NHWC input buffer made by data.prepare_batch


def mark_padded_input(buf):
    """data.prepare_batch registers the buffer whose [N,3,H,W]-shaped view it returns: only such buffers are read in place
    by the stem convolution (their padding lanes are zeros by construction; an arbitrary 4-channel channels_last tensor
    sliced [:, :3] is NOT one of them and takes the copy path)."""
    for k in [k for k, r in _PADDED_INPUTS.items() if r() is None]:
        del _PADDED_INPUTS[k]
    _PADDED_INPUTS[buf.data_ptr()] = weakref.ref(buf)


def is_padded_input(ptr):
    r = _PADDED_INPUTS.get(ptr)
    t = r() if r is not None else None
    return t is not None and t.data_ptr() == ptr


def ceil4(c):
    return (c + 3) // 4 * 4


def ceil_to(c, a):
    return (c + a - 1) // a * a


class UpCat:
    """A convolution input that is never materialised: cat([nearest_x2(a), skip], channels); ``skip`` may be None."""
    __slots__ = ("a", "skip")

    def __init__(self, a, skip):
        self.a, self.skip = a, skip


class UpGrad:
    """Destination of a fused decoder input's data gradient in the phase form: ``da`` the gradient of the HALF-resolution source
    itself (accumulated when ``acc``), ``d_skip`` the skip source's (or None)."""
    __slots__ = ("da", "acc", "d_skip")

    def __init__(self, da, acc, d_skip):
        self.da, self.acc, self.d_skip = da, acc, d_skip


class LazyAct:
    """act(batchnorm(y)) that is never written: ``y`` is the producer's raw convolution output, ``scale`` / ``shift`` the finalised
    per-channel coefficients (udaseg_bn_finalize).  Its single consumer applies act(fma(y, scale, shift)) rounded to bf16 while
    staging (forward convolution, weight gradient); the producer's BatchNorm backward re-evaluates the mask the same way."""
    __slots__ = ("y", "scale", "shift", "act", "slope", "z", "z_valid")

    def __init__(self, y, scale, shift, act, slope, z=None):
        self.y, self.scale, self.shift, self.act, self.slope = y, scale, shift, act, slope
        # fp32 write-through (round 4): a buffer the consumer's FORWARD fills with the activation while it stages y (z_valid then);
        # the consumer's weight gradient reads it like any tensor
        self.z, self.z_valid = z, False

    shape = property(lambda self: self.y.shape)
    dtype = property(lambda self: self.y.dtype)
    device = property(lambda self: self.y.device)

    def materialize(self):
        """The activation as a tensor (tests / debugging only): one rounding of y * scale + shift like the kernels' fused
        multiply-add (evaluated in f64, rounded once to fp32), activation, round to the storage type."""
        if self.z_valid:
            return self.z
        t = torch.addcmul(self.shift.double(), self.y.double(), self.scale.double()).float()
        if self.act != ACT_NONE:
            t = torch.where(t > 0, t, self.slope * t)
        return t.to(self.y.dtype)


class ConvP(nn.Module):
    """Parameter holder for one convolution (logical OIHW ``weight``, optional ``bias``); no forward of its own."""

    def __init__(self, cin, cout, k, stride=1, pad=0, bias=False):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self.needs_dgrad = True
        self.up_ca = 0          # > 0: a decoder conv1 whose first up_ca input channels are a nearest x2 up-sampled tensor
        self.align = 4          # physical channel granule: 4 (fp32 storage) or 8 (bf16 storage), set by the owning network

    @property
    def cin_p(self):
        return ceil_to(self.cin, self.align)

    @property
    def cout_p(self):
        return ceil_to(self.cout, self.align)

    def extra_repr(self):
        return f"{self.cin}, {self.cout}, kernel_size={self.k}, stride={self.stride}, padding={self.pad}, bias={self.bias is not None}"


class BNP(nn.Module):
    """Parameter/buffer holder for one BatchNorm2d (eps 1e-5, momentum 0.1: the traced reference values)."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.c, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def extra_repr(self):
        return f"{self.c}, eps={self.eps}, momentum={self.momentum}"


class ArenaModule(nn.Module):
    """Base of the HIP-backed networks: owns the flat parameter / buffer arenas."""

    def __init__(self):
        super().__init__()
        self._arena = None          # flat fp32 parameters
        self._buf_arena = None      # flat fp32 BN running stats
        self._entries = []          # (param, offset, numel_physical, physical_shape, logical_view_fn)
        self._pvec_cache = {}       # (id(module), name) -> view of the arena (Plan.pvec)
        self._param_list = []
        self._wt_arena = None       # dgrad-packed weights (scratch, refreshed every backward)
        self._wt_off = {}
        self._nbt = None            # int64 arena of the BN num_batches_tracked counters
        self._idx = {}
        self._nbn = 0
        self.compute_dtype = torch.float32

    def set_compute_dtype(self, dtype):
        """torch.float32 (default; BASELINE configs 1-2, 4) or torch.bfloat16 (configs 3, 5): bf16 activations / weight
        copies / MFMA with fp32 accumulation, statistics, master weights, gradients and optimizer state."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        for m in self.modules():
            if isinstance(m, ConvP):
                m.align = 8 if dtype == torch.bfloat16 else 4
        if self._arena is not None:
            self.build_arena()          # physical channel padding changed: re-lay the arena (values are preserved)
        return self

    # ---- arena construction ------------------------------------------------------------------------------------
    def _phys_shape(self, owner, name, p):
        if isinstance(owner, ConvP) and name == "weight":
            return (owner.cout_p, owner.k, owner.k, owner.cin_p)
        return (ceil4(p.numel()),)

    @staticmethod
    def _logical_view(flat, owner, name, p_shape):
        if isinstance(owner, ConvP) and name == "weight":
            co, ci, k, _ = p_shape
            return flat.view(owner.cout_p, k, k, owner.cin_p)[:co, :, :, :ci].permute(0, 3, 1, 2)
        return flat[: p_shape[0]] if len(p_shape) == 1 else flat.view(p_shape)

    def _owners(self):
        for mod in self.modules():
            for name, p in mod._parameters.items():
                if p is not None:
                    yield mod, name, p

    def build_arena(self, device=None):
        """(Re)build the flat arenas from the parameters' current values and re-point ``.data`` into them."""
        owners = list(self._owners())
        device = device or owners[0][2].device
        layout, off = [], 0
        for mod, name, p in owners:
            shp = self._phys_shape(mod, name, p)
            n = math.prod(shp)
            layout.append((mod, name, p, off, n, shp))
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        arena = torch.zeros(off, device=device, dtype=torch.float32)
        self._entries, self._param_list = [], []
        with torch.no_grad():
            for mod, name, p, o, n, shp in layout:
                view = self._logical_view(arena[o:o + n], mod, name, tuple(p.shape))
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = view
                self._entries.append((p, o, n, shp, mod, name))
                self._param_list.append(p)
        self._arena = arena
        _ARENA_OWNERS[arena.untyped_storage().data_ptr()] = weakref.ref(self)
        # BN running statistics
        bns = [m for m in self.modules() if isinstance(m, BNP)]
        boff, blay = 0, []
        for m in bns:
            blay.append((m, boff))
            boff += 2 * ((m.c + ALIGN - 1) // ALIGN * ALIGN)
        bufs = torch.zeros(max(boff, 1), device=device, dtype=torch.float32)
        with torch.no_grad():
            for m, o in blay:
                half = (m.c + ALIGN - 1) // ALIGN * ALIGN
                rm, rv = bufs[o:o + m.c], bufs[o + half:o + half + m.c]
                rm.copy_(m.running_mean.to(device))
                rv.copy_(m.running_var.to(device))
                m._buffers["running_mean"], m._buffers["running_var"] = rm, rv
        self._buf_arena = bufs
        nbt = torch.zeros(max(len(bns), 1), device=device, dtype=torch.long)
        with torch.no_grad():
            for i, m in enumerate(bns):
                nbt[i] = m.num_batches_tracked.to(device)
                m._buffers["num_batches_tracked"] = nbt[i]
        self._nbt = nbt
        self._nbn = sum(2 * ceil4(m.c) for m in bns)
        self._idx = {(id(e[4]), e[5]): (e[1], e[2], e[3]) for e in self._entries}
        # scratch for dgrad-packed weights
        self._wt_off, woff = {}, 0
        for m in self.modules():
            if isinstance(m, ConvP) and m.needs_dgrad:
                self._wt_off[id(m)] = woff
                woff += m.cout_p * m.k * m.k * m.cin_p
        self._wt_arena = torch.empty(max(woff, 8), device=device, dtype=self.compute_dtype)
        rows = []
        for m in self.modules():
            if isinstance(m, ConvP) and m.needs_dgrad:
                rows.append([self._idx[(id(m), "weight")][0], self._wt_off[id(m)], m.cout_p, m.k * m.k, m.cin_p])
        self._wt_table = torch.tensor(rows or [[0, 0, 0, 1, 0]], dtype=torch.int32, device=device)
        # bf16 storage: MFMA-fragment packings of the stride-1 3x3 / 1x1 convolutions for the bf16-first kernels
        # (csrc/conv_halo_bf16.hip), refreshed once per step by one batched launch per direction
        self._frag_off, self._frag_fwd_table, self._frag_bwd_table, self._frag_arena = {}, None, None, None
        f32 = self.compute_dtype == torch.float32
        planes = 3 if f32 else 1          # fp32: the three split planes of every packing
        if (USE_FRAG_KERNELS and self.compute_dtype == torch.bfloat16) or (USE_F32_SPLIT and f32):
            foff, frows, brows = 0, [], []
            for m in self.modules():
                if isinstance(m, ConvP) and m.stride == 1 and ((m.k == 3 and m.pad == 1) or (m.k == 1 and m.pad == 0 and not f32)):
                    nf = planes * K.frag_elems(m.cout_p, m.cin_p, m.k)
                    frows.append([0, self._idx[(id(m), "weight")][0], foff, m.cout_p, m.cin_p, m.k])
                    ent = [foff, nf, None, 0]
                    foff += nf
                    if m.needs_dgrad:
                        nd = planes * K.frag_elems(m.cin_p, m.cout_p, m.k)
                        brows.append([1, self._wt_off[id(m)], foff, m.cin_p, m.cout_p, m.k])
                        ent[2], ent[3] = foff, nd
                        foff += nd
                    self._frag_off[id(m)] = tuple(ent)
                elif (isinstance(m, ConvP) and not f32 and m.k == 4 and m.stride == 2 and m.pad == 1
                      and m.cin_p & (m.cin_p - 1) == 0):
                    # the discriminator's 4x4 / stride 2 convolutions as 2x2 windows: the forward over 4 cin phase-major "virtual"
                    # channels (pack mode 2), the data gradient as four parity classes (modes 3..6), see csrc/conv_halo_bf16.hip
                    nf = K.frag_elems(m.cout_p, 4 * m.cin_p, 2)
                    frows.append([2, self._idx[(id(m), "weight")][0], foff, m.cout_p, m.cin_p, 4])
                    ent = [foff, nf, None, 0]
                    foff += nf
                    if m.needs_dgrad:
                        fe = K.frag_elems(m.cin_p, m.cout_p, 2)
                        for e in range(4):
                            brows.append([3 + e, self._wt_off[id(m)], foff + e * fe, m.cin_p, m.cout_p, 4])
                        ent[2], ent[3] = foff, 4 * fe
                        foff += 4 * fe
                    self._frag_off[id(m)] = tuple(ent)
            # fp32: phase packings of the decoder conv1 layers (csrc/conv_up_f32x3.hip; table rows of 8: mode, src, dst, N, K, ldk)
            self._up_off, urows_f, urows_b = {}, [], []
            if f32 and USE_UP_PHASE:
                for m in self.modules():
                    cs = m.cin_p - m.up_ca if isinstance(m, ConvP) else 0
                    if not (isinstance(m, ConvP) and m.up_ca > 0 and m.k == 3 and m.stride == 1 and m.pad == 1 and m.needs_dgrad
                            and m.bias is None and m.up_ca % 16 == 0 and cs % 8 == 0 and m.cout_p % 8 == 0):
                        continue
                    wo, wto, ent = self._idx[(id(m), "weight")][0], self._wt_off[id(m)], {}
                    for key, mode, src, nn_, kk, ldk in (("up_fwd", 2, wo, m.cout_p, m.up_ca, m.cin_p),
                                                        ("up_bwd", 3, wto, m.up_ca, m.cout_p, m.cout_p),
                                                        ("skip_fwd", 0, wo + m.up_ca, m.cout_p, cs, m.cin_p),
                                                        ("skip_bwd", 1, wto + m.up_ca * 9 * m.cout_p, cs, m.cout_p, m.cout_p)):
                        if key.startswith("skip") and cs == 0:
                            continue
                        ne = 3 * K.frag_elems(nn_, kk, 4 if key.startswith("up") else 3)
                        (urows_f if key.endswith("fwd") else urows_b).append([mode, src, foff, nn_, kk, ldk, 0, 0])
                        ent[key] = (foff, ne)
                        foff += ne
                    self._up_off[id(m)] = ent
            # fp32: sixteen-wide-tile packings (csrc/conv_n16_f32x3.hip) of the 3x3 layers that produce (forward) or return gradients
            # for (data gradient) exactly 16 channels: decoder block 4 conv2, the head's data gradient
            self._n16_off = {}
            if f32 and USE_N16:
                for m in self.modules():
                    if not (isinstance(m, ConvP) and m.k == 3 and m.stride == 1 and m.pad == 1 and m.up_ca == 0):
                        continue
                    ent = {}
                    if m.cout_p == 16 and m.cin_p % 8 == 0 and m.cin_p <= 32 and m.bias is None:
                        ne = K.n16_frag_elems(m.cin_p)
                        urows_f.append([4, self._idx[(id(m), "weight")][0], foff, 16, m.cin_p, m.cin_p, 0, 0])
                        ent["fwd"] = (foff, ne)
                        foff += ne
                    if m.cin_p == 16 and m.needs_dgrad and m.cout_p % 8 == 0 and m.cout_p <= 32:
                        ne = K.n16_frag_elems(m.cout_p)
                        urows_b.append([5, self._wt_off[id(m)], foff, 16, m.cout_p, m.cout_p, 0, 0])
                        ent["bwd"] = (foff, ne)
                        foff += ne
                    if ent:
                        self._n16_off[id(m)] = ent
            self._stem_off = {}
            if f32 and USE_STEM:
                for m in self.modules():
                    if (isinstance(m, ConvP) and m.k == 7 and m.stride == 2 and m.pad == 3 and m.cin_p == 4 and m.cout_p == 64
                            and m.bias is None):
                        urows_f.append([8, self._idx[(id(m), "weight")][0], foff, 64, 4, 4, 0, 0])
                        self._stem_off[id(m)] = (foff, K.STEM_FRAG_ELEMS)
                        foff += K.STEM_FRAG_ELEMS
            self._up_fwd_table = torch.tensor(urows_f, dtype=torch.int32, device=device) if urows_f else None
            self._up_bwd_table = torch.tensor(urows_b, dtype=torch.int32, device=device) if urows_b else None
            if frows:
                self._frag_arena = torch.empty(foff, device=device, dtype=torch.bfloat16)
                self._frag_fwd_table = torch.tensor(frows, dtype=torch.int32, device=device)
                self._frag_bwd_table = torch.tensor(brows, dtype=torch.int32, device=device) if brows else None
        return self

    def _arena_ok(self, full=True):
        if self._arena is None:
            return False
        base = self._arena.data_ptr()
        ents = self._entries if full else (self._entries[0], self._entries[-1])
        for p, o, n, shp, mod, name in ents:
            if p.data_ptr() != base + 4 * o or p.device != self._arena.device:
                return False
        return True

    def ensure_arena(self):
        """Cheap per-forward check (first/last entry); anything that re-homes parameters goes through _apply."""
        if not self._arena_ok(full=False):
            self.build_arena()

    def _side_stream(self):
        """The weight-gradient side stream: ONE per device, shared by every network.  HIP multiplexes streams onto a few
        hardware queues; a fresh stream per network meant that the fourth network of a process (bench.py's cfg 5 leg) got a
        side stream aliased with the compute stream's queue, where every cross-stream event wait then blocked the
        compute stream itself: 240 instead of 320 images/s (round 2)."""
        dev = self._arena.device
        st = _SIDE_STREAMS.get(dev)
        if st is None:
            st = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev, priority=SIDE_STREAM_PRIORITY)
        return st

    def tick_batchnorm_counters(self):
        if self._nbt.is_cuda:
            K.check(K.ops.udaseg_add_i64(self._nbt, self._nbt.numel(), 1, None), "add_i64")      # one 4-us call instead of a torch add_
        else:
            self._nbt.add_(1)

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._arena = None  # .to()/.cuda() replaced the parameter storages: rebuild lazily
        return out

    # ---- helpers used by the plans -------------------------------------------------------------------------------
    def phys_weight(self, conv):
        """The conv's weight as its physical OHWI tensor [co_p, k, k, ci_p] (a view of the arena)."""
        for p, o, n, shp, mod, name in self._entries_of(conv):
            if name == "weight":
                return self._arena[o:o + n].view(shp)
        raise KeyError("conv not in arena")

    def _entries_of(self, mod):
        return [e for e in self._entries if e[4] is mod]

    def entry_index(self):
        """{(id(module), name): (offset, numel, phys_shape)}"""
        return self._idx

    def deliver_grads(self, garena):
        """Hand a finished gradient arena to the parameters' ``.grad`` (what autograd's AccumulateGrad would do, minus its
        clone of non-dense views): first backward since zero_grad -> ``.grad`` become views of ``garena``; later ones add
        onto the arena those views live in with one axpy.  Parameters with ``requires_grad=False`` get no ``.grad`` (a frozen
        encoder stays frozen under any optimizer)."""
        ps = [p for p in self._param_list if p.requires_grad]
        views = [g for p, g in zip(self._param_list, self.grad_views(garena)) if p.requires_grad]
        if all(p.grad is None for p in ps):
            for p, g in zip(ps, views):
                p.grad = g
            self._grad_arena = garena
            return
        if getattr(self, "grad_ready_hook", None) is not None:
            # the buckets of the previous backward's arena are being averaged in place on the all-reduce stream, and that
            # arena has already been divided by the world size: adding a second backward onto it would be wrong twice over
            raise RuntimeError("gradient accumulation over several backward passes is not supported while a GradAllReducer is "
                               "attached: call optimizer.zero_grad() (or detach the reducer) between backward passes")
        ga = getattr(self, "_grad_arena", None)
        if (ga is not None and ga.numel() == garena.numel() and all(p.grad is not None for p in ps) and ps
                and ps[0].grad.data_ptr() == views[0].data_ptr() - garena.data_ptr() + ga.data_ptr()
                and ps[-1].grad.data_ptr() == views[-1].data_ptr() - garena.data_ptr() + ga.data_ptr()):
            K.axpy(ga, garena, 1.0)
            return
        with torch.no_grad():
            for p, g in zip(ps, views):
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.add_(g)

    def grad_views(self, garena):
        """Logical-shape gradient views (one per parameter, in ``self._param_list`` order) over a grad arena: ONE as_strided per
        parameter from a cached (size, stride, offset) recipe -- the slice / view / slice / permute chain they replace cost 0.35 ms
        of host time per backward pass (92 parameters), and BASELINE cfg 3 is close to host-bound (profiles/r04_host_bound.txt)."""
        rec = getattr(self, "_view_recipes", None)
        if rec is None or rec[0] is not self._entries:
            rs = []
            for p, o, n, shp, mod, name in self._entries:
                v = self._logical_view(garena[o:o + n], mod, name, tuple(p.shape))
                rs.append((tuple(v.shape), tuple(v.stride()), v.storage_offset() - garena.storage_offset()))
            rec = self._view_recipes = (self._entries, rs)
        base = garena.storage_offset()
        return [garena.as_strided(sz, st, base + off) for sz, st, off in rec[1]]


class Plan:
    """One forward (and later backward) pass over an ArenaModule: scratch arenas + the launch helpers."""

    def __init__(self, net, training, save):
        self.net = net
        self.training = training
        self.save = save
        self.st = K.stream()
        self.idx = net.entry_index()
        dev = net._arena.device
        self.R = K.bn_replicas()
        K.ensure_workspace(dev)
        self.adt = net.compute_dtype                      # activation storage type
        self.bf16 = self.adt == torch.bfloat16
        # bf16 storage: one cast of the whole fp32 master arena per forward (same offsets / physical shapes)
        self.w16 = K.cast_to_bf16(net._arena, st=self.st) if self.bf16 else None
        self.frag = training and getattr(net, "_frag_arena", None) is not None      # bf16 kernels / fp32 three-term split
        if self.frag:
            K.pack_frag_batched(self.w16 if self.bf16 else net._arena, None, net._frag_arena, net._frag_fwd_table, self.st)
            if getattr(net, "_up_fwd_table", None) is not None:
                K.pack_up_batched(net._arena, None, net._frag_arena, net._up_fwd_table, self.st)
        nbn = net._nbn
        self.dev = dev
        if training:
            self.stats = K.zeros((max(nbn, 2) * self.R,), torch.float64, dev, self.st)  # [R][sum | sumsq] per BN
            self.saved_stats = torch.empty(max(nbn, 2), dtype=torch.float32, device=dev)  # [mean | rstd] per BN
            self.coefs = torch.empty(max(nbn, 2), dtype=torch.float32, device=dev)  # [scale | shift] of lazy BNs
        else:
            # inference: BatchNorm is folded into the conv weights (scratch refreshed per forward: one pass over the weights)
            self.fold_w = torch.empty_like(net._arena)
            self.fold_b = torch.empty(max(nbn // 2, 4), dtype=torch.float32, device=dev)
            self._fold_off = 0
        self._stat_off = 0
        self.garena = None
        self.bstats = None
        self._zeroed_early = False
        self._bstat_off = 0
        self.conv_flops = 0.0
        self._packed = None
        self._packed_all = False
        self._bnb = {}                    # id(conv output y) -> BN-backward sums already made by the consumer's data gradient
        self._producer = {}               # id(activation) -> the conv+BN+activation record that produced it (decoder block outputs)
        self._pre = {}                    # id(decoder conv1) -> (y holding its skip half, side stream handle): prelaunch_skip
        # the library's switchboard may change between two plans of one network (udaseg_set_option, the force_config entry points:
        # tests, tuning): every cached routing answer below is keyed with the number of overrides made so far
        self.epoch = K.option_epoch()
        self._wgrad_halo = USE_WGRAD_HALO and (self.bf16 or USE_F32_SPLIT)      # halo-resident weight gradients (bf16 / fp32 split)
        if training and save and SIDE_STREAM_WGRAD and PREPACK_DGRAD:
            self._prepack_dgrad_weights()

    def _prepack_dgrad_weights(self):
        """The data-gradient kernels want the weights as [ci][taps][co]: one batched repack of the arena per step.  The
        weights are final once the optimizer has stepped, so the repack runs on the side stream DURING the forward (it is
        a 0.1 ms HBM-bound pass the matrix-core-bound forward convolutions hide) instead of in front of the backward."""
        net = self.net
        main, side = torch.cuda.current_stream(), net._side_stream()
        side.wait_stream(main)                      # the optimizer step / load_state_dict that produced the weights
        K.pack_dgrad_batched(net._arena, net._wt_arena, net._wt_table, side.cuda_stream)
        if PREPACK_ALL:
            # ... and the two fragment packings made from it (fp32 split / bf16 kernels): everything the backward pass needs of the weights
            if self.frag and net._frag_bwd_table is not None:
                K.pack_frag_batched(None, net._wt_arena, net._frag_arena, net._frag_bwd_table, side.cuda_stream)
            if self.frag and getattr(net, "_up_bwd_table", None) is not None:
                K.pack_up_batched(None, net._wt_arena, net._frag_arena, net._up_bwd_table, side.cuda_stream)
            self._packed_all = True
            # ... and the zeroed gradient arena / BatchNorm-backward sums of this step (two fills that sat in front of the backward)
            self.garena = K.zeros_like(net._arena, side.cuda_stream)
            self.garena.record_stream(side)
            self.bstats = K.zeros((max(net._nbn, 2) * self.R,), torch.float64, self.dev, side.cuda_stream)
            self.bstats.record_stream(side)
            self._zeroed_early = True
        self._packed = torch.cuda.Event()
        self._packed.record(side)

    # -- parameter access
    def w(self, conv):
        o, n, shp = self.idx[(id(conv), "weight")]
        return (self.w16 if self.bf16 else self.net._arena)[o:o + n].view(shp)

    def b(self, conv):
        o, n, shp = self.idx[(id(conv), "bias")]
        return self.net._arena[o:o + n]

    def pvec(self, mod, name):
        # parameter vectors are views of the network's arena, which lives as long as the network does: cached there (a slice costs
        # ~1.3 us of host time, a step asks for ~140 of them)
        cache = self.net.__dict__.setdefault("_pvec_cache", {})
        key = (id(mod), name)
        v = cache.get(key)
        if v is None or v._base is not self.net._arena:
            o, n, shp = self.idx[key]
            v = cache[key] = self.net._arena[o:o + n]
        return v

    def gvec(self, mod, name):
        o, n, shp = self.idx[(id(mod), name)]
        return self.garena.as_strided((n,), (1,), self.garena.storage_offset() + o)

    def gw(self, conv):
        o, n, shp = self.idx[(id(conv), "weight")]
        return self.garena.as_strided(shp, (shp[1] * shp[2] * shp[3], shp[2] * shp[3], shp[3], 1), self.garena.storage_offset() + o)

    def offset_of(self, mod, name="weight"):
        return self.idx[(id(mod), name)][0]

    def wfrag(self, conv, d, dgrad=False, up_ca=0):
        """The conv's fragment-packed weights (forward or data gradient) when this launch can take the bf16-first kernels."""
        if not self.frag:
            return None
        ent = self.net._frag_off.get(id(conv))
        if ent is None or (dgrad and ent[2] is None):
            return None
        # the library's answer is a function of the geometry alone: asked once per (convolution, geometry, policy), the fragment
        # view cached with it (two ctypes calls and a slice per convolution and direction otherwise: host time, and cfg 3 runs close
        # to the host's launch rate)
        cache = self.net.__dict__.setdefault("_wfrag_cache", {})
        key = (id(conv), d.n, d.hi, d.wi, d.ci, d.co, bool(dgrad), up_ca, FRAG_POLICY, self.bf16, self.epoch)
        hit = cache.get(key)
        if hit is not None and hit[0] is self.net._frag_arena:
            return hit[1]
        ok = K.conv_frag_ok if FRAG_POLICY == "always" else K.conv_frag_preferred
        view = None
        if ok(d, dgrad, up_ca, f32=not self.bf16):
            o, n = (ent[2], ent[3]) if dgrad else (ent[0], ent[1])
            view = self.net._frag_arena[o:o + n]
        cache[key] = (self.net._frag_arena, view)
        return view

    def prelaunch_skip(self, conv, skip, up_ca):
        """Start the skip half of decoder conv1 ``conv`` (phase form) on the side stream now: y = conv3x3(skip, W[:, up_ca:]).  Called
        by the network as soon as the encoder feature ``skip`` exists; conv_bn_act picks the result up (self._pre)."""
        if not (PRELAUNCH_SKIP and SIDE_STREAM_WGRAD and self.training and self.save and skip is not None):
            return
        n, h, w, cs = skip.shape
        d = K.conv_desc(n, h, w, up_ca + cs, conv.cout_p, 3, 1, 1)
        upw = self.up_frag(conv, d, up_ca)
        if upw is None or "skip_fwd" not in upw:
            return
        side = self.net._side_stream()
        y = torch.empty((n, h, w, conv.cout_p), device=skip.device, dtype=self.adt)
        K.stream_wait(side.cuda_stream, self.st)             # the feature (and y's allocation) are ordered before the launch
        y.record_stream(side)
        skip.record_stream(side)
        K.conv2d_fwd_frag(K.conv_desc(n, h, w, cs, conv.cout_p, 3, 1, 1), skip, None, upw["skip_fwd"], None, y, st=side.cuda_stream)
        self._pre[id(conv)] = (y, side.cuda_stream)

    def up_frag(self, conv, d, up_ca):
        """The phase packings of a decoder conv1 ({"up_fwd", "up_bwd"[, "skip_fwd", "skip_bwd"]} -> views of the fragment arena) when
        this launch can run its up-sampled half as four 2x2 phase convolutions (csrc/conv_up_f32x3.hip), else None."""
        if not (self.frag and USE_UP_PHASE and not self.bf16):
            return None
        ent = getattr(self.net, "_up_off", {}).get(id(conv))
        if ent is None or up_ca != conv.up_ca:
            return None
        cache = self.net.__dict__.setdefault("_upfrag_cache", {})
        key = (id(conv), d.n, d.hi, d.wi, d.ci, d.co, self.epoch)
        hit = cache.get(key)
        if hit is not None and hit[0] is self.net._frag_arena:
            return hit[1]
        views = None
        if K.conv_up_ok(d, up_ca):
            views = {k: self.net._frag_arena[o:o + n] for k, (o, n) in ent.items()}
        cache[key] = (self.net._frag_arena, views)
        return views

    def stem_frag(self, conv, d):
        """The stem's fragment packing when this launch can take csrc/conv_stem_f32x3.hip."""
        if not (self.frag and USE_STEM and USE_F32_SPLIT and not self.bf16):
            return None
        ent = getattr(self.net, "_stem_off", {}).get(id(conv))
        if ent is None:
            return None
        cache = self.net.__dict__.setdefault("_stem_cache", {})
        key = (id(conv), d.n, d.hi, d.wi, self.epoch)
        hit = cache.get(key)
        if hit is not None and hit[0] is self.net._frag_arena:
            return hit[1]
        view = self.net._frag_arena[ent[0]:ent[0] + ent[1]] if K.conv_stem_ok(d) else None
        cache[key] = (self.net._frag_arena, view)
        return view

    def n16_frag(self, conv, d, dgrad):
        """The conv's sixteen-wide-tile packing (forward or data gradient) when this launch can take csrc/conv_n16_f32x3.hip."""
        if not (self.frag and USE_N16 and USE_F32_SPLIT and not self.bf16):
            return None
        ent = getattr(self.net, "_n16_off", {}).get(id(conv))
        key2 = "bwd" if dgrad else "fwd"
        if ent is None or key2 not in ent:
            return None
        cache = self.net.__dict__.setdefault("_n16_cache", {})
        key = (id(conv), d.n, d.hi, d.wi, d.ci, d.co, bool(dgrad), self.epoch)
        hit = cache.get(key)
        if hit is not None and hit[0] is self.net._frag_arena:
            return hit[1]
        view = None
        if K.conv_n16_ok(d, dgrad):
            o, n = ent[key2]
            view = self.net._frag_arena[o:o + n]
        cache[key] = (self.net._frag_arena, view)
        return view

    # -- forward pieces
    def conv(self, conv, x, act=ACT_NONE, slope=0.0, out_dtype=None):
        n, h, w, ci = x.shape
        assert ci == conv.cin_p, (ci, conv.cin_p)
        d = K.conv_desc(n, h, w, ci, conv.cout_p, conv.k, conv.stride, conv.pad)
        y = torch.empty((n, d.ho, d.wo, conv.cout_p), device=x.device, dtype=out_dtype or self.adt)
        wf = self.wfrag(conv, d)
        if isinstance(x, LazyAct):
            assert wf is not None, "a LazyAct input needs the fragment kernels (decided by the producer)"
            K.conv2d_fwd_frag(d, x.y, None, wf, self.b(conv) if conv.bias is not None else None, y, act, slope,
                              in_scale=x.scale, in_shift=x.shift, in_act=x.act, in_slope=x.slope, st=self.st)
        elif wf is not None:
            K.conv2d_fwd_frag(d, x, None, wf, self.b(conv) if conv.bias is not None else None, y, act, slope, st=self.st)
        else:
            K.conv2d_fwd(d, x, self.w(conv), self.b(conv) if conv.bias is not None else None, y, act, slope, False, self.st)
        return y, d

    def bn(self, bn, y, act, slope, residual=None, sums=None):
        """sums: this BN's statistic accumulators already filled by the producing conv's epilogue (else a stats pass)."""
        c = ceil4(bn.c)
        z = torch.empty_like(y)
        gamma, beta = self.pvec(bn, "weight"), self.pvec(bn, "bias")
        if self.training:
            if sums is None:
                sums, o = self._next_stats(c)
                K.bn_stats(y, sums, self.st)
            else:
                sums, o = sums
            mean, rstd = self.saved_stats[o:o + c], self.saved_stats[o + c:o + 2 * c]
            K.bn_apply(y, sums, gamma, beta, residual, z, bn.eps, bn.momentum, bn.running_mean, bn.running_var, mean, rstd,
                       act, slope, self.st)
            return z, (mean, rstd)
        K.bn_apply_eval(y, gamma, beta, bn.running_mean, bn.running_var, residual, z, bn.eps, act, slope, self.st)
        return z, None

    def _next_stats(self, c):
        o = self._stat_off
        self._stat_off += 2 * c
        return self.stats[o * self.R:(o + 2 * c) * self.R], o

    def like(self, t):
        """An empty tensor shaped like activation ``t`` (which may be a LazyAct)."""
        return torch.empty_like(t.y if isinstance(t, LazyAct) else t)

    def _lazy_ok(self, c, n, ho, wo, consumer, act, residual, up=False):
        """Cached per (consumer, geometry, switches): see _lazy_ok_uncached."""
        if consumer is None:
            return False
        cache = self.net.__dict__.setdefault("_lazy_cache", {})
        key = (id(consumer), c, n, ho, wo, act, residual is None, up, self.bf16, FUSE_BN_APPLY, FUSE_BN_APPLY_1X1_ONLY, FUSE_BN_REDUCE,
               FUSE_BN_APPLY_F32, FUSE_BN_APPLY_F32_SMALL_ONLY, FUSE_BN_APPLY_F32_UP, FUSE_BN_APPLY_F32_WRITE, USE_F32_SPLIT, FRAG_POLICY,
               self.frag, self.epoch)
        v = cache.get(key)
        if v is None:
            v = cache[key] = self._lazy_ok_uncached(c, n, ho, wo, consumer, act, residual, up)
        return v

    def _lazy_ok_uncached(self, c, n, ho, wo, consumer, act, residual, up=False):
        """May BatchNorm + activation of this [n, ho, wo, c] output stay unwritten?  Only when its single consumer runs on the
        bf16-first kernels in BOTH directions (the forward applies the transform while staging; the data gradient's epilogue
        makes this layer's BatchNorm-backward sums, which then need no activation either)."""
        if not self.bf16:
            # fp32: a 3x3 / stride 1 consumer of <= 32 channels on either side that takes the split forward kernel and the
            # small-channel weight gradient (kernels.conv_bnin_ok); nothing is asked of its data gradient
            if not (FUSE_BN_APPLY_F32 and USE_F32_SPLIT and self.frag and consumer is not None and residual is None and act != ACT_NONE):
                return False
            if consumer.k != 3 or consumer.stride != 1 or consumer.pad != 1 or consumer.cin_p != c or c % 8 != 0:
                return False
            if c > 32 or consumer.cout_p > 32:
                # wide layers: the consumer's forward writes the activation out as it stages it (its loader waves have the slack) and
                # its weight gradient reads that buffer -- the transform inside the weight gradient's staging costs more than the
                # pass saves ("2": that form everywhere, for A/B)
                if up:
                    return False
                d2 = K.conv_desc(n, ho, wo, c, consumer.cout_p, 3, 1, 1)
                if self.wfrag(consumer, d2) is None:
                    return False
                if FUSE_BN_APPLY_F32_WRITE and K.conv_bnin_writes(d2):
                    return "write"
                return (not FUSE_BN_APPLY_F32_SMALL_ONLY) and K.conv_bnin_ok(d2)
            if up and not FUSE_BN_APPLY_F32_UP:
                return False
            # up: the consumer is the next decoder block's conv1 behind a nearest x2 up-sampling, no skip input
            d2 = K.conv_desc(n, 2 * ho, 2 * wo, c, consumer.cout_p, 3, 1, 1) if up else K.conv_desc(n, ho, wo, c, consumer.cout_p, 3, 1, 1)
            if up and not K.upcat_fusable(c, 0, consumer.cout_p, torch.float32):      # the consumer block would materialise up(h)
                return False
            return K.conv_bnin_ok(d2, up) and self.wfrag(consumer, d2, up_ca=c if up else 0) is not None
        if up:
            return False
        if not (FUSE_BN_APPLY and FUSE_BN_REDUCE and self.frag and self.bf16 and consumer is not None and residual is None
                and act != ACT_NONE):
            return False
        if consumer.stride != 1 or consumer.cin_p != c or c % 16 != 0 or (FUSE_BN_APPLY_1X1_ONLY and consumer.k != 1):
            return False
        d2 = K.conv_desc(n, ho, wo, c, consumer.cout_p, consumer.k, 1, consumer.pad)
        return self.wfrag(consumer, d2) is not None and self.wfrag(consumer, d2, dgrad=True) is not None

    def conv_bn_act(self, conv, bn, x, act=ACT_LEAKY, slope=0.0, residual=None, lazy_for=None, lazy_up=False):
        """z = act(bn(conv(x)) (+ residual)); returns (z, record for backward).  In training the conv's epilogue also
        accumulates the BN statistics of its output (no separate pass over y).

        ``x`` may be an ``UpCat(a, skip)``: the convolution then runs on cat([nearest_x2(a), skip], channels) without that
        tensor ever being written (fused gather; smp's decoder block input)."""
        up = isinstance(x, UpCat)
        lazy_in = isinstance(x, LazyAct)
        if up:
            n, h, w = x.a.shape[0], 2 * x.a.shape[1], 2 * x.a.shape[2]
            ci, dev = x.a.shape[3] + (0 if x.skip is None else x.skip.shape[3]), x.a.device
            assert residual is None
        else:
            (n, h, w, ci), dev = x.shape, x.device
        bias = self.b(conv) if conv.bias is not None else None
        if self.training:
            d = K.conv_desc(n, h, w, ci, conv.cout_p, conv.k, conv.stride, conv.pad)
            pre = self._pre.pop(id(conv), None) if up else None
            y = pre[0] if pre is not None else torch.empty((n, d.ho, d.wo, conv.cout_p), device=dev, dtype=self.adt)
            sums = self._next_stats(ceil4(bn.c))
            wf = self.wfrag(conv, d, up_ca=x.a.shape[3] if up else 0)
            upw = self.up_frag(conv, d, x.a.shape[3]) if (up and not isinstance(x.a, LazyAct)) else None
            if pre is not None:
                assert upw is not None and x.skip is not None and tuple(y.shape) == (n, d.ho, d.wo, conv.cout_p)
                K.stream_wait(self.st, pre[1])                 # the skip half, started on the side stream when its feature appeared
            if upw is not None:
                # phase form: the skip half as a plain 3x3 convolution of its own, the up-sampled half (4 taps per phase) on top,
                # BatchNorm statistics of the sum in the second launch's epilogue
                assert bias is None
                if x.skip is not None and pre is None:
                    ds = K.conv_desc(n, h, w, x.skip.shape[3], conv.cout_p, 3, 1, 1)
                    K.conv2d_fwd_frag(ds, x.skip, None, upw["skip_fwd"], None, y, st=self.st)
                K.conv2d_fwd_up(d, x.a, upw["up_fwd"], y, accumulate=x.skip is not None, stats=sums[0], st=self.st)
            elif (not up and bias is None and not (lazy_in and x.z is not None)
                  and self.n16_frag(conv, d, False) is not None):          # 16 produced channels: the sixteen-wide tile
                n16 = self.n16_frag(conv, d, False)
                if lazy_in:
                    K.conv2d_fwd_n16(d, x.y, n16, y, stats=sums[0], in_scale=x.scale, in_shift=x.shift, in_act=x.act, in_slope=x.slope,
                                     st=self.st)
                else:
                    K.conv2d_fwd_n16(d, x, n16, y, stats=sums[0], st=self.st)
            elif not up and not lazy_in and bias is None and conv.k == 7 and self.stem_frag(conv, d) is not None:
                K.conv2d_fwd_stem(d, x, self.stem_frag(conv, d), y, stats=sums[0], st=self.st)      # the 7x7 / stride 2 stem
            elif lazy_in:
                assert wf is not None, "a LazyAct input needs the bf16-first kernels (decided by the producer)"
                K.conv2d_fwd_frag(d, x.y, None, wf, bias, y, stats=sums[0], in_scale=x.scale, in_shift=x.shift, in_act=x.act,
                                  in_slope=x.slope, st=self.st, z_out=x.z)
                x.z_valid = x.z is not None
            elif wf is not None and up and isinstance(x.a, LazyAct):       # the up-sampled source is an unwritten activation (fp32)
                assert x.skip is None
                K.conv2d_fwd_frag(d, x.a.y, None, wf, bias, y, stats=sums[0], in_scale=x.a.scale, in_shift=x.a.shift, in_act=x.a.act,
                                  in_slope=x.a.slope, up=True, st=self.st)
            elif wf is not None:
                K.conv2d_fwd_frag(d, x.a if up else x, x.skip if up else None, wf, bias, y, stats=sums[0], up=up, st=self.st)
            elif up:
                K.conv2d_fwd_upcat(d, x.a, x.skip, self.w(conv), bias, y, ACT_NONE, 0.0, sums[0], self.st)
            else:
                K.conv2d_fwd_bnstats(d, x, self.w(conv), bias, y, sums[0], self.st)
            c = ceil4(bn.c)
            lazy = y.shape[-1] == c and self._lazy_ok(c, n, d.ho, d.wo, lazy_for, act, residual, lazy_up)
            if lazy:
                so, o = sums
                mean, rstd = self.saved_stats[o:o + c], self.saved_stats[o + c:o + 2 * c]
                scale, shift = self.coefs[o:o + c], self.coefs[o + c:o + 2 * c]
                K.bn_finalize(so, self.pvec(bn, "weight"), self.pvec(bn, "bias"), n * d.ho * d.wo, bn.eps, bn.momentum,
                              bn.running_mean, bn.running_var, mean, rstd, scale, shift, self.st)
                z, ms = LazyAct(y, scale, shift, act, slope, torch.empty_like(y) if lazy == "write" else None), (mean, rstd)
            else:
                z, ms = self.bn(bn, y, act, slope, residual, sums)
        else:
            # eval mode: ONE kernel per conv+BN(+add)+activation block
            d = K.conv_desc(n, h, w, ci, conv.cout_p, conv.k, conv.stride, conv.pad)
            o, nel, shp = self.idx[(id(conv), "weight")]
            wf = self.fold_w[o:o + nel].view(shp)
            c = ceil4(bn.c)
            bf = self.fold_b[self._fold_off:self._fold_off + c]
            self._fold_off += c
            w32 = self.net._arena[o:o + nel].view(shp)
            K.bn_fold(w32, bias, self.pvec(bn, "weight"),
                      self.pvec(bn, "bias"), bn.running_mean, bn.running_var, bn.eps, wf, bf, self.st)
            if self.bf16:
                wf = K.cast_to_bf16(wf, st=self.st)
            z = torch.empty((n, d.ho, d.wo, conv.cout_p), device=dev, dtype=self.adt)
            if up:
                K.conv2d_fwd_upcat(d, x.a, x.skip, wf, bf, z, act, slope, None, self.st)
            else:
                K.conv2d_fwd_fused(d, x, wf, bf, residual, z, act, slope, self.st)
            return z, None
        # has_res tells the backward whether the activation's argument can be re-evaluated from y alone
        rec = (conv, bn, d, x, y, z, ms, act, slope, residual is not None) if self.save else None
        return z, rec

    # -- backward pieces
    def begin_backward(self):
        net = self.net
        self.st = K.stream()
        self.main_stream = torch.cuda.current_stream()
        # weight gradients are off the backward critical path (only Adam / the all-reduce consume them): they run on a
        # side HIP stream and fill the matrix cores while the main chain sits in HBM-bound BatchNorm-backward kernels
        self.side_stream = net._side_stream() if SIDE_STREAM_WGRAD else None
        nbn = net._nbn
        if self._zeroed_early:                      # zeroed on the side stream with the early packings (_prepack_dgrad_weights): once
            self._zeroed_early = False
        else:
            self.garena = K.zeros_like(net._arena, self.st)
            self.bstats = K.zeros((max(nbn, 2) * self.R,), torch.float64, self.dev, self.st)
        self._side_h = self.side_stream.cuda_stream if self.side_stream is not None else None
        self._main_h = self.main_stream.cuda_stream
        self._bstat_off = 0
        # dgrad needs the weights as [ci][taps][co]: one batched repack of the whole arena per step, normally issued
        # on the side stream at the start of the forward (_prepack_dgrad_weights)
        if self._packed is not None:
            self.main_stream.wait_event(self._packed)
        else:
            K.pack_dgrad_batched(net._arena, net._wt_arena, net._wt_table, self.st)
        if self.frag and net._frag_bwd_table is not None and not self._packed_all:
            K.pack_frag_batched(None, net._wt_arena, net._frag_arena, net._frag_bwd_table, self.st)
        if self.frag and getattr(net, "_up_bwd_table", None) is not None and not self._packed_all:
            K.pack_up_batched(None, net._wt_arena, net._frag_arena, net._up_bwd_table, self.st)
        if self.side_stream is not None:
            self.side_stream.wait_stream(self.main_stream)      # zeroed gradient arena is visible to the side stream

    def _wgrad_up_ok(self, conv, d, x):
        """May this fused decoder input's weight gradient run in the phase form?  (fp32 split kernels on, both halves served.)"""
        if self.bf16 or not (USE_UP_PHASE and USE_UP_PHASE_WGRAD and self._wgrad_halo and USE_F32_SPLIT) or isinstance(x.a, LazyAct):
            return False
        cache = self.net.__dict__.setdefault("_wgup_cache", {})
        key = (id(conv), d.n, d.hi, d.wi, d.ci, d.co, x.a.shape[-1], self.epoch)
        v = cache.get(key)
        if v is None:
            v = K.conv2d_wgrad_up_ok(d, x.a.shape[-1])
            if v and x.skip is not None:
                v = K.conv2d_wgrad_halo_ok(K.conv_desc(d.n, d.hi, d.wi, x.skip.shape[-1], d.co, 3, 1, 1), f32=True)
            cache[key] = v
        return v

    def packed_wt(self, conv):
        o = self.net._wt_off[id(conv)]
        n = conv.cout_p * conv.k * conv.k * conv.cin_p
        return self.net._wt_arena[o:o + n]

    def _next_bstats(self, c):
        o = self._bstat_off
        self._bstat_off += 2 * c
        return self.bstats[o * self.R:(o + 2 * c) * self.R]

    def conv_bwd(self, conv, d, x, dy, dx=None, dx_acc=False, dbias=None, prev=None):
        """dW (+ dbias) into the grad arena; dx (+)= dgrad when dx is given.  dbias: already-computed channel sums of dy.

        ``x`` an ``UpCat(a, skip)`` (fused decoder input): the weight gradient is one launch per source, ``dx`` is the pair
        (gradient of the UP-SAMPLED a [n,2h,2w,ca], gradient of skip) and both are overwritten.

        ``prev``: the conv+BN+activation record that PRODUCED x, when this convolution is x's only consumer: dx is then that
        activation's complete gradient and the data-gradient kernel's epilogue also makes the two reductions of its BatchNorm
        backward (``bn_bwd`` finds them in ``self._bnb`` and skips its reduce pass)."""
        side = self.side_stream
        if side is not None:
            wst = self._side_h
            K.stream_wait(wst, self._main_h)                    # dy is final here (a pooled event inside the library)
            dy.record_stream(side)                              # the caching allocator must not recycle dy under the side stream
        else:
            wst = self.st
        if isinstance(x, UpCat):
            gw = self.gw(conv)
            if isinstance(x.a, LazyAct):
                assert x.skip is None
                K.conv2d_wgrad_bnin(d, x.a.y, x.a.scale, x.a.shift, x.a.act, x.a.slope, dy, gw, True, wst, up=True)
            elif self._wgrad_up_ok(conv, d, x):
                # phase form: 16 phase-tap correlations at a's resolution for the up-sampled half, the skip half as a slice
                K.conv2d_wgrad_up(d, x.a, dy, gw, st=wst)
                if x.skip is not None:
                    ds = K.conv_desc(d.n, d.hi, d.wi, x.skip.shape[-1], d.co, 3, 1, 1)
                    K.conv2d_wgrad_halo_slice(ds, x.skip, dy, gw, x.a.shape[-1], st=wst)
            elif self._wgrad_halo and x.skip is not None and K.conv2d_wgrad_halo_ok(d, x.a.shape[-1], f32=not self.bf16):
                K.conv2d_wgrad_halo(d, x.a, x.skip, dy, gw, up=True, st=wst)      # both sources in one launch
            else:
                K.conv2d_wgrad_part(d, x.a, 0, True, dy, gw, True, wst)
                if x.skip is not None:
                    K.conv2d_wgrad_part(d, x.skip, x.a.shape[-1], False, dy, gw, True, wst)
            assert not dx_acc and conv.bias is None
            if isinstance(dx, UpGrad):          # phase form: the gradient of the half-resolution source at its own resolution
                upw = self.up_frag(conv, d, x.a.shape[-1])
                # prev: the record of the layer that produced x.a, whose only consumer this convolution is -- da is then that
                # activation's complete gradient (no 2x2 sum-pool pass follows any more) and the epilogue makes its BatchNorm-backward sums
                bn = None
                if (prev is not None and FUSE_BN_REDUCE and not dx.acc and not prev[9] and prev[7] != ACT_NONE
                        and not isinstance(prev[5], LazyAct) and prev[4].shape == dx.da.shape):
                    p_bn, p_y, (p_mean, p_rstd) = prev[1], prev[4], prev[6]
                    bs = self._next_bstats(ceil4(p_bn.c))
                    bn = (p_y, p_mean, p_rstd, self.pvec(p_bn, "weight"), self.pvec(p_bn, "bias"), prev[7], prev[8], bs)
                    self._bnb[id(p_y)] = bs
                K.conv2d_dgrad_up(d, dy, x.a.shape[-1], upw["up_bwd"], dx.da, accumulate=dx.acc, bn=bn, st=self.st)
                if x.skip is not None:
                    ds = K.conv_desc(d.n, d.hi, d.wi, x.skip.shape[-1], d.co, 3, 1, 1)
                    K.conv2d_dgrad_frag(ds, dy, upw["skip_bwd"], dx.d_skip, st=self.st)
                return
            d_up, d_skip = dx
            wfd = self.wfrag(conv, d, dgrad=True)
            if wfd is not None and (x.skip is None or x.a.shape[-1] % 32 == 0):
                K.conv2d_dgrad_frag(d, dy, wfd, d_up, d_skip if x.skip is not None else None, st=self.st)
            elif x.skip is None:
                K.conv2d_dgrad(d, dy, self.packed_wt(conv), d_up, False, self.st)
            else:
                K.conv2d_dgrad_split(d, dy, self.packed_wt(conv), d_up, d_skip, self.st)
            return
        if isinstance(x, LazyAct) and x.z_valid:
            x = x.z                                            # written out by this convolution's own forward
        if isinstance(x, LazyAct):
            K.conv2d_wgrad_bnin(d, x.y, x.scale, x.shift, x.act, x.slope, dy, self.gw(conv), True, wst)
        elif self._wgrad_halo and K.conv2d_wgrad_halo_ok(d, f32=not self.bf16):
            K.conv2d_wgrad_halo(d, x, None, dy, self.gw(conv), st=wst)
        else:
            K.conv2d_wgrad(d, x, dy, self.gw(conv), True, wst)
        if conv.bias is not None:
            if dbias is not None:
                K.axpy(self.gvec(conv, "bias"), dbias, 1.0, wst)
            else:
                K.channel_sum(dy, self.gvec(conv, "bias"), True, wst)
        if dx is not None:
            wfd = self.wfrag(conv, d, dgrad=True)
            n16 = self.n16_frag(conv, d, True) if not dx_acc else None      # 16 gradient channels: the sixteen-wide tile
            fuse = (prev is not None and FUSE_BN_REDUCE and not dx_acc and not prev[9] and prev[7] != ACT_NONE
                    and prev[4].shape == dx.shape and (wfd is not None or n16 is not None or K.conv2d_dgrad_bnreduce_ok(d, dy.dtype)))
            if prev is not None and isinstance(prev[5], LazyAct) and not fuse and self.bf16:
                raise RuntimeError("internal: the producer's activation was not written, its consumer's data gradient must make "
                                   "the BatchNorm-backward sums")
            if fuse:
                p_bn, p_y, (p_mean, p_rstd) = prev[1], prev[4], prev[6]
                bs = self._next_bstats(p_y.shape[-1] if self.bf16 else ceil4(p_bn.c))
                if n16 is not None:
                    K.conv2d_dgrad_n16(d, dy, n16, dx, bn=(p_y, p_mean, p_rstd, self.pvec(p_bn, "weight"), self.pvec(p_bn, "bias"),
                                                           prev[7], prev[8], bs), st=self.st)
                elif wfd is not None:
                    K.conv2d_dgrad_frag(d, dy, wfd, dx, bn=(p_y, p_mean, p_rstd, self.pvec(p_bn, "weight"), self.pvec(p_bn, "bias"),
                                                            prev[7], prev[8], bs), st=self.st)
                else:
                    K.conv2d_dgrad_bnreduce(d, dy, self.packed_wt(conv), dx, p_y, p_mean, p_rstd, self.pvec(p_bn, "weight"),
                                            self.pvec(p_bn, "bias"), prev[7], prev[8], bs, self.st)
                self._bnb[id(p_y)] = bs
            elif n16 is not None:
                K.conv2d_dgrad_n16(d, dy, n16, dx, st=self.st)
            elif wfd is not None:
                K.conv2d_dgrad_frag(d, dy, wfd, dx, accumulate=dx_acc, st=self.st)
            else:
                K.conv2d_dgrad(d, dy, self.packed_wt(conv), dx, dx_acc, self.st)

    def bn_bwd(self, bn, y, z, ms, dz, act, slope, dres=None, dres_acc=False, has_res=True):
        """In place: dz becomes dy (grad w.r.t. the conv output).  dres (+)= masked grad for the residual branch.
        fp32 layers without a residual input do not read z: the kernels re-evaluate the activation's argument from y."""
        c = ceil4(bn.c)
        mean, rstd = ms
        gamma = self.pvec(bn, "weight")
        beta = None
        if isinstance(z, LazyAct) and not self.bf16:
            z, has_res = None, False         # fp32: the z-less kernels below re-evaluate the activation's argument from y
        if isinstance(z, LazyAct):
            bs = self._bnb.pop(id(y))        # made by the consumer's data gradient (conv_bwd refuses to run without them)
            K.bn_bwd_apply_recompute(dz, y, z.scale, z.shift, mean, rstd, gamma, bs, dz, self.gvec(bn, "weight"),
                                     self.gvec(bn, "bias"), act, slope, self.st)
            return dz
        if not has_res and not self.bf16 and act != ACT_NONE:
            z, beta = None, self.pvec(bn, "bias")
        bs = self._bnb.pop(id(y), None)           # the consumer's data gradient already made the two reductions
        if bs is None:
            bs = self._next_bstats(c)
            K.bn_bwd_reduce(dz, z, y, mean, rstd, bs, act, slope, self.st, gamma=gamma, beta=beta)
        K.bn_bwd_apply(dz, z, y, mean, rstd, gamma, bs, dz, dres, self.gvec(bn, "weight"), self.gvec(bn, "bias"), act, slope,
                       False, dres_acc, False, self.st, beta=beta)
        return dz

    def conv_bn_act_bwd(self, rec, dz, dx=None, dx_acc=False, dres=None, dres_acc=False, prev=None):
        conv, bn, d, x, y, z, ms, act, slope, has_res = rec
        dy = self.bn_bwd(bn, y, z, ms, dz, act, slope, dres, dres_acc, has_res)
        self.conv_bwd(conv, d, x, dy, dx, dx_acc, prev=prev)


    def ready_events(self):
        """Events after which every gradient issued so far is final (one per stream in use); for the all-reduce stream."""
        evs = [torch.cuda.Event()]
        evs[0].record(self.main_stream)
        if self.side_stream is not None:
            e = torch.cuda.Event()
            e.record(self.side_stream)
            evs.append(e)
        return evs

    def join_side_stream(self):
        """Order the main stream after every side-stream weight gradient issued so far."""
        if self.side_stream is not None:
            self.main_stream.wait_stream(self.side_stream)


class GradSlots:
    """Gradient buffers of activations during one backward: first writer overwrites, later writers accumulate."""

    def __init__(self):
        self._g = {}

    def slot(self, t):
        """-> (buffer, accumulate?)"""
        k = id(t)
        if k in self._g:
            return self._g[k], True
        b = torch.empty_like(t.y if isinstance(t, LazyAct) else t)
        self._g[k] = b
        return b, False

    def get(self, t):
        return self._g[id(t)]

    def has(self, t):
        return id(t) in self._g

    def put(self, t, g):
        self._g[id(t)] = g

    def pop(self, t):
        return self._g.pop(id(t))
