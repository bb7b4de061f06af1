"""``Unet`` -- the encoder-decoder the reference creates with ``smp.Unet(encoder_name=..., encoder_weights=...,
in_channels=..., classes=...)`` (reference ``src/test_system.py:90-95``, ``src/models/train.py:572-577``), as a
drop-in ``nn.Module`` whose arithmetic runs in libudaseg_hip.so.

Architecture (pinned by the reference's traced-graph fixture): torchvision-ResNet encoder
(resnet18/34: BasicBlock, resnet50: Bottleneck with the stride on the 3x3), smp ``UnetDecoder`` with channels
(256,128,64,32,16) -- nearest x2 upsample, ``cat([up, skip], 1)``, two conv3x3+BN+ReLU per block -- and a
``Conv2d(16, classes, 3, padding=1)`` head.  ``state_dict`` keys follow smp (``encoder.layer1.0.conv1.weight``,
``decoder.blocks.0.conv1.0.weight``, ``segmentation_head.0.bias`` ...), so reference checkpoints
(``src/models/train.py:491-500``) load unchanged.

``forward`` returns logits shaped ``[N, classes, H, W]`` like the reference; physically they are NHWC with the class
dimension padded to a multiple of 4 (a strided view), which ``losses.CrossEntropyLoss`` consumes without a copy.
"""
import os
import warnings

import torch
import torch.nn as nn

from . import kernels as K
from ._lib import ACT_LEAKY, ACT_NONE, require_gpu
from .engine import LazyAct, ArenaModule, BNP, ConvP, GradSlots, Plan, UpCat, UpGrad, ceil4, is_padded_input

ENCODERS = {
    "resnet18": ("basic", (2, 2, 2, 2), (64, 64, 128, 256, 512)),
    "resnet34": ("basic", (3, 4, 6, 3), (64, 64, 128, 256, 512)),
    "resnet50": ("bottleneck", (3, 4, 6, 3), (64, 256, 512, 1024, 2048)),
}
DECODER_CHANNELS = (256, 128, 64, 32, 16)
RELU = (ACT_LEAKY, 0.0)
# UDASEG_MATERIALIZE_UPCAT=1: write cat([up(x), skip]) with the stand-alone kernel instead of gathering it inside conv1
# (cross-check of the fused gather; tests flip this switch)
FUSE_UPCAT = os.environ.get("UDASEG_MATERIALIZE_UPCAT", "0") != "1"


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = ConvP(inplanes, planes, 3, stride, 1)
        self.bn1 = BNP(planes)
        self.conv2 = ConvP(planes, planes, 3, 1, 1)
        self.bn2 = BNP(planes)
        self.downsample = nn.Sequential(ConvP(inplanes, planes, 1, stride, 0), BNP(planes)) if downsample else None

    def fwd(self, P, x):
        a1, r1 = P.conv_bn_act(self.conv1, self.bn1, x, *RELU, lazy_for=self.conv2)      # a1 feeds conv2 only
        if self.downsample is not None:
            idt, rd = P.conv_bn_act(self.downsample[0], self.downsample[1], x, ACT_NONE, 0.0)
        else:
            idt, rd = x, None
        out, r2 = P.conv_bn_act(self.conv2, self.bn2, a1, *RELU, residual=idt)
        return out, (x, a1, idt, r1, r2, rd)

    def bwd(self, P, G, rec, out):
        x, a1, idt, r1, r2, rd = rec
        d_out = G.pop(out)
        dx, dx_acc = G.slot(x)
        if rd is None:
            # identity branch: the masked gradient goes straight into dx; conv1's dgrad accumulates on top
            d_a1 = P.like(a1)
            P.conv_bn_act_bwd(r2, d_out, dx=d_a1, dres=dx, dres_acc=dx_acc, prev=r1)     # a1 feeds conv2 only
            P.conv_bn_act_bwd(r1, d_a1, dx=dx, dx_acc=True)
        else:
            d_idt = torch.empty_like(idt)
            d_a1 = P.like(a1)
            P.conv_bn_act_bwd(r2, d_out, dx=d_a1, dres=d_idt, prev=r1)
            P.conv_bn_act_bwd(rd, d_idt, dx=dx, dx_acc=dx_acc)
            P.conv_bn_act_bwd(r1, d_a1, dx=dx, dx_acc=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = ConvP(inplanes, planes, 1, 1, 0)
        self.bn1 = BNP(planes)
        self.conv2 = ConvP(planes, planes, 3, stride, 1)
        self.bn2 = BNP(planes)
        self.conv3 = ConvP(planes, planes * 4, 1, 1, 0)
        self.bn3 = BNP(planes * 4)
        self.downsample = nn.Sequential(ConvP(inplanes, planes * 4, 1, stride, 0), BNP(planes * 4)) if downsample else None

    def fwd(self, P, x):
        a1, r1 = P.conv_bn_act(self.conv1, self.bn1, x, *RELU, lazy_for=self.conv2)      # a1 feeds conv2 only
        a2, r2 = P.conv_bn_act(self.conv2, self.bn2, a1, *RELU, lazy_for=self.conv3)     # a2 feeds conv3 only
        if self.downsample is not None:
            idt, rd = P.conv_bn_act(self.downsample[0], self.downsample[1], x, ACT_NONE, 0.0)
        else:
            idt, rd = x, None
        out, r3 = P.conv_bn_act(self.conv3, self.bn3, a2, *RELU, residual=idt)
        return out, (x, a1, a2, idt, r1, r2, r3, rd)

    def bwd(self, P, G, rec, out):
        x, a1, a2, idt, r1, r2, r3, rd = rec
        d_out = G.pop(out)
        dx, dx_acc = G.slot(x)
        d_a2 = P.like(a2)
        d_a1 = P.like(a1)
        if rd is None:
            P.conv_bn_act_bwd(r3, d_out, dx=d_a2, dres=dx, dres_acc=dx_acc, prev=r2)     # a2 feeds conv3 only
        else:
            d_idt = torch.empty_like(idt)
            P.conv_bn_act_bwd(r3, d_out, dx=d_a2, dres=d_idt, prev=r2)
            P.conv_bn_act_bwd(rd, d_idt, dx=dx, dx_acc=dx_acc)
        P.conv_bn_act_bwd(r2, d_a2, dx=d_a1, prev=r1)                                    # a1 feeds conv2 only
        P.conv_bn_act_bwd(r1, d_a1, dx=dx, dx_acc=True)


class ResNetEncoder(nn.Module):
    def __init__(self, name, in_channels=3):
        super().__init__()
        kind, layers, chans = ENCODERS[name]
        block = BasicBlock if kind == "basic" else Bottleneck
        self.out_channels = (in_channels,) + chans
        self.conv1 = ConvP(in_channels, 64, 7, 2, 3)
        self.conv1.needs_dgrad = False            # the image needs no gradient
        self.bn1 = BNP(64)
        inpl = 64
        for li, (planes, nblk, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2)), start=1):
            blocks = []
            for b in range(nblk):
                s = stride if b == 0 else 1
                ds = b == 0 and (s != 1 or inpl != planes * block.expansion)
                blocks.append(block(inpl, planes, s, ds))
                inpl = planes * block.expansion
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        for m in self.modules():                  # torchvision ResNet initialisation
            if isinstance(m, ConvP):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def stages(self):
        return (self.layer1, self.layer2, self.layer3, self.layer4)


class DecoderBlock(nn.Module):
    def __init__(self, in_ch, skip_ch, out_ch, upsample="nearest"):
        super().__init__()
        self.in_ch, self.skip_ch, self.out_ch, self.upsample = in_ch, skip_ch, out_ch, upsample
        self.conv1 = nn.Sequential(ConvP(in_ch + skip_ch, out_ch, 3, 1, 1), BNP(out_ch))
        self.conv2 = nn.Sequential(ConvP(out_ch, out_ch, 3, 1, 1), BNP(out_ch))
        if upsample == "nearest":
            self.conv1[0].up_ca = in_ch       # the first in_ch input channels are nearest_x2(x): engine.Plan.up_frag

    def fwd(self, P, x, skip, lazy_for=None, lazy_up=False):
        """lazy_for: the convolution that is the ONLY consumer of this block's output (the segmentation head behind the last
        block; lazy_up: the next block's conv1 behind its up-sampling, when that block has no skip input), or None: the output's
        BatchNorm + ReLU may then stay unwritten (engine.LazyAct)."""
        ca, cb = x.shape[-1], 0 if skip is None else skip.shape[-1]
        if self.upsample == "bilinear":      # north_star's alternate mode: a stand-alone HBM-bound pass (csrc/bilinear.hip)
            cat = K.upsample2x_bilinear_concat_fwd(x, skip, P.st)
        elif FUSE_UPCAT and K.upcat_fusable(ca, cb, self.conv1[0].cout_p, x.dtype):
            cat = UpCat(x, skip)              # conv1 gathers straight from x (at (iy >> 1, ix >> 1)) and skip
        else:
            cat = K.upsample2x_concat_fwd(x, skip, P.st)
        a1, r1 = P.conv_bn_act(self.conv1[0], self.conv1[1], cat, *RELU, lazy_for=self.conv2[0])   # a1 feeds conv2 only
        out, r2 = P.conv_bn_act(self.conv2[0], self.conv2[1], a1, *RELU, lazy_for=lazy_for, lazy_up=lazy_up)
        if r2 is not None:
            P._producer[id(out)] = r2          # the next block's conv1 may be its only consumer (Plan.conv_bwd, phase form)
        return out, (x, skip, cat, a1, r1, r2)

    @staticmethod
    def relu_outputs(rec):
        """The block's intermediate ReLU outputs held on the tape (tests compare them with the oracle's)."""
        return [rec[3]]

    def bwd(self, P, G, rec, out):
        x, skip, cat, a1, r1, r2 = rec
        d_out = G.pop(out)
        d_a1 = P.like(a1)
        P.conv_bn_act_bwd(r2, d_out, dx=d_a1, prev=r1)             # a1 feeds conv2 only
        dx, dx_acc = G.slot(x)
        if skip is not None:
            ds, ds_acc = G.slot(skip)
        else:
            ds, ds_acc = None, False
        if isinstance(cat, UpCat) and not isinstance(x, LazyAct) and P.up_frag(self.conv1[0], r1[2], x.shape[-1]) is not None:
            # phase form (csrc/conv_up_f32x3.hip): conv1's data gradient lands in dx at x's own resolution -- no gradient of the
            # up-sampled tensor, no 2x2 sum-pool pass
            d_skip = (torch.empty_like(skip) if ds_acc else ds) if skip is not None else None
            P.conv_bn_act_bwd(r1, d_a1, dx=UpGrad(dx, dx_acc, d_skip), prev=None if dx_acc else P._producer.get(id(x)))
            if ds_acc:
                ds.add_(d_skip)
            return
        if isinstance(cat, UpCat):
            n, h, w, ca = x.shape
            d_up = torch.empty((n, 2 * h, 2 * w, ca), device=x.device, dtype=x.dtype)   # gradient of the up-sampled x
            d_skip = torch.empty_like(skip) if ds_acc else ds       # the skip's other consumer comes later in backward
            P.conv_bn_act_bwd(r1, d_a1, dx=(d_up, d_skip))
            if ds_acc:
                ds.add_(d_skip)
            K.upsample2x_concat_bwd(d_up, dx, None, ca, 0, dx_acc, False, P.st)
            return
        d_cat = torch.empty_like(cat)
        P.conv_bn_act_bwd(r1, d_a1, dx=d_cat)
        bwd = K.upsample2x_bilinear_concat_bwd if self.upsample == "bilinear" else K.upsample2x_concat_bwd
        bwd(d_cat, dx, ds, x.shape[-1], 0 if skip is None else skip.shape[-1], dx_acc, ds_acc, P.st)


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=DECODER_CHANNELS, upsample="nearest"):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        in_ch = [enc[0]] + list(decoder_channels[:-1])
        skip_ch = enc[1:] + [0]
        self.blocks = nn.ModuleList(DecoderBlock(i, s, o, upsample) for i, s, o in zip(in_ch, skip_ch, decoder_channels))
        for m in self.modules():                  # smp initialize_decoder
            if isinstance(m, ConvP):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")


class _UnetFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward = kernel plan, backward = the reverse plan."""

    @staticmethod
    def forward(ctx, net, x, want, *params):
        outs, tape = net._forward_plan(x, True, want)
        ctx.net, ctx.tape, ctx.want = net, tape, want
        ctx.set_materialize_grads(False)          # an output the loss never touched arrives as None, not as zeros
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        net, tape = ctx.net, ctx.tape
        if tape is None:
            raise RuntimeError("Unet: backward needs a training-mode forward with grad enabled")
        ctx.tape = None
        net._backward_plan(tape, dict(zip(ctx.want, grads)))   # delivers .grad itself (arena views), see deliver_grads
        return (None, None, None) + (None,) * len(net._param_list)


class Unet(ArenaModule):
    """smp.Unet-shaped factory (same keyword arguments as the reference's call sites)."""

    def __init__(self, encoder_name="resnet34", encoder_depth=5, encoder_weights=None, decoder_use_batchnorm=True,
                 decoder_channels=DECODER_CHANNELS, in_channels=3, classes=1, activation=None,
                 compute_dtype=torch.float32, upsample="nearest", decoder_interpolation=None, **unused):
        """``upsample`` (alias ``decoder_interpolation``, the keyword newer smp releases use): "nearest" -- what the
        reference's traced model does, fused into the decoder convolutions -- or "bilinear" (align_corners=False),
        the mode north_star names."""
        super().__init__()
        upsample = decoder_interpolation or upsample
        if upsample not in ("nearest", "bilinear"):
            raise ValueError(f"upsample must be 'nearest' or 'bilinear', got {upsample!r}")
        self.upsample = upsample
        if encoder_name not in ENCODERS:
            raise ValueError(f"unsupported encoder {encoder_name!r}; available: {sorted(ENCODERS)}")
        if encoder_depth != 5 or tuple(decoder_channels) != DECODER_CHANNELS or not decoder_use_batchnorm or activation:
            raise ValueError("only the reference's configuration is built: depth 5, decoder (256,128,64,32,16), batchnorm, "
                             "no head activation")
        self.name = f"u-{encoder_name}"
        self.classes = classes
        self.in_channels = in_channels
        self.encoder = ResNetEncoder(encoder_name, in_channels)
        self.decoder = UnetDecoder(self.encoder.out_channels, DECODER_CHANNELS, upsample)
        head = ConvP(DECODER_CHANNELS[-1], classes, 3, 1, 1, bias=True)
        nn.init.xavier_uniform_(head.weight)      # smp initialize_head
        nn.init.constant_(head.bias, 0)
        self.segmentation_head = nn.Sequential(head)
        self.grad_ready_hook = None               # set by ddp.GradAllReducer: called with the lowest finished offset
        self.debug_keep_tape = False              # tests: keep the last forward's activations in self._last_tape
        self._last_tape = None
        if isinstance(encoder_weights, str) and encoder_weights not in ("imagenet",):
            self.load_state_dict(torch.load(encoder_weights, map_location="cpu"), strict=False)
        elif encoder_weights == "imagenet":
            warnings.warn("encoder_weights='imagenet' needs a download; offline build keeps the seeded random init "
                          "(pass a state_dict path to load pretrained weights)")
        self.build_arena()
        if compute_dtype != torch.float32:
            self.set_compute_dtype(compute_dtype)

    # ------------------------------------------------------------------------------------------------ forward
    OUTPUTS = ("logits", "decoder", "features")

    def forward(self, x):
        return self.forward_parts(x, ("logits",))

    def forward_parts(self, x, want=("logits",)):
        """One pass, several differentiable results (all ``[N,C,h,w]``-shaped views of NHWC buffers), in ``want`` order:
        ``"logits"`` -- what ``forward`` returns; ``"decoder"`` -- the decoder's 16-channel output before the head
        (``model.decoder(*features)`` upstream, reference ``src/models/uda.py:68``); ``"features"`` -- the deepest encoder
        feature (``model.encoder(x)[-1]``, reference ``src/models/uda.py:64,74-76``).  A single name returns a tensor,
        several a tuple.  The head is only computed when ``"logits"`` is asked for."""
        require_gpu()
        want = tuple(want)
        if not want or any(w not in self.OUTPUTS for w in want) or len(set(want)) != len(want):
            raise ValueError(f"forward_parts: want must be distinct names out of {self.OUTPUTS}, got {want}")
        if x.device.type != "cuda":
            raise RuntimeError("Unet.forward: input must live on the GPU (no CPU path in this build)")
        self.ensure_arena()
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected [N,{self.in_channels},H,W], got {tuple(x.shape)}")
        if x.shape[2] % 32 or x.shape[3] % 32:
            raise RuntimeError(f"Wrong input shape height={x.shape[2]}, width={x.shape[3]}. Expected image height and width "
                               f"divisible by 32.")  # smp's check_input_shape
        if self._padded_input_view(x) is None:
            x = x.float()
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self._param_list):
            return _UnetFunction.apply(self, x, want, *self._param_list)
        with torch.no_grad():                     # inference / validation: no tape, no autograd node
            outs, _ = self._forward_plan(x, False, want)
        return outs[0] if len(outs) == 1 else tuple(outs)

    def _padded_input_view(self, x):
        """``data.prepare_batch`` hands the images over as an [N,3,H,W]-shaped view of the channel-padded NHWC buffer the
        stem convolution reads: recognise it (registered by prepare_batch + layout + dtype) and return that buffer, else
        None (any other tensor, e.g. an RGBA batch sliced [:, :3], is converted by udaseg_nchw_to_nhwc)."""
        cp = self.encoder.conv1.cin_p
        n, c, h, w = x.shape
        es = x.element_size()
        if (is_padded_input(x.data_ptr()) and x.dtype == self.compute_dtype and x.stride() == (h * w * cp, 1, w * cp, cp)
                and x.untyped_storage().nbytes() - es * x.storage_offset() >= es * n * h * w * cp
                and (x.data_ptr() % 16) == 0):
            return x.as_strided((n, h, w, cp), (h * w * cp, w * cp, cp, 1), x.storage_offset())
        return None

    def _forward_plan(self, x, save, want=("logits",)):
        P = Plan(self, self.training, save)
        enc = self.encoder
        tape = []
        x4 = self._padded_input_view(x.detach())
        if x4 is None:
            x4 = K.nchw_to_nhwc(x, enc.conv1.cin_p, P.st, dtype=P.adt)
        f1, r_stem = P.conv_bn_act(enc.conv1, enc.bn1, x4, *RELU)
        pooled, pidx = K.maxpool_fwd(f1, P.st)
        feats = [f1]
        h = pooled
        dblocks = self.decoder.blocks
        nskip = len(enc.out_channels) - 2             # features f1 .. f4 are skip inputs of decoder blocks 3 .. 0

        def prelaunch(fi):
            # feature fi (0 = f1) feeds decoder block nskip - 1 - fi as its skip input: its half of that block's conv1 starts now
            bi = nskip - 1 - fi
            if FUSE_UPCAT and 0 <= bi < len(dblocks) and dblocks[bi].upsample == "nearest" and not isinstance(feats[fi], LazyAct):
                P.prelaunch_skip(dblocks[bi].conv1[0], feats[fi], dblocks[bi].conv1[0].up_ca)
        prelaunch(0)
        for si, stage in enumerate(enc.stages()):
            for blk in stage:
                h, rec = blk.fwd(P, h)
                tape.append((blk, rec, h))
            feats.append(h)
            if si + 1 < nskip:
                prelaunch(si + 1)
        skips = feats[:-1][::-1]                  # f4, f3, f2, f1
        head = self.segmentation_head[0]
        nblk = len(self.decoder.blocks)
        for i, blk in enumerate(self.decoder.blocks):
            skip = skips[i] if i < len(skips) else None
            # the last block's output feeds the head alone unless the caller asked for the decoder features as well
            only, only_up = None, False
            if i == nblk - 1:
                # (fp32 only: the bf16 form needs the consumer's data gradient to make this layer's BatchNorm-backward sums)
                only = head if ("logits" in want and "decoder" not in want and not P.bf16) else None
            elif i + 1 >= len(skips) and self.decoder.blocks[i + 1].upsample != "bilinear" and FUSE_UPCAT and not P.bf16:
                only, only_up = self.decoder.blocks[i + 1].conv1[0], True      # the next block gathers up(h) alone
            h, rec = blk.fwd(P, h, skip, lazy_for=only, lazy_up=only_up)
            tape.append((blk, rec, h))
        logits, d_head = None, None
        if "logits" in want:
            logits, d_head = P.conv(head, h, out_dtype=torch.float32)  # logits stay fp32 (loss accuracy) in every mode
        if self.training:
            self.tick_batchnorm_counters()
        top = feats[-1]
        views = {"logits": lambda: None if logits is None else logits.permute(0, 3, 1, 2)[:, : self.classes],
                 "decoder": lambda: h.permute(0, 3, 1, 2)[:, : DECODER_CHANNELS[-1]],
                 "features": lambda: top.permute(0, 3, 1, 2)[:, : self.encoder.out_channels[-1]]}
        outs = [views[w]() for w in want]
        if not save:
            return outs, None
        tape_all = (P, tape, (r_stem, f1, pooled, pidx), (head, d_head, h), top)
        if self.debug_keep_tape:
            self._last_tape = tape_all
        return outs, tape_all

    # ----------------------------------------------------------------------------------------------- backward
    @staticmethod
    def _seed(G, act, grad):
        """Put an incoming NCHW-shaped gradient of activation ``act`` (NHWC, padded channels) into its gradient slot."""
        buf, acc = G.slot(act)
        c = grad.shape[1]
        if not acc:
            if buf.shape[-1] != c:
                buf.zero_()
            buf.permute(0, 3, 1, 2)[:, :c].copy_(grad)
        else:
            buf.permute(0, 3, 1, 2)[:, :c].add_(grad.to(buf.dtype))

    def _backward_plan(self, tape_all, grads):
        """grads: {"logits" | "decoder" | "features": NCHW-shaped gradient or None}, or the logits gradient itself."""
        if torch.is_tensor(grads):
            grads = {"logits": grads}
        P, tape, (r_stem, f1, pooled, pidx), (head, d_head, h_last), top = tape_all
        dlogits, d_dec, d_top = grads.get("logits"), grads.get("decoder"), grads.get("features")
        if dlogits is None and d_dec is None and d_top is None:
            return
        P.begin_backward()
        G = GradSlots()
        hook = self.grad_ready_hook
        if dlogits is not None:
            n, _, hh, ww = dlogits.shape
            cp = head.cout_p
            # dlogits arrives NCHW-shaped; the CE kernel hands over the padded NHWC buffer as a strided view -- use it as is
            if (dlogits.dtype == torch.float32 and dlogits.stride(1) == 1 and dlogits.stride(3) == cp
                    and dlogits.stride(2) == cp * ww and dlogits.stride(0) == cp * ww * hh and dlogits.storage_offset() == 0
                    and dlogits.untyped_storage().nbytes() >= 4 * n * hh * ww * cp):
                dl = dlogits.as_strided((n, hh, ww, cp), (hh * ww * cp, ww * cp, cp, 1), 0)
            else:
                dl = torch.zeros((n, hh, ww, cp), device=dlogits.device, dtype=torch.float32)
                dl.permute(0, 3, 1, 2)[:, : self.classes].copy_(dlogits)
            dh, _ = G.slot(h_last)
            from .losses import COLSUM_SIDE_TABLE
            dbias = COLSUM_SIDE_TABLE.pop(dl.data_ptr(), None)     # made by ce_bwd in the same pass as dl, when it was
            if P.bf16:
                dl = K.cast_to_bf16(dl, st=P.st)                   # the head's dgrad / wgrad take bf16 operands
            P.conv_bwd(head, d_head, h_last, dl, dx=dh, dx_acc=False, dbias=dbias)
            if hook is not None:
                hook(P, P.offset_of(head))
        if d_dec is not None:
            self._seed(G, h_last, d_dec)
        if d_top is not None:
            self._seed(G, top, d_top)
        for blk, rec, out in reversed(tape):
            if not G.has(out):
                continue                                           # e.g. the whole decoder when only "features" was used
            blk.bwd(P, G, rec, out)
            if hook is not None:
                first = next(m for m in blk.modules() if isinstance(m, ConvP))
                hook(P, P.offset_of(first))
        # stem: maxpool backward accumulates onto the skip gradient of f1, then BN+ReLU and the 7x7 wgrad
        d_pooled = G.pop(pooled)
        d_f1, acc = G.slot(f1)
        K.maxpool_bwd(d_pooled, pidx, d_f1, acc, P.st)
        P.conv_bn_act_bwd(r_stem, d_f1, dx=None)
        if hook is not None:
            hook(P, 0)            # the reducer's stream waits on events of both streams (Plan.ready_events)
        P.join_side_stream()
        self.deliver_grads(P.garena)


def create_model(encoder_name="resnet50", encoder_weights=None, in_channels=3, classes=23):
    """The build's model-creation entry point (mirrors reference ``model_creation_suite``, src/test_system.py:87-101)."""
    return Unet(encoder_name=encoder_name, encoder_weights=encoder_weights, in_channels=in_channels, classes=classes)
