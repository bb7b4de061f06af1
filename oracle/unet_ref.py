"""CPU oracle: pure-torch restatement of the encoder-decoder the reference builds.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this file;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only
as the checker / the reported CPU baseline.

What it restates
----------------
The reference never defines the network: it calls the third-party factory
``smp.Unet(encoder_name=..., encoder_weights=..., in_channels=..., classes=...)``
(reference ``src/test_system.py:90-95``, ``src/models/train.py:572-577``,
``src/models/uda.py:42-48``).  ``segmentation-models-pytorch>=0.3.0`` and
``torchvision>=0.15.0`` (reference ``requirements.txt:2-3``, unpinned) are absent
offline; they only *compose* ``torch.nn`` layers, so this file composes the same
layers from ``torch.nn`` directly:

* encoder  = torchvision ``ResNet`` trunk (``BasicBlock`` for resnet18/34,
  ``Bottleneck`` for resnet50, stride on the 3x3), features taken after
  ``relu``, ``layer1`` .. ``layer4``;
* decoder  = smp ``UnetDecoder`` with channels (256,128,64,32,16): per block
  nearest x2 -> cat([up, skip], 1) -> (conv3x3 no-bias, BN, ReLU) x2;
* head     = ``Conv2d(16, classes, 3, padding=1)`` with bias, no activation.

Pinned by: ``tests/golden/unet_r50_trace.json`` -- the op list / shapes / conv,
BN and pool hyper-parameters decoded (``oracle/decode_trace.py``) from the
``add_graph`` trace the reference committed in ``test_logs/*/events.out.tfevents.*``
(written by ``src/visualization/tensorboard_logger.py:79-83``).  ``tests/test_oracle_unet.py``
checks ``UnetRef('resnet50', classes=23)`` against it op by op at 1x3x256x256.
Numerically the reference's own tests pin nothing for this path (shape/range
asserts only: ``src/test_system.py:90-97,209-215``), so numeric parity is defined
against this restatement running torch's CPU kernels.

state_dict keys follow smp naming so reference checkpoints
(``src/models/train.py:491-500``) round-trip.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

ENCODERS = {
    # name: (block kind, layers, feature channels after [stem, layer1..4])
    "resnet18": ("basic", (2, 2, 2, 2), (64, 64, 128, 256, 512)),
    "resnet34": ("basic", (3, 4, 6, 3), (64, 64, 128, 256, 512)),
    "resnet50": ("bottleneck", (3, 4, 6, 3), (64, 256, 512, 1024, 2048)),
}
DECODER_CHANNELS = (256, 128, 64, 32, 16)


class _Tracer:
    """Optional op log so the structure can be compared with the reference's traced graph."""

    def __init__(self):
        self.ops = []

    def add(self, op, out, **kw):
        e = {"op": op, "out": list(out.shape)}
        e.update(kw)
        self.ops.append(e)


_TR = None  # module-level tracer handle, set by UnetRef.trace()


def _conv(x, m):
    y = m(x)
    if _TR is not None:
        _TR.add("_convolution", y, **{"in": list(x.shape), "bias": m.bias is not None, "stride": list(m.stride),
                                       "padding": list(m.padding), "dilation": list(m.dilation), "groups": m.groups,
                                       "kernel": list(m.kernel_size)})
    return y


def _bn(x, m):
    y = m(x)
    if _TR is not None:
        _TR.add("batch_norm", y, momentum=m.momentum, eps=m.eps)
    return y


def _relu(x):
    y = F.relu(x, inplace=False)
    if _TR is not None:
        _TR.add("relu_", y)
    return y


def _max_pool(x):
    # torchvision ResNet.maxpool = MaxPool2d(3, 2, 1); a function of its own so tests/_parity.py can make both paths pick the same
    # window winners (like _relu's mask)
    return F.max_pool2d(x, 3, 2, 1)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        out = _relu(_bn(_conv(x, self.conv1), self.bn1))
        out = _bn(_conv(out, self.conv2), self.bn2)
        idt = x
        if self.downsample is not None:
            idt = _bn(_conv(x, self.downsample[0]), self.downsample[1])
        out = out + idt
        if _TR is not None:
            _TR.add("add_", out)
        return _relu(out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)  # stride on the 3x3 (trace: layer2.0.conv2)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        out = _relu(_bn(_conv(x, self.conv1), self.bn1))
        out = _relu(_bn(_conv(out, self.conv2), self.bn2))
        out = _bn(_conv(out, self.conv3), self.bn3)
        idt = x
        if self.downsample is not None:
            idt = _bn(_conv(x, self.downsample[0]), self.downsample[1])
        out = out + idt
        if _TR is not None:
            _TR.add("add_", out)
        return _relu(out)


class ResNetEncoderRef(nn.Module):
    def __init__(self, name, in_channels=3):
        super().__init__()
        kind, layers, chans = ENCODERS[name]
        block = BasicBlock if kind == "basic" else Bottleneck
        self.out_channels = (in_channels,) + chans
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        # torchvision ResNet.__init__ initialisation
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        feats = [x]
        x = _relu(_bn(_conv(x, self.conv1), self.bn1))
        feats.append(x)
        x = _max_pool(x)
        if _TR is not None:
            _TR.add("max_pool2d", x, kernel=[3, 3], stride=[2, 2], padding=[1, 1], dilation=[1, 1], ceil_mode=0)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
            feats.append(x)
        return feats


class DecoderBlockRef(nn.Module):
    def __init__(self, in_ch, skip_ch, out_ch, upsample="nearest"):
        super().__init__()
        self.upsample = upsample
        self.conv1 = nn.Sequential(nn.Conv2d(in_ch + skip_ch, out_ch, 3, padding=1, bias=False),
                                   nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(nn.Conv2d(out_ch, out_ch, 3, padding=1, bias=False),
                                   nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True))

    def forward(self, x, skip=None):
        if self.upsample == "nearest":
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            if _TR is not None:
                _TR.add("upsample_nearest2d", x)
        else:  # north_star's named alternate
            x = F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False)
            if _TR is not None:
                _TR.add("upsample_bilinear2d", x)
        if skip is not None:
            a, b = x, skip
            x = torch.cat([x, skip], dim=1)
            if _TR is not None:
                _TR.add("cat", x, in_shapes=[list(a.shape), list(b.shape)], dim=1)
        x = _relu(_bn(_conv(x, self.conv1[0]), self.conv1[1]))
        x = _relu(_bn(_conv(x, self.conv2[0]), self.conv2[1]))
        return x


class UnetDecoderRef(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=DECODER_CHANNELS, upsample="nearest"):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]          # drop the image, deepest first
        head = enc[0]
        in_ch = [head] + list(decoder_channels[:-1])
        skip_ch = enc[1:] + [0]
        self.blocks = nn.ModuleList(DecoderBlockRef(i, s, o, upsample)
                                    for i, s, o in zip(in_ch, skip_ch, decoder_channels))
        # smp.base.initialization.initialize_decoder
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, *features):
        feats = features[1:][::-1]
        x = feats[0]
        skips = feats[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class UnetRef(nn.Module):
    """``smp.Unet``-shaped factory result (same kwargs as reference ``src/test_system.py:90-95``)."""

    def __init__(self, encoder_name="resnet50", encoder_weights=None, in_channels=3, classes=23, upsample="nearest"):
        super().__init__()
        if encoder_weights not in (None,):
            raise ValueError("offline build: encoder_weights must be None (load a state_dict instead)")
        self.encoder = ResNetEncoderRef(encoder_name, in_channels)
        self.decoder = UnetDecoderRef(self.encoder.out_channels, DECODER_CHANNELS, upsample)
        self.segmentation_head = nn.Sequential(nn.Conv2d(DECODER_CHANNELS[-1], classes, 3, padding=1))
        # smp.base.initialization.initialize_head
        nn.init.xavier_uniform_(self.segmentation_head[0].weight)
        nn.init.constant_(self.segmentation_head[0].bias, 0)

    def forward(self, x):
        feats = self.encoder(x)
        d = self.decoder(*feats)
        return _conv(d, self.segmentation_head[0])

    @torch.no_grad()
    def trace(self, x):
        """Run once in eval mode and return the op list in the fixture's vocabulary."""
        global _TR
        was = self.training
        self.eval()
        _TR = _Tracer()
        try:
            self.forward(x)
            return _TR.ops
        finally:
            _TR = None
            self.train(was)


def conv_flops_fwd(model, n, h, w):
    """Forward conv FLOPs (multiply-add = 2) of ``model`` on an n x C x h x w batch."""
    ops = model.trace(torch.zeros(1, model.encoder.conv1.in_channels, h, w))
    tot = 0
    for o in ops:
        if o["op"] == "_convolution":
            _, cout, ho, wo = o["out"]
            tot += 2 * cout * ho * wo * o["in"][1] * o["kernel"][0] * o["kernel"][1]
    return tot * n
