"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/udaseg.h declares, the ctypes
binding covers exactly that set, bad arguments come back as error codes (no exception crosses the boundary), and the product
package never reaches into oracle/."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "uda_aerial_semantic_segmentation_research_amd")


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "udaseg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(udaseg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from uda_aerial_semantic_segmentation_research_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/udaseg.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, set(_lib.SIGNATURES) ^ set(syms)
    assert lib.udaseg_version() >= 100
    assert lib.udaseg_bn_replicas() >= 1 and lib.udaseg_ce_partials() >= 1


def test_error_convention_without_gpu():
    """Argument validation happens before any launch: rc < 0 and a message, never an exception or a crash."""
    from uda_aerial_semantic_segmentation_research_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc(1, 8, 8, 6, 8, 8, 8, 3, 3, 1, 1)            # ci = 6: not a multiple of 4
    rc = lib.udaseg_conv2d_fwd(ctypes.byref(d), 16, 16, None, 16, 0, 0.0, 0, None)
    assert rc == -1 and b"multiples of 4" in lib.udaseg_last_error()
    rc = lib.udaseg_conv2d_fwd(None, 16, 16, None, 16, 0, 0.0, 0, None)
    assert rc == -1
    rc = lib.udaseg_ce_fwd(16, 16, 100, 23, 23, 16, 16, 16, None)  # ldc not a multiple of 4
    assert rc == -1
    assert lib.udaseg_adam_flat(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.001, None) == -1
    d2 = _lib.ConvDesc(2, 16, 16, 64, 16, 16, 64, 3, 3, 1, 1)
    assert lib.udaseg_conv_flops(ctypes.byref(d2)) == 2.0 * 2 * 16 * 16 * 64 * 64 * 9


def test_modules_fail_loudly_on_cpu_tensors():
    import torch
    from uda_aerial_semantic_segmentation_research_amd.losses import AdversarialLoss, CrossEntropyLoss
    with pytest.raises(RuntimeError, match="no CPU path"):
        CrossEntropyLoss()(torch.zeros(1, 23, 32, 32), torch.zeros(1, 32, 32, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU path"):
        AdversarialLoss().discriminator_loss(torch.rand(4, 1), torch.rand(4, 1))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
                assert "/root/reference" not in src


def test_state_dict_schema_matches_oracle_and_reference_keys():
    """Host logic only (no kernels): smp / reference key schema and shapes, arena views, load round trip."""
    import torch
    from oracle.adversarial_ref import DomainDiscriminatorRef
    from oracle.unet_ref import UnetRef
    from uda_aerial_semantic_segmentation_research_amd.discriminator import DomainDiscriminator
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    for name in ("resnet18", "resnet50"):
        ref, net = UnetRef(name, classes=23), Unet(name, encoder_weights=None, in_channels=3, classes=23)
        sa, sb = ref.state_dict(), net.state_dict()
        assert list(sa) == list(sb)
        assert all(sa[k].shape == sb[k].shape for k in sa)
        net.load_state_dict(sa)
        assert all(torch.equal(sa[k], net.state_dict()[k]) for k in sa)
        assert net._arena_ok()
        w = net.phys_weight(net.encoder.conv1)                      # [64,7,7,4] OHWI, padded input channel stays zero
        assert tuple(w.shape) == (64, 7, 7, 4) and float(w[..., 3].abs().max()) == 0.0
    d, r = DomainDiscriminator(), DomainDiscriminatorRef()
    assert list(d.state_dict()) == list(r.state_dict())
    assert sum(p.numel() for p in d.parameters()) == 2758849


def test_bn_bindings_refuse_mismatched_operands_before_any_launch():
    """The element-wise C-ABI entry points take one (pixels, c) extent and raw pointers: an operand of another dtype or size
    would be read or written past its end by the kernel (round 2's tools/bn_bandwidth.py run ended in a GPU memory fault, not
    an error code: profiles/r02_bn_bandwidth.txt).  The binding refuses such operands with ValueError -- checked here on CPU
    tensors, i.e. strictly before the library is called."""
    import torch
    from uda_aerial_semantic_segmentation_research_amd import kernels as K
    R = K.bn_replicas()
    c, px = 16, 64
    bf, f32 = torch.bfloat16, torch.float32
    y = torch.zeros(px, c, dtype=bf)
    vec = lambda: torch.zeros(c)
    sums = torch.zeros(2 * c * R, dtype=torch.float64)
    good = dict(y=y, sums=sums, gamma=vec(), beta=vec(), residual=None, z=torch.zeros(px, c, dtype=bf), eps=1e-5, momentum=0.1,
                running_mean=vec(), running_var=vec(), save_mean=vec(), save_rstd=vec(), act=1, slope=0.0)
    for bad in (dict(z=torch.zeros(px, c, dtype=f32)),                 # fp32 output for a bf16 launch: written at half its size
                dict(z=torch.zeros(px // 2, c, dtype=bf)),              # half the pixels
                dict(residual=torch.zeros(px, c, dtype=f32)),
                dict(z=torch.zeros(px, 2 * c, dtype=bf)[:, :c]),        # not contiguous
                dict(gamma=torch.zeros(c // 2)), dict(save_mean=torch.zeros(c, dtype=torch.float64)),
                dict(sums=torch.zeros(2 * c, dtype=torch.float64)),     # one replica instead of R
                dict(sums=torch.zeros(2 * c * R))):                     # fp32 accumulators
        with pytest.raises(ValueError):
            K.bn_apply(**{**good, **bad})
    dz, z = torch.zeros(px, c, dtype=bf), torch.zeros(px, c, dtype=bf)
    with pytest.raises(ValueError):
        K.bn_bwd_reduce(dz.float(), z, y, vec(), vec(), sums, 1, 0.0)
    with pytest.raises(ValueError):
        K.bn_bwd_reduce(dz, z[: px // 2], y, vec(), vec(), sums, 1, 0.0)
    with pytest.raises(ValueError):
        K.bn_bwd_apply(dz, z, y, vec(), vec(), vec(), sums, torch.zeros(px, c, dtype=f32), None, vec(), vec(), 1, 0.0)
    with pytest.raises(ValueError):
        K.bn_bwd_apply(dz, z, y, vec(), vec(), vec(), sums[: c], dz, None, vec(), vec(), 1, 0.0)
