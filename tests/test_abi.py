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


def _scalars_for(entry, roles, ConvDesc):
    """One valid-looking set of scalar arguments per entry point (tensor slots left as None)."""
    special = {"up_ca": 0, "ca": 8, "split": 0, "src_c": 8, "c_off": 0, "up": 0, "out_f32": 0, "out_bf16": 1, "bf16_": 1, "cpad": 8,
               "classes": 5, "ldc": 8, "pooled": 0, "entries": 3, "blocks": 4, "bytes": 64, "scratch_bytes": 256, "count": 48}
    args = []
    for r in roles:
        if r[0] == "desc":
            args.append(ConvDesc(2, 8, 8, 16, 8, 8, 16, 3, 3, 1, 1))
        elif r[0] == "int":
            args.append(special.get(r[1], 4))
        elif r[0] == "float":
            args.append(0.5)
        elif r[0] == "stream":
            args.append(0)               # the NULL stream (None would ask torch for its current stream: no GPU here)
        else:
            args.append(None)            # tensors, host pointers
    return args


def test_every_entry_point_checks_its_operands_before_the_library_is_called():
    """Walks EVERY name of _lib.SIGNATURES: it has a row in the operand table whose argument kinds match the ctypes signature,
    and for every tensor role of every pointer-passing entry point a wrong dtype, a short buffer, a non-contiguous tensor and a
    missing required operand raise ValueError while the (stubbed) library function is never reached; the well-formed call reaches
    it exactly once with the tensors' addresses.  CPU tensors: the device rule is switched off for this test and tested by
    itself at the end."""
    import torch
    from uda_aerial_semantic_segmentation_research_amd import _lib, _operands as O
    kinds = {"desc": (_lib._D,), "stream": (ctypes.c_void_p,), "tensor": (ctypes.c_void_p,),
             "int": (ctypes.c_int, ctypes.c_int64, ctypes.c_size_t), "float": (ctypes.c_float,)}
    assert sorted(O.OPERANDS) == sorted(_lib.SIGNATURES)
    other = {torch.float32: torch.bfloat16, torch.bfloat16: torch.float32, torch.float64: torch.float32, torch.int64: torch.int32,
             torch.int32: torch.int64, torch.uint8: torch.int32}
    calls = []
    n_roles = n_entries = 0
    O.set_require_cuda(False)
    saved = dict(O._FN)
    try:
        for entry, (res, argtypes) in _lib.SIGNATURES.items():
            roles = O.OPERANDS[entry]
            assert len(roles) == len(argtypes), entry
            for r, a in zip(roles, argtypes):
                assert (a in kinds[r[0]]) if r[0] != "host" else (a is ctypes.c_void_p or hasattr(a, "_type_")), (entry, r, a)
            tens = [i for i, r in enumerate(roles) if r[0] == "tensor"]
            if not tens:
                continue
            n_entries += 1
            O._FN[entry] = lambda *a, _e=entry: calls.append((_e, a)) or 0
            base = _scalars_for(entry, roles, _lib.ConvDesc)
            req = {nm: (dt, cnt, opt) for nm, dt, cnt, opt in O.requirements(entry, *base)}

            def make(i, dtype=None, count=None):
                nm = roles[i][1]
                dt, cnt, _ = req[nm]
                dt = torch.float32 if dt == "raw32" else dt
                return torch.zeros(max(cnt if count is None else count, 0), dtype=dtype or dt)

            good = list(base)
            for i in tens:
                good[i] = make(i)
            fn = getattr(O.ops, entry)
            calls.clear()
            assert fn(*good) == 0 and len(calls) == 1, entry
            sent = calls[0][1]
            for i in tens:
                assert sent[i] == good[i].data_ptr(), (entry, roles[i][1])
            for i in tens:
                nm = roles[i][1]
                dt, cnt, opt = req[nm]
                assert cnt >= 1, (entry, nm, "the test's scalars must give every operand a positive extent")
                n_roles += 1
                trials = [make(i, count=cnt - 1)]                                  # short by one element
                if dt != "raw32":
                    trials.append(make(i, dtype=other[dt]))                         # same element count, other dtype
                trials.append(torch.zeros(2 * cnt + 2, dtype=torch.float32 if dt == "raw32" else dt)[::2])      # not contiguous
                trials.append("not a tensor")
                if not opt:
                    trials.append(None)
                for t in trials:
                    bad = list(good)
                    bad[i] = t
                    calls.clear()
                    with pytest.raises(ValueError):
                        fn(*bad)
                    assert not calls, (entry, nm, "the library was reached with a bad operand")
                if opt:                                                             # NULL is legal there
                    okn = list(good)
                    okn[i] = None
                    calls.clear()
                    assert fn(*okn) == 0 and calls[0][1][i] is None
        assert n_entries >= 85 and n_roles >= 390, (n_entries, n_roles)
        # the device rule: CPU tensors are refused when it is on (what the product runs with)
        O.set_require_cuda(True)
        d = _lib.ConvDesc(2, 8, 8, 16, 8, 8, 16, 3, 3, 1, 1)
        O._FN["udaseg_conv2d_dgrad"] = lambda *a: calls.append(a) or 0
        calls.clear()
        with pytest.raises(ValueError, match="GPU"):
            O.ops.udaseg_conv2d_dgrad(d, torch.zeros(2048), torch.zeros(2304), torch.zeros(2048), 0, None)
        assert not calls
    finally:
        O.set_require_cuda(True)
        O._FN.clear()
        O._FN.update(saved)


def test_python_option_mirror_matches_the_library_table():
    """_lib.OPTIONS (the one Python mirror of the switchboard) against udaseg_option_count / udaseg_option_name; set / get / epoch
    work without a GPU; the header declares every key."""
    import re
    from uda_aerial_semantic_segmentation_research_amd import _lib
    lib = _lib.load()
    n = lib.udaseg_option_count()
    assert sorted(_lib.OPTIONS.values()) == list(range(n))
    hdr = open(os.path.join(ROOT, "include", "udaseg.h")).read()
    keys = dict(re.findall(r"#define UDASEG_OPT_(\w+) (\d+)", hdr))
    assert int(keys.pop("COUNT")) == n
    assert {k: int(v) for k, v in keys.items()} == _lib.OPTIONS
    for name, key in _lib.OPTIONS.items():
        env = lib.udaseg_option_name(key).decode()
        assert env.startswith("UDASEG_"), (name, env)
    e0 = lib.udaseg_option_epoch()
    before = lib.udaseg_get_option(_lib.OPTIONS["F3_CFG"])
    assert lib.udaseg_set_option(_lib.OPTIONS["F3_CFG"], 5) == 0 and lib.udaseg_get_option(_lib.OPTIONS["F3_CFG"]) == 5
    assert lib.udaseg_set_option(_lib.OPTIONS["F3_CFG"], -1) == 0 and lib.udaseg_get_option(_lib.OPTIONS["F3_CFG"]) == before
    assert lib.udaseg_option_epoch() == e0 + 2
    assert lib.udaseg_set_option(n, 1) != 0 and lib.udaseg_set_option(0, -2) != 0
    # one key for both gather loops, as before the table existed
    lib.udaseg_set_option(_lib.OPTIONS["GENERIC_GATHER"], 1)
    assert lib.udaseg_get_option(_lib.OPTIONS["WGRAD_GENERIC"]) == 1
    lib.udaseg_set_option(_lib.OPTIONS["GENERIC_GATHER"], -1)
    assert lib.udaseg_get_option(_lib.OPTIONS["WGRAD_GENERIC"]) == 0
    # the library reads the environment in ONE place
    import glob
    csrc = os.path.join(ROOT, "uda_aerial_semantic_segmentation_research_amd", "csrc")
    sites = [f for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) if "getenv(" in open(f).read()]
    assert [os.path.basename(f) for f in sites] == ["api.hip"], sites
