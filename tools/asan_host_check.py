#!/usr/bin/env python
"""Drive the HOST side of libudaseg_hip through its AddressSanitizer + UBSan build (make -C <pkg>/csrc asan).

    LD_PRELOAD=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so ...) ASAN_OPTIONS=detect_leaks=0 \\
        UDASEG_LIB=<pkg>/libudaseg_hip_asan.so python tools/asan_host_check.py

CPU box only.  Every entry point that builds a launch description is called with well-formed and malformed geometry: the
argument checks, the tap / parity-class tables of the implicit GEMM, the K-slice and split-K plans, the fused
upsample+concat and split-output descriptions all run on the host before the first HIP call, which then fails for want of
a device -- so each call must come back with an error CODE (BADARG / UNSUPPORTED / HIP), never crash, and the sanitizers
must stay silent.  Prints "asan host check ok: N calls"."""
import ctypes as C
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from uda_aerial_semantic_segmentation_research_amd import _lib
    assert "asan" in os.path.basename(_lib.LIB_PATH), f"UDASEG_LIB must point at the sanitizer build, got {_lib.LIB_PATH}"
    lib = _lib.load()
    P = 4096          # any non-null "device pointer": nothing is dereferenced on the host
    calls = 0
    ok_codes = {-1, -2, -3, -4}

    def desc(n, hi, wi, ci, co, k, s, p):
        ho, wo = (hi + 2 * p - k) // s + 1, (wi + 2 * p - k) // s + 1
        return _lib.ConvDesc(n, hi, wi, ci, ho, wo, co, k, k, s, p)

    geoms = []
    for n, hw, ci, co, (k, s, p) in itertools.product((1, 8), (8, 32, 512), (4, 16, 32, 64, 192, 2048), (4, 16, 24, 64, 256),
                                                      ((1, 1, 0), (1, 2, 0), (3, 1, 1), (3, 2, 1), (4, 2, 1), (7, 2, 3), (8, 4, 2))):
        if n * hw * hw * max(ci, co) < (1 << 31):
            geoms.append(desc(n, hw, hw, ci, co, k, s, p))
    geoms += [desc(3, 9, 7, 8, 36, 3, 1, 1), desc(1, 5, 5, 8, 8, 3, 2, 1), desc(2, 40, 20, 16, 16, 3, 1, 1),
              _lib.ConvDesc(1, 8, 8, 6, 8, 8, 8, 3, 3, 1, 1), _lib.ConvDesc(1, 8, 8, 8, 9, 9, 8, 3, 3, 1, 1),
              _lib.ConvDesc(0, 8, 8, 8, 8, 8, 8, 3, 3, 1, 1), _lib.ConvDesc(1, 8, 8, 8, 8, 8, 8, 9, 9, 1, 4),
              _lib.ConvDesc(1, 8, 8, 8, 2, 2, 8, 3, 3, 5, 1)]
    for d in geoms:
        r = C.byref(d)
        for rc in (lib.udaseg_conv2d_fwd(r, P, P, None, P, 0, 0.0, 0, None),
                   lib.udaseg_conv2d_fwd(r, P, P, P, P, 1, 0.2, 1, None),
                   lib.udaseg_conv2d_fwd_bnstats(r, P, P, None, P, P, None),
                   lib.udaseg_conv2d_fwd_fused(r, P, P, P, P, P, 1, 0.0, None),
                   lib.udaseg_conv2d_fwd_bf16(r, P, P, None, None, P, 0, 0, 0.0, P, None),
                   lib.udaseg_conv2d_dgrad(r, P, P, P, 0, None), lib.udaseg_conv2d_dgrad(r, P, P, P, 1, None),
                   lib.udaseg_conv2d_dgrad_bf16(r, P, P, P, 0, None),
                   lib.udaseg_conv2d_wgrad(r, P, P, P, 0, None), lib.udaseg_conv2d_wgrad(r, P, P, P, 1, None),
                   lib.udaseg_conv2d_wgrad_bf16(r, P, P, P, 1, None),
                   lib.udaseg_pack_dgrad_weights(r, P, P, None)):
            assert rc in ok_codes, (rc, [getattr(d, f) for f, _ in d._fields_])
            calls += 1
        assert lib.udaseg_conv_flops(r) >= 0 and lib.udaseg_workspace_bytes(r) >= 0
        ok = lib.udaseg_conv2d_dgrad_bnreduce_ok(r)
        rc = lib.udaseg_conv2d_dgrad_bnreduce(r, P, P, P, P, P, P, P, P, 1, 0.0, P, None)
        assert ok in (0, 1) and rc in ok_codes and (ok == 1 or rc == -2 or rc == -1), (ok, rc)
        calls += 2
        # fused decoder input / split output / gradient slices: every split of the input channels
        if d.stride == 1 and d.kh == 3:
            for ca in sorted({d.ci, d.ci // 2, d.ci // 4 * 3, 64, 32, 8, 0, -4}):
                skip = None if ca == d.ci else P
                for rc in (lib.udaseg_conv2d_fwd_upcat(r, P, skip, ca, P, None, P, 0, 0.0, P, None),
                           lib.udaseg_conv2d_fwd_upcat_bf16(r, P, skip, ca, P, P, P, 1, 0.0, None, None),
                           lib.udaseg_conv2d_dgrad_split(r, P, P, P, P, ca, None),
                           lib.udaseg_conv2d_dgrad_split_bf16(r, P, P, P, P, ca, None),
                           lib.udaseg_conv2d_wgrad_part(r, P, max(ca, 4), 0, 1, P, P, 1, None),
                           lib.udaseg_conv2d_wgrad_part(r, P, max(d.ci - ca, 4), max(ca, 0), 0, P, P, 1, None),
                           lib.udaseg_conv2d_wgrad_part_bf16(r, P, max(ca, 8), 0, 1, P, P, 0, None)):
                    assert rc in ok_codes, (rc, ca, [getattr(d, f) for f, _ in d._fields_])
                    calls += 1
                # halo-resident kernels (bf16 and the fp32 three-term split): support queries and launch descriptions
                upc = ca if 0 < ca <= d.ci else 0
                sk = P if 0 < upc < d.ci else None
                split = ca if 0 < ca < d.ci else 0
                for q in (lib.udaseg_conv_frag_ok(r, 0, upc), lib.udaseg_conv_frag_preferred(r, 1, 0),
                          lib.udaseg_conv_f32x3_ok(r, 0, upc), lib.udaseg_conv_f32x3_preferred(r, 1, 0),
                          lib.udaseg_conv2d_wgrad_halo_bf16_ok(r, upc), lib.udaseg_conv2d_wgrad_halo_f32x3_ok(r, upc)):
                    assert q in (0, 1), q
                    calls += 1
                for rc in (lib.udaseg_conv2d_fwd_frag_bf16(r, P, sk, upc, P, None, None, None, 0, 0.0, P, 0, 0, 0.0, P, None),
                           lib.udaseg_conv2d_dgrad_frag_bf16(r, P, P, P, P if split else None, split, None, None, None, None, None,
                                                             0, 0.0, None, 0, None),
                           lib.udaseg_conv2d_fwd_f32x3(r, P, sk, upc, P, None, P, 0, 0.0, P, None),
                           lib.udaseg_conv2d_dgrad_f32x3(r, P, P, P, P if split else None, split, None, None, None, None, None,
                                                         0, 0.0, None, 0, None),
                           lib.udaseg_conv2d_dgrad_f32x3(r, P, P, P, None, 0, P, P, P, P, P, 1, 0.0, P, 0, None),
                           lib.udaseg_conv2d_wgrad_halo_bf16(r, P, sk, upc if sk else 0, P, P, None),
                           lib.udaseg_conv2d_wgrad_halo_f32x3(r, P, sk, upc if sk else 0, P, P, None)):
                    assert rc in ok_codes, (rc, ca, [getattr(d, f) for f, _ in d._fields_])
                    calls += 1
    for rc in (lib.udaseg_bn_stats_bf16(P, 16, 64, P, None), lib.udaseg_bn_stats_bf16(P, 16, 60, P, None), lib.udaseg_bn_stats_bf16(None, 16, 64, P, None)):
        assert rc in ok_codes, rc
        calls += 1
    # element-wise / loss / optimizer entry points: shape validation
    for c in (0, 3, 4, 24, 4096, 4100):
        for pixels in (0, 1, 1 << 20):
            for rc in (lib.udaseg_bn_stats(P, pixels, c, P, None),
                       lib.udaseg_bn_apply(P, P, P, P, None, P, pixels, c, 1e-5, 0.1, P, P, P, P, 1, 0.0, None),
                       lib.udaseg_bn_bwd_reduce(P, None, P, P, P, P, P, pixels, c, P, 1, 0.0, None),
                       lib.udaseg_bn_bwd_apply(P, None, P, P, P, P, P, P, P, None, P, P, pixels, c, 1, 0.0, 0, 0, 0, None),
                       lib.udaseg_channel_sum(P, pixels, c, P, 0, None),
                       lib.udaseg_ce_fwd(P, P, pixels, 23, c, P, P, P, None),
                       lib.udaseg_ce_bwd(P, P, P, None, pixels, 23, c, P, P, P, None),
                       lib.udaseg_argmax_confusion(P, P, pixels, 23, c, P, None, None),
                       lib.udaseg_upsample2x_concat_fwd(P, P, P, 1, 4, 4, c, c, None),
                       lib.udaseg_upsample2x_bilinear_concat_fwd(P, None, P, 1, 4, 4, c, 0, 0, None),
                       lib.udaseg_upsample2x_bilinear_concat_bwd(P, P, P, 1, 4, 4, c, c, 0, 1, 1, None),
                       lib.udaseg_adam_flat(P, P, P, P, pixels, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.001, None)):
                assert rc in ok_codes or rc == 0, rc        # zero-size launches may legitimately return OK
                calls += 1
    # NULL sweep over the operand table (uda_aerial_semantic_segmentation_research_amd/_operands.py): for EVERY entry point that takes
    # device pointers, each required pointer in turn is NULL -- the C side must answer BADARG (-1) before any launch description is
    # built -- and the all-valid call must come back with an error CODE (no device on this box), never crash.  What the C side can
    # check without extents: NULL-ness, channel granules (c % 4 / c % 8), enum ranges; extents are the binding's job (the table).
    from uda_aerial_semantic_segmentation_research_amd import _operands as O
    special = {"up_ca": 0, "ca": 8, "split": 0, "src_c": 8, "c_off": 0, "up": 0, "out_f32": 0, "out_bf16": 1, "bf16_": 1, "cpad": 8,
               "classes": 5, "ldc": 8, "pooled": 0, "entries": 3, "blocks": 4, "bytes": 64, "scratch_bytes": 256, "count": 48, "c": 16}
    missing = []
    null_ok = {"udaseg_set_workspace", "udaseg_set_stats_scratch", "udaseg_debug_set_timeline"}      # NULL un-binds there
    for entry, roles in O.OPERANDS.items():
        tens = [i for i, r in enumerate(roles) if r[0] == "tensor"]
        if not tens or entry in null_ok:
            continue
        base = []
        for r in roles:
            base.append(C.byref(desc(2, 8, 8, 16, 16, 3, 1, 1)) if r[0] == "desc" else special.get(r[1], 4) if r[0] == "int"
                        else 0.5 if r[0] == "float" else None if r[0] == "stream" else (C.c_float * 3)(1, 1, 1) if r[0] == "host" else P)
        fn = getattr(lib, entry)
        rc = fn(*base)
        assert rc in ok_codes, (entry, rc)
        calls += 1
        for i in tens:
            if roles[i][4]:
                continue                                      # optional operand: NULL is a legal value
            a = list(base)
            a[i] = None
            rc = fn(*a)
            if rc not in (-1, -2):                            # BADARG, or UNSUPPORTED geometry found first: both precede any launch
                missing.append((entry, roles[i][1], rc))
            calls += 1
    assert not missing, f"required pointers the C side does not check for NULL: {missing}"
    assert lib.udaseg_set_workspace(P, 1 << 20) in ok_codes          # no current device on this box
    assert lib.udaseg_debug_set_timeline(None, 0) == 0 and lib.udaseg_debug_set_timeline(P, 16) == 0
    d = desc(8, 128, 128, 64, 64, 3, 1, 1)
    assert lib.udaseg_conv2d_fwd(C.byref(d), P, P, None, P, 0, 0.0, 0, None) in ok_codes      # with the timeline hook armed
    assert lib.udaseg_debug_set_timeline(None, 0) == 0
    assert lib.udaseg_last_error() is not None
    print(f"asan host check ok: {calls} calls")


if __name__ == "__main__":
    main()
