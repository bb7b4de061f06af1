"""TEST INFRASTRUCTURE (oracle): numpy restatement of the exact three-term bf16 split and of the six-product evaluation that
csrc/conv_halo_f32x3.hip / conv_wgrad_halo_f32x3_kernel use for the fp32 convolutions of reference smp.Unet (src/models/train.py:341,
343 reach torch.nn.functional.conv2d and its autograd; there the products are plain fp32).  Only tests/ may import this.

    x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (round to nearest even, exact subtractions)
    a * b ~= sum_{i + j <= 2} a_i * b_j ;   what is left out is a1 b2 + a2 b1 + a2 b2, at most 2^-23 |a b| (1 + 2^-8)
"""
import numpy as np


def bf16_round(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32.  Finite inputs."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    """The three bf16 terms of every fp32 value (each returned as fp32): x == t0 + t1 + t2 exactly."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    t0 = bf16_round(x)
    r1 = (x - t0).astype(np.float32)
    t1 = bf16_round(r1)
    t2 = bf16_round((r1 - t1).astype(np.float32))
    return t0, t1, t2


def six_products(a, b):
    """sum over i + j <= 2 of a_i * b_j in float64 (the products themselves are exact in fp32: 8 x 8 significand bits)."""
    sa, sb = split3(a), split3(b)
    out = np.zeros(np.broadcast(a, b).shape, dtype=np.float64)
    for i in range(3):
        for j in range(3 - i):
            out += sa[i].astype(np.float64) * sb[j].astype(np.float64)
    return out


def bf16_bits(t):
    """The 16 stored bits of bf16-representable fp32 values."""
    return (np.ascontiguousarray(t, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def negated_groups(nk16):
    """The (16-channel chunk, dx) groups G = 3 * chunk + dx of a 3x3 layer's K loop that csrc/conv_halo_f32x3.hip multiplies with
    negated weights on a negated accumulator (csrc/halo_common.h::f3_negated_groups: the bf16 MFMA adder truncates toward minus
    infinity; + - - + over the quarter points cancels most of that bias).  Returns range(q1, q3)."""
    ng = 3 * nk16
    q1 = (ng + 2) // 4
    return range(q1, ng - q1)
