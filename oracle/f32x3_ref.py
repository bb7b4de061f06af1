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


# ---- round 5: conv3x3(nearest_x2(a)) as four 2x2 phase convolutions of a (csrc/conv_up_f32x3.hip) -------------------------------
# smp's DecoderBlock (the model the reference creates at src/test_system.py:90-95 and calls at src/models/train.py:341) up-samples
# with F.interpolate(scale_factor=2, mode="nearest") and convolves cat([up, skip]) 3x3 / pad 1.  Rows oy - 1, oy, oy + 1 of the
# up-sampled tensor are rows q - 1, q, q of `a` for oy = 2 q and rows q, q, q + 1 for oy = 2 q + 1 (columns alike): taps that read
# the same pixel of `a` can be summed ahead of time.
UP_TAPS = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}      # (phase, u) -> kernel rows (columns) summed into tap u


def phase_weights(w):
    """w: [co][ci][3][3] fp32 -> W'[py][px][co][ci][u][v] fp32, W'_{py,px}[u][v] = sum_{ky in Ky(py,u)} sum_{kx in Kx(px,v)} w[ky][kx],
    summed in fp32 in the packer's order (ky outer, kx inner, starting from zero)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    co, ci = w.shape[:2]
    out = np.zeros((2, 2, co, ci, 2, 2), dtype=np.float32)
    for py in range(2):
        for px in range(2):
            for u in range(2):
                for v in range(2):
                    acc = np.zeros((co, ci), dtype=np.float32)
                    for ky in UP_TAPS[(py, u)]:
                        for kx in UP_TAPS[(px, v)]:
                            acc = (acc + w[:, :, ky, kx]).astype(np.float32)
                    out[py, px, :, :, u, v] = acc
    return out


def up_conv_phases(a, w, dtype=np.float64):
    """conv3x3(nearest_x2(a), w, pad 1) evaluated as the four 2x2 phase convolutions of a: a [n][ci][h][w], w [co][ci][3][3]
    -> y [n][co][2h][2w].  y[2q+py, 2r+px] = sum_{u,v} W'_{py,px}[u][v] . a[q + py - 1 + u, r + px - 1 + v] (zero outside a)."""
    a = np.asarray(a, dtype=dtype)
    pw = phase_weights(w).astype(dtype)
    n, ci, h, wd = a.shape
    co = w.shape[0]
    ap = np.zeros((n, ci, h + 2, wd + 2), dtype=dtype)
    ap[:, :, 1:-1, 1:-1] = a
    y = np.zeros((n, co, 2 * h, 2 * wd), dtype=dtype)
    for py in range(2):
        for px in range(2):
            acc = np.zeros((n, co, h, wd), dtype=dtype)
            for u in range(2):
                for v in range(2):
                    win = ap[:, :, py + u:py + u + h, px + v:px + v + wd]          # a[q + py - 1 + u, r + px - 1 + v]
                    acc += np.einsum("oc,nchw->nohw", pw[py, px, :, :, u, v], win)
            y[:, :, py::2, px::2] = acc
    return y


def up_dgrad_phases(dy, w, dtype=np.float64):
    """Gradient of `a` through conv3x3(nearest_x2(a)) from the phase form: dy [n][co][2h][2w] -> da [n][ci][h][w];
    da[i, j] = sum_{py,px,u,v} W'_{py,px}[u][v]^T . dy[2 (i - py + 1 - u) + py, 2 (j - px + 1 - v) + px]."""
    dy = np.asarray(dy, dtype=dtype)
    pw = phase_weights(w).astype(dtype)
    n, co, h2, w2 = dy.shape
    h, wd = h2 // 2, w2 // 2
    ci = w.shape[1]
    da = np.zeros((n, ci, h + 2, wd + 2), dtype=dtype)          # padded: contributions that fall outside a are dropped
    for py in range(2):
        for px in range(2):
            g = dy[:, :, py::2, px::2]
            for u in range(2):
                for v in range(2):
                    da[:, :, py + u:py + u + h, px + v:px + v + wd] += np.einsum("oc,nohw->nchw", pw[py, px, :, :, u, v], g)
    return da[:, :, 1:-1, 1:-1]


def up_negated_groups(ng):
    """Groups [q1, q3) of the phase kernels' K loop that run on negated weights (csrc/conv_up_f32x3.hip::up_negated_groups);
    ng = 4 * chunks: four (px, ex) groups per chunk (forward), four phases = virtual chunks per chunk (data gradient)."""
    q1 = (ng + 2) // 4
    return range(q1, ng - q1)
