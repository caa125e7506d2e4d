"""CPU restatement (numpy) of the JPEG reconstruction stages that csrc/jpeg.hip runs on the GPU -- TEST INFRASTRUCTURE ONLY (tests/,
__graft_entry__.smoke(), bench.py's cpu_baseline); the product never imports it.

What it restates: the reference decodes images with Pillow (`Image.open(path).convert('RGB')`, ref/modules/loader.py:15), i.e.
with libjpeg-turbo in its default configuration (a third-party dependency of the reference, absent from /root/reference; the
Pillow wheel in this image bundles libjpeg-turbo 3.x): JDCT_ISLOW inverse DCT (Loeffler-Ligtenberg-Moschytz, CONST_BITS = 13,
PASS1_BITS = 2), "fancy" triangle-filter chroma upsampling (plain replication when the sub-sampled width is <= 2), 16-bit
fixed-point YCbCr -> RGB.  Pinned: tests/test_jpeg_cpu.py checks this module (fed with the coefficients of the product's host
Huffman decoder) byte for byte against Pillow's own decode of the same files, so oracle and entropy decoder are both anchored on
the reference's real decoder.
"""
import numpy as np

F0_298, F0_390, F0_541, F0_765, F0_899, F1_175 = 2446, 3196, 4433, 6270, 7373, 9633
F1_501, F1_847, F1_961, F2_053, F2_562, F3_072 = 12299, 15137, 16069, 16819, 20995, 25172


def _llm(x):
    """one 1-D pass along axis -2 of int64 [..., 8, n]; returns the eight un-descaled outputs"""
    i0, i1, i2, i3, i4, i5, i6, i7 = (x[..., k, :] for k in range(8))
    z1 = (i2 + i6) * F0_541
    tmp2 = z1 - i6 * F1_847
    tmp3 = z1 + i2 * F0_765
    tmp0 = (i0 + i4) << 13
    tmp1 = (i0 - i4) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = i7, i5, i3, i1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F1_175
    t0, t1, t2, t3 = t0 * F0_298, t1 * F2_053, t2 * F3_072, t3 * F1_501
    z1, z2, z3, z4 = -z1 * F0_899, -z2 * F2_562, -z3 * F1_961 + z5, -z4 * F0_390 + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    return np.stack([tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3], axis=-2)


def idct_islow(coefs, qt):
    """coefs int16 [nb, 64] (natural order), qt uint16 [64] -> uint8 [nb, 8, 8]"""
    x = coefs.astype(np.int64).reshape(-1, 8, 8) * qt.astype(np.int64).reshape(1, 8, 8)
    ws = (_llm(x) + (1 << 10)) >> 11                       # columns: [nb, k(row), col]
    out = (_llm(ws.transpose(0, 2, 1)) + (1 << 17)) >> 18  # rows: input [nb, k(col), row] -> output [nb, col, row]
    out = out.transpose(0, 2, 1)
    sx = ((out & 1023) ^ 512) - 512                        # the 10-bit wrap of libjpeg's range-limit table
    return np.clip(sx + 128, 0, 255).astype(np.uint8)


def plane(coefs, qt, bw, bh):
    s = idct_islow(coefs, qt).reshape(bh, bw, 8, 8)
    return s.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def upsample(pl, cw, ch, hsub, vsub, width, height):
    """chroma plane (padded) -> int32 [height, width]"""
    p = pl[:ch, :cw].astype(np.int32)
    if hsub == 1 and vsub == 1:
        return p[:height, :width]
    fancy = cw > 2
    xs = np.arange(width)
    i = xs >> 1
    if vsub == 1:
        if not fancy:
            return p[:height][:, i]
        nb = np.where(xs & 1, np.minimum(i + 1, cw - 1), np.maximum(i - 1, 0))
        rnd = np.where(xs & 1, 2, 1)
        return (3 * p[:height][:, i] + p[:height][:, nb] + rnd) >> 2
    ys = np.arange(height)
    r0 = ys >> 1
    if not fancy:
        return p[r0][:, i]
    r1 = np.clip(np.where(ys & 1, r0 + 1, r0 - 1), 0, ch - 1)
    cs = 3 * p[r0] + p[r1]                                  # [height, cw]
    nb = np.where(xs & 1, np.minimum(i + 1, cw - 1), np.maximum(i - 1, 0))
    rnd = np.where(xs & 1, 7, 8)
    return (3 * cs[:, i] + cs[:, nb] + rnd) >> 4


def reconstruct(coefs, qt, info):
    """coefs int16 [coef_blocks, 64] of ONE image (component-major), qt uint16 [3, 64], info with the klab_jpeg_info fields
    -> HWC uint8 RGB, as `Image.open(...).convert('RGB')`"""
    W, H = info.width, info.height
    planes, b0 = [], 0
    for c in range(info.ncomp):
        nb = info.bw[c] * info.bh[c]
        planes.append(plane(coefs[b0:b0 + nb], qt[c], info.bw[c], info.bh[c]))
        b0 += nb
    Y = planes[0][:H, :W].astype(np.int32)
    if info.ncomp == 1:
        return np.repeat(Y[:, :, None], 3, axis=2).astype(np.uint8)
    hsub, vsub = info.hmax // info.hs[1], info.vmax // info.vs[1]
    cw, ch = -(-W // hsub), -(-H // vsub)
    c1 = upsample(planes[1], cw, ch, hsub, vsub, W, H)
    c2 = upsample(planes[2], cw, ch, hsub, vsub, W, H)
    if info.colour == 2:  # stored as RGB
        return np.stack([Y, c1, c2], axis=2).astype(np.uint8)
    cb, cr = c1 - 128, c2 - 128
    r = Y + ((91881 * cr + 32768) >> 16)
    g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = Y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=2), 0, 255).astype(np.uint8)
