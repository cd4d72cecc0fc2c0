"""CPU restatement of the split-bf16 operand formats of the conv kernels (TEST INFRASTRUCTURE ONLY).

Not part of the reference (fy-vision/DiGA computes its convolutions in fp32 through cuDNN); these are the build's own
byte formats, restated in numpy so that the device kernels that produce them are pinned bit for bit:

  split          x (fp32) -> hi = bf16(x) (round to nearest even), lo = bf16(x - float(hi))          csrc/conv.hip split4
  twin           [M][C] fp32 -> per row and group of 8 channels 16 B of hi + 16 B of lo              make_twin_kernel
  weight image   [K][RS][C] fp32 -> per tile of bn output channels and 32-channel K-step (tap-major):
                 hi plane then lo plane, one 64-byte row per output channel, the 16-byte k-slot s of row r
                 stored at slot s ^ swz(r)                                                             split_image_kernel
"""
import numpy as np


def bf16_rne(x):
    """fp32 array -> uint16 bf16 bit patterns, round to nearest even (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def split(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = bf16_rne(x)
    lo = bf16_rne(x - bf16_to_f32(hi))
    return hi, lo


def twin(x):
    """x [M, C] fp32 (C % 8 == 0) -> uint8 [M * C * 4]."""
    m, c = x.shape
    hi, lo = split(x)
    out = np.empty((m, c // 8, 2, 8), dtype=np.uint16)
    out[:, :, 0, :] = hi.reshape(m, c // 8, 8)
    out[:, :, 1, :] = lo.reshape(m, c // 8, 8)
    return out.reshape(-1).view(np.uint8)


def lds_swz(r):
    return ((0x78 >> (((r >> 2) & 3) * 2)) & 3) ^ (((r >> 1) & 1) << 1)


def weight_image(w, bn=None):
    """w [K, RS, C] fp32 (C % 32 == 0) -> uint8 image; bn = 128 if K > 64 else 64."""
    k, rs, c = w.shape
    bn = bn or (128 if k > 64 else 64)
    tiles, cch = -(-k // bn), c // 32
    hi, lo = split(w)
    img = np.zeros((tiles, rs * cch, 2, bn, 4, 8), dtype=np.uint16)
    for t in range(tiles):
        for r in range(bn):
            n = min(t * bn + r, k - 1)
            sw = lds_swz(r)
            for tap in range(rs):
                for cc in range(cch):
                    for s in range(4):
                        sl = slice(cc * 32 + 8 * s, cc * 32 + 8 * s + 8)
                        img[t, tap * cch + cc, 0, r, s ^ sw] = hi[n, tap, sl]
                        img[t, tap * cch + cc, 1, r, s ^ sw] = lo[n, tap, sl]
    return img.reshape(-1).view(np.uint8)
