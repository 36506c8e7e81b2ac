"""CPU restatement (numpy, integer arithmetic) of what PIL's Image.open(path).convert('RGB') does to a baseline JPEG AFTER entropy
decoding: the reference's image loaders (C_score/extract_feature.py:65-66, llava/mm_utils.py:78-95) decode with Pillow, whose bundled
libjpeg-turbo (3.x, API level 6.2; a third-party dependency, absent from /root/reference) runs
  jidctint.c  jpeg_idct_islow            dequantise + accurate integer IDCT (CONST_BITS 13, PASS1_BITS 2, range_limit[x & 1023])
  jdsample.c  h2v1_fancy_upsample / h2v2_fancy_upsample (plain replication when a component is <= 2 samples wide), with the
              context rows of jdmainct.c (first / last real row replicated)
  jdcolor.c   ycc_rgb_convert            16-bit fixed-point tables, FIX(x) = (int)(x * 65536 + .5)
Test infrastructure only (tests/, smoke): the checker of csrc/jpeg_decode.hip's device kernels.  PINNED against PIL itself on files PIL
wrote (tests/test_host_jpeg.py: 4:4:4 / 4:2:2 / 4:2:0, grey, odd sizes, restart markers, optimised Huffman tables, qualities 30-100)."""
import numpy as np


def range_limit_idct(x):
    i = x & 1023
    return np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896)))


def _idct_1d(i, shift):
    """i: int64 [..., 8] -> [..., 8]; 32-bit wrap-around like the SIMD kernels (inputs are 16-bit products)."""
    w = lambda v: ((v + (1 << 31)) % (1 << 32)) - (1 << 31)
    z2, z3 = i[..., 2], i[..., 6]
    z1 = w((z2 + z3) * 4433)
    tmp2 = w(z1 + z3 * -15137)
    tmp3 = w(z1 + z2 * 6270)
    z2, z3 = i[..., 0], i[..., 4]
    tmp0, tmp1 = w((z2 + z3) << 13), w((z2 - z3) << 13)
    tmp10, tmp13, tmp11, tmp12 = w(tmp0 + tmp3), w(tmp0 - tmp3), w(tmp1 + tmp2), w(tmp1 - tmp2)
    tmp0, tmp1, tmp2, tmp3 = i[..., 7], i[..., 5], i[..., 3], i[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = w((z3 + z4) * 9633)
    tmp0, tmp1, tmp2, tmp3 = w(tmp0 * 2446), w(tmp1 * 16819), w(tmp2 * 25172), w(tmp3 * 12299)
    z1, z2, z3, z4 = w(z1 * -7373), w(z2 * -20995), w(z3 * -16069), w(z4 * -3196)
    z3, z4 = w(z3 + z5), w(z4 + z5)
    tmp0, tmp1, tmp2, tmp3 = w(tmp0 + z1 + z3), w(tmp1 + z2 + z4), w(tmp2 + z2 + z3), w(tmp3 + z1 + z4)
    r = 1 << (shift - 1)
    d = lambda v: w(v + r) >> shift
    return np.stack([d(tmp10 + tmp3), d(tmp11 + tmp2), d(tmp12 + tmp1), d(tmp13 + tmp0),
                     d(tmp13 - tmp0), d(tmp12 - tmp1), d(tmp11 - tmp2), d(tmp10 - tmp3)], -1)


def idct_plane(coef, q, bh, bw):
    """coef int16 [bh*bw*64] (blocks row-major, natural order), q uint16 [64] -> uint8 [bh*8, bw*8]"""
    c = coef.astype(np.int64).reshape(bh * bw, 8, 8) * q.astype(np.int64).reshape(1, 8, 8)
    ws = _idct_1d(c.transpose(0, 2, 1), 11).transpose(0, 2, 1)          # pass 1: columns
    out = range_limit_idct(_idct_1d(ws, 18))                            # pass 2: rows
    return out.reshape(bh, bw, 8, 8).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8).astype(np.uint8)


def upsample(p, cw, ch, W, H, hmax, vmax):
    """component plane (real samples [ch, cw]) -> full resolution int32 [H, W]"""
    p = p[:ch, :cw].astype(np.int32)
    if hmax == 1:
        return p[:H, :W]
    x = np.arange(W)
    c = x >> 1
    if vmax == 1:
        v = p[:H][:, c]
        if cw <= 2:
            return v
        left, right = p[:H][:, np.maximum(c - 1, 0)], p[:H][:, np.minimum(c + 1, cw - 1)]
        even = np.where(c == 0, v, (3 * v + left + 1) >> 2)
        odd = np.where(c == cw - 1, v, (3 * v + right + 2) >> 2)
        return np.where((x & 1) == 0, even, odd)
    y = np.arange(H)
    r = y >> 1
    if cw <= 2:
        return p[r][:, c]
    rf = np.clip(np.where(y & 1, r + 1, r - 1), 0, ch - 1)
    colsum = 3 * p[r] + p[rf]                                           # [H, cw]
    cur = colsum[:, c]
    last, nxt = colsum[:, np.maximum(c - 1, 0)], colsum[:, np.minimum(c + 1, cw - 1)]
    even = np.where(c == 0, (cur * 4 + 8) >> 4, (cur * 3 + last + 8) >> 4)
    odd = np.where(c == cw - 1, (cur * 4 + 7) >> 4, (cur * 3 + nxt + 7) >> 4)
    return np.where((x & 1) == 0, even, odd)


def ycc_to_rgb(Y, cb, cr):
    Y, cb, cr = Y.astype(np.int64), cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = Y + ((91881 * cr + 32768) >> 16)
    g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = Y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def reconstruct(info, coef, qtab):
    """info: fields of VisrepJpegInfo (object with width, height, ncomp, blocks_w/h, comp_w/h, hmax, vmax); coef int16 [coef_count];
    qtab uint16 [ncomp, 64] -> RGB uint8 [H, W, 3] (= np.asarray(Image.open(f).convert('RGB')))"""
    planes, off = [], 0
    for c in range(info.ncomp):
        n = info.blocks_w[c] * info.blocks_h[c] * 64
        planes.append(idct_plane(coef[off:off + n], qtab[c], info.blocks_h[c], info.blocks_w[c]))
        off += n
    W, H = info.width, info.height
    Y = planes[0][:H, :W]
    if info.ncomp == 1:
        return np.repeat(Y[:, :, None], 3, 2)
    cb = upsample(planes[1], info.comp_w[1], info.comp_h[1], W, H, info.hmax, info.vmax)
    cr = upsample(planes[2], info.comp_w[2], info.comp_h[2], W, H, info.hmax, info.vmax)
    return ycc_to_rgb(Y, cb, cr)
