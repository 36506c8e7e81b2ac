"""JPEG decode, host half + oracle, on the CPU: the C-ABI's header parser and baseline Huffman decoder
(visrep_jpeg_info / visrep_jpeg_entropy_decode: host-only entry points) followed by oracle/jpeg.py's integer restatement of libjpeg-turbo's
islow IDCT / fancy upsampling / YCbCr->RGB must reproduce PIL's Image.open(...).convert('RGB') BIT FOR BIT - that pins the oracle the
device kernels are checked against (tests/test_gpu_jpeg.py) and the Huffman decoder the product ships."""
import io

import numpy as np
import pytest
from PIL import Image

from law_of_vision_representation_in_mllms_amd import device_jpeg as DJ
from oracle import jpeg as OJ


def photo(w, h, seed, grey=False):
    """smooth + textured content (a pure-noise image would put every coefficient at its ceiling)"""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    chans = []
    for c in range(1 if grey else 3):
        f = 127 + 90 * np.sin(xx / (7 + 3 * c) + rs.rand() * 6) * np.cos(yy / (11 + 2 * c) + rs.rand() * 6) + rs.normal(0, 12 + 10 * rs.rand(), (h, w))
        f[h // 3: h // 2, w // 4: w // 2] = 255 * rs.rand()                      # a flat patch with hard edges
        chans.append(np.clip(f, 0, 255).astype(np.uint8))
    return Image.fromarray(chans[0] if grey else np.stack(chans, -1))


def encode(img, **kw):
    b = io.BytesIO()
    img.save(b, "JPEG", **kw)
    return b.getvalue()


CASES = [
    (64, 48, dict(quality=75, subsampling=2)), (97, 61, dict(quality=75, subsampling=2)), (333, 251, dict(quality=90, subsampling=2)),
    (17, 9, dict(quality=75, subsampling=2)), (8, 8, dict(quality=50, subsampling=2)), (2, 3, dict(quality=75, subsampling=2)),
    (120, 77, dict(quality=85, subsampling=1)), (33, 40, dict(quality=60, subsampling=1)), (5, 31, dict(quality=75, subsampling=1)),
    (120, 77, dict(quality=95, subsampling=0)), (31, 33, dict(quality=30, subsampling=0)), (200, 150, dict(quality=100, subsampling=0)),
    (255, 257, dict(quality=75, subsampling=2, optimize=True)), (256, 256, dict(quality=75, subsampling=2, restart_marker_blocks=3)),
    (131, 94, dict(quality=80, subsampling=1, restart_marker_rows=1)), (500, 375, dict(quality=75)),
]


@pytest.mark.parametrize("w,h,kw", CASES)
def test_host_decode_plus_oracle_equals_pil(w, h, kw):
    data = encode(photo(w, h, w * 1000 + h), **kw)
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    info, why = DJ.parse_info(data)
    assert why is None, why
    assert (info.width, info.height, info.ncomp) == (w, h, 3)
    info, coef, qtab = DJ.entropy_decode(data)
    got = OJ.reconstruct(info, coef, qtab)
    assert got.shape == want.shape
    assert np.array_equal(got, want), (np.abs(got.astype(int) - want).max(), int((got != want).sum()))


def test_greyscale_and_restart_interval_header():
    data = encode(photo(77, 50, 5, grey=True), quality=80)
    info, coef, qtab = DJ.entropy_decode(data)
    assert info.ncomp == 1 and info.blocks_w[0] == 10 and info.blocks_h[0] == 7
    assert np.array_equal(OJ.reconstruct(info, coef, qtab), np.asarray(Image.open(io.BytesIO(data)).convert("RGB")))
    data = encode(photo(256, 256, 6), quality=75, restart_marker_blocks=3)
    assert DJ.parse_info(data)[0].restart_interval == 3


def test_files_this_decoder_does_not_take_say_why():
    prog = encode(photo(64, 64, 7), quality=75, progressive=True)
    info, why = DJ.parse_info(prog)
    assert info.progressive == 1 and "progressive" in why
    with pytest.raises(ValueError, match="progressive"):
        DJ.entropy_decode(prog)
    cmyk = encode(photo(32, 32, 8).convert("CMYK"), quality=75)
    assert "CMYK" in DJ.parse_info(cmyk)[1]
    assert "SOI" in DJ.parse_info(b"not a jpeg at all")[1]
    good = encode(photo(64, 64, 9), quality=75)
    with pytest.raises(ValueError):
        DJ.entropy_decode(good[: len(good) // 6])                              # truncated inside the headers


@pytest.mark.parametrize("kw", [dict(quality=75, subsampling=2), dict(quality=75, subsampling=2, restart_marker_blocks=3)])
def test_truncated_or_short_entropy_data_is_an_error_not_garbage(kw):
    """A scan that ends before its MCUs (file cut inside the entropy-coded segment, or an EOI placed early) must not yield blocks decoded
    from the zero padding: PIL raises "image file is truncated" for the cut file (the reference stops there), so the host decoder reports
    an error and DeviceJpegDecoder hands the file to PIL, which raises like the reference."""
    good = encode(photo(160, 120, 21), **kw)
    info, _, _ = DJ.entropy_decode(good)                                       # the whole file decodes
    for cut in (len(good) // 2, len(good) - 40):
        with pytest.raises(ValueError, match="truncated|restart"):
            DJ.entropy_decode(good[:cut])
        with pytest.raises(OSError):
            Image.open(io.BytesIO(good[:cut])).convert("RGB")
    early_eoi = good[: len(good) // 2] + b"\xff\xd9"                           # complete markers, short scan
    with pytest.raises(ValueError, match="truncated|restart"):
        DJ.entropy_decode(early_eoi)
