"""Device-side image pre-processing: everything after JPEG decode runs on the MI355X.

The reference resizes with `PIL.Image.resize` (default BICUBIC: C_score/extract_feature.py:65-66; HF CLIPImageProcessor /
llava `process_images`, llava/mm_utils.py:64-95) and then does ToTensor / normalise on the host, one image at a time.  Here
the decoded uint8 image is copied to the GPU once and

    resize_u8   Pillow's 8-bit separable bicubic resampling, BIT-EXACT (visrep_resample_u8, two passes)
    to_tensor   crop + /255 + (x - mean) / std in IEEE fp32, bit-identical to the CPU processors (visrep_u8hwc_to_chw_norm)

`pil_coeffs` restates Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` (src/libImaging/Resample.c, third-party, version
of the installed Pillow; checked bit-for-bit against `Image.resize` itself in tests/test_host_preprocess.py) in Python
floats, i.e. C doubles, with the same operation order: the tables are tiny and cached per (in, out) size pair.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import Sequence, Tuple

import numpy as np
import torch

from . import _lib

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:                      # Pillow's lanczos_filter: truncated sinc, a = 3
    return _sinc(x) * _sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


def _bilinear(x: float) -> float:
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


# name -> (kernel, support) as in Pillow's Resample.c (BICUBIC: PIL's default for Image.resize; LANCZOS: the GeoAware-SC image loader,
# C_score/utils/utils_correspondence.py:75-114)
FILTERS = {"bicubic": (_bicubic, 2.0), "lanczos": (_lanczos, 3.0), "bilinear": (_bilinear, 1.0)}


@lru_cache(maxsize=256)
def pil_coeffs(in_size: int, out_size: int, filter: str = "bicubic") -> Tuple[np.ndarray, np.ndarray, int]:
    """(bounds int32 [out, 2] = (xmin, count), kk int32 [out, ksize], ksize) of one of Pillow's resampling filters for one axis."""
    kernel, base_support = FILTERS[filter]
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = base_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [kernel((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def resample_reference(img: np.ndarray, size: Tuple[int, int], filter: str = "bicubic") -> np.ndarray:
    """The same two fixed-point passes in numpy (host check of the tables; not used by the product path)."""
    ow, oh = size
    out = img
    for axis, (n_in, n_out) in ((1, (img.shape[1], ow)), (0, (img.shape[0], oh))):
        if n_in == n_out:
            continue
        bounds, kk, _ = pil_coeffs(n_in, n_out, filter)
        src = np.moveaxis(out.astype(np.int64), axis, 0)
        dst = np.empty((n_out,) + src.shape[1:], np.uint8)
        for xx in range(n_out):
            x0, n = bounds[xx]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0))
            dst[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        out = np.moveaxis(dst, 0, axis)
    return np.ascontiguousarray(out)


_TABLES = {}


def _device_tables(n_in, n_out, device, filter="bicubic"):
    key = (n_in, n_out, str(device), filter)
    if key not in _TABLES:
        bounds, kk, ksize = pil_coeffs(n_in, n_out, filter)
        _TABLES[key] = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
    return _TABLES[key]


def resize_u8(img: torch.Tensor, size: Tuple[int, int], filter: str = "bicubic") -> torch.Tensor:
    """img uint8 [H, W, 3] on the GPU -> uint8 [OH, OW, 3], bit-identical to PIL `Image.resize((OW, OH), <filter>)`
    (filter: "bicubic" = PIL's default, "lanczos", "bilinear" - the kernel only sees coefficient tables)."""
    lib = _lib.require_gpu()
    if img.dtype != torch.uint8 or img.dim() != 3:
        raise ValueError("resize_u8 wants a uint8 [H, W, C] tensor")
    ow, oh = size
    H, W, C = img.shape
    cur = img.contiguous()
    if W != ow:                                                   # horizontal pass: line = row
        b, k, ks = _device_tables(W, ow, cur.device, filter)
        nxt = torch.empty(H, ow, C, dtype=torch.uint8, device=cur.device)
        rc = lib.visrep_resample_u8(_lib.ptr(cur), _lib.ptr(nxt), H, ow, C, W * C, C, ow * C, C, _lib.ptr(b), _lib.ptr(k), ks, _lib.stream_ptr())
        _lib.check(rc, "visrep_resample_u8")
        cur = nxt
    if H != oh:                                                   # vertical pass: line = column
        b, k, ks = _device_tables(H, oh, cur.device, filter)
        nxt = torch.empty(oh, ow, C, dtype=torch.uint8, device=cur.device)
        rc = lib.visrep_resample_u8(_lib.ptr(cur), _lib.ptr(nxt), ow, oh, C, C, ow * C, C, ow * C, _lib.ptr(b), _lib.ptr(k), ks, _lib.stream_ptr())
        _lib.check(rc, "visrep_resample_u8")
        cur = nxt
    return cur


def geoaware_geometry(width: int, height: int, target_res: int):
    """(resized (w, h), (top, left) of the image inside the target_res square) of the GeoAware-SC loader
    (utils_correspondence.py:75-114): long side -> target_res, short side np.around(), centred with floor-divided offsets."""
    if height <= width:
        w, h = target_res, int(np.around(target_res * height / width))
        return (w, h), ((w - h) // 2, 0)
    w, h = int(np.around(target_res * width / height)), target_res
    return (w, h), (0, (h - w) // 2)


def geoaware_resize(img: torch.Tensor, target_res: int = 224, edge: bool = False) -> torch.Tensor:
    """Device twin of utils_correspondence.resize(img, target_res, resize=True, edge=...): uint8 [H, W, 3] on the GPU ->
    uint8 [target_res, target_res, 3]: LANCZOS resize of the long side to target_res, then zero padding (edge=False) or
    edge replication (edge=True, np.pad mode='edge') of the short side.  Bit-identical to the PIL + numpy original."""
    H, W, _ = img.shape
    (w, h), (top, left) = geoaware_geometry(W, H, target_res)
    small = resize_u8(img, (w, h), "lanczos")
    if not edge:
        canvas = torch.zeros(target_res, target_res, 3, dtype=torch.uint8, device=img.device)
        canvas[top:top + h, left:left + w] = small
        return canvas
    if H <= W:                                                 # rows are missing: replicate the first / last row
        tp = (target_res - h) // 2
        rows = torch.cat([small[:1].expand(tp, -1, -1), small, small[-1:].expand(target_res - h - tp, -1, -1)], 0)
        return rows.contiguous()
    lp = (target_res - w) // 2
    return torch.cat([small[:, :1].expand(-1, lp, -1), small, small[:, -1:].expand(-1, target_res - w - lp, -1)], 1).contiguous()


def geoaware_resize_reference(img: np.ndarray, target_res: int = 224, edge: bool = False) -> np.ndarray:
    """geoaware_resize in numpy over resample_reference (host check; not used by the product path)."""
    H, W, _ = img.shape
    (w, h), (top, left) = geoaware_geometry(W, H, target_res)
    small = resample_reference(img, (w, h), "lanczos")
    if not edge:
        canvas = np.zeros((target_res, target_res, 3), np.uint8)
        canvas[top:top + h, left:left + w] = small
        return canvas
    if H <= W:
        tp = (target_res - h) // 2
        return np.pad(small, [(tp, target_res - h - tp), (0, 0), (0, 0)], mode='edge')
    lp = (target_res - w) // 2
    return np.pad(small, [(0, 0), (lp, target_res - w - lp), (0, 0)], mode='edge')


def to_tensor(img: torch.Tensor, box: Tuple[int, int, int, int], mean: Sequence[float], std: Sequence[float], dtype=torch.float32,
              out: torch.Tensor = None) -> torch.Tensor:
    """img uint8 [H, W, 3] on the GPU, box = (left, top, width, height) -> [3, height, width]: (x / 255 - mean) / std."""
    import ctypes as C
    lib = _lib.require_gpu()
    H, W, _ = img.shape
    l, t, w, h = box
    if out is None:
        out = torch.empty(3, h, w, dtype=dtype, device=img.device)
    m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    rc = lib.visrep_u8hwc_to_chw_norm(_lib.ptr(img), H, W, l, t, h, w, m, s, _lib.ptr(out), _lib.F32 if out.dtype == torch.float32 else _lib.BF16,
                                      _lib.stream_ptr())
    _lib.check(rc, "visrep_u8hwc_to_chw_norm")
    return out


def expand2square_u8(img: torch.Tensor, background) -> torch.Tensor:
    """llava/mm_utils.py:78-91 expand2square on an RGB u8 [H, W, 3] device tensor: paste centred on a square of the background colour."""
    H, W, _ = img.shape
    if W == H:
        return img
    side = max(W, H)
    out = torch.empty(side, side, 3, dtype=torch.uint8, device=img.device)
    out[:] = torch.tensor([int(c) for c in background], dtype=torch.uint8, device=img.device)
    if W > H:
        out[(W - H) // 2: (W - H) // 2 + H] = img
    else:
        out[:, (H - W) // 2: (H - W) // 2 + W] = img
    return out


class DevicePreprocessor:
    """Device twin of image_processing.SimpleImageProcessor (same geometry, same arithmetic): PIL images in, pixel batch out."""

    def __init__(self, resize_to, crop, mean, std, square_resize=False, device=None, dtype=torch.float32):
        self.resize_to, self.crop, self.mean, self.std, self.square_resize = resize_to, crop, list(mean), list(std), square_resize
        self.image_mean, self.image_std = self.mean, self.std                # the attributes llava/mm_utils.py process_images reads
        self.crop_size, self.size = {"height": crop, "width": crop}, {"shortest_edge": resize_to}
        self.device = torch.device(device if device is not None else "cuda")
        self.dtype = dtype

    @classmethod
    def like(cls, proc, device=None, dtype=torch.float32):
        return cls(proc.resize_to, proc.crop, proc.image_mean, proc.image_std, proc.square_resize, device, dtype)

    def geometry(self, w, h):
        if self.square_resize:
            return (self.crop, self.crop), (0, 0, self.crop, self.crop)
        from .llava.model.multimodal_encoder.image_processing import shortest_edge_size
        nw, nh = shortest_edge_size(w, h, self.resize_to)            # HF rule: the long side is truncated, not rounded
        return (nw, nh), ((nw - self.crop) // 2, (nh - self.crop) // 2, self.crop, self.crop)

    @torch.no_grad()
    def preprocess(self, images, return_tensors="pt"):
        if not isinstance(images, (list, tuple)):
            images = [images]
        out = torch.empty(len(images), 3, self.crop, self.crop, dtype=self.dtype, device=self.device)
        for i, im in enumerate(images):
            if isinstance(im, torch.Tensor):                                     # RGB u8 [H, W, 3] already in HBM (device_jpeg.DeviceJpegDecoder)
                dev = im.to(self.device)
            else:
                dev = torch.from_numpy(np.array(im.convert("RGB"))).to(self.device, non_blocking=True)   # PIL image: decoded on the host
            size, box = self.geometry(dev.shape[1], dev.shape[0])
            to_tensor(resize_u8(dev, size), box, self.mean, self.std, out=out[i])
        return {"pixel_values": out}

    __call__ = preprocess
