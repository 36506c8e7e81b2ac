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
_HOST_BOUNDS = {}


def _device_tables(n_in, n_out, device, filter="bicubic"):
    key = (n_in, n_out, str(device), filter)
    if key not in _TABLES:
        bounds, kk, ksize = pil_coeffs(n_in, n_out, filter)
        _TABLES[key] = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
        _HOST_BOUNDS[(n_in, n_out, filter)] = bounds.reshape(-1, 2)
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


PD_FIELDS = 24            # int64 fields per image descriptor (csrc/convnet.hip: PD_*; include/visrep.h visrep_preprocess_u8_batch)


class _PinnedRing:
    """A few pinned host buffers handed out in turn, each guarded by the event of the upload that last read it: descriptor uploads
    stay asynchronous (a pageable `tensor.to(device)` would block the host behind everything already queued on the stream, e.g. the
    previous batch's tower forward)."""

    def __init__(self, n=4):
        self.buf, self.ev, self.turn = [None] * n, [None] * n, 0

    def upload(self, host: np.ndarray, device) -> torch.Tensor:
        k = self.turn
        self.turn = (k + 1) % len(self.buf)
        if self.ev[k] is not None:
            self.ev[k].synchronize()
        raw = host.view(np.uint8).reshape(-1)
        if self.buf[k] is None or self.buf[k].numel() < raw.size:
            self.buf[k] = torch.empty(max(raw.size, 1 << 16), dtype=torch.uint8).pin_memory()
        self.buf[k].numpy()[: raw.size] = raw
        dev = self.buf[k][: raw.size].to(device, non_blocking=True)
        self.ev[k] = torch.cuda.Event()
        self.ev[k].record(torch.cuda.current_stream(device))
        return dev


_RING = _PinnedRing()


def preprocess_batch(images: Sequence[torch.Tensor], sizes: Sequence[Tuple[int, int]], boxes: Sequence[Tuple[int, int, int, int]],
                     mean: Sequence[float], std: Sequence[float], dtype=torch.float32, pad_background=None, flip: bool = False,
                     filter: str = "bicubic", out: torch.Tensor = None) -> torch.Tensor:
    """images: uint8 [H, W, 3] device tensors of any sizes; sizes[i] = (ow, oh) the resize target; boxes[i] = (left, top, w, h) the crop
    in the resized image (all crops the same w x h) -> [n, 3, h, w] normalised pixels, in TWO launches for the whole batch
    (visrep_preprocess_u8_batch).  pad_background = (r, g, b): the image is first pasted centred on a square canvas of that colour
    (llava/mm_utils.py:78-91 expand2square); flip: mirrored left-right first (pck_train.py:112).  Bit-identical to
    to_tensor(resize_u8(img)) per image, i.e. to PIL + the CPU processors."""
    import ctypes as C
    lib = _lib.require_gpu()
    n = len(images)
    if not (n == len(sizes) == len(boxes)):
        raise ValueError("preprocess_batch: images, sizes and boxes must have the same length")
    _, _, cw, ch = boxes[0]
    device = images[0].device if n else torch.device("cuda")
    if out is None:
        out = torch.empty(n, 3, ch, cw, dtype=dtype, device=device)
    if n == 0:
        return out
    desc = np.zeros((n, PD_FIELDS), np.int64)
    mid_off, max_rows, keep = 0, 0, []
    for i, (im, (ow, oh), (l, t, w, h)) in enumerate(zip(images, sizes, boxes)):
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_cuda:
            raise ValueError("preprocess_batch wants uint8 [H, W, 3] device tensors")
        if (w, h) != (cw, ch) or l < 0 or t < 0 or l + w > ow or t + h > oh:
            raise ValueError("preprocess_batch: every crop must have the same size and lie inside its resized image")
        im = im.contiguous()
        keep.append(im)
        H, W = int(im.shape[0]), int(im.shape[1])
        VH, VW, py, px, bg = H, W, 0, 0, 0
        if pad_background is not None and W != H:
            VH = VW = max(W, H)
            py, px = ((W - H) // 2, 0) if W > H else (0, (H - W) // 2)
            bg = int(pad_background[0]) | int(pad_background[1]) << 8 | int(pad_background[2]) << 16
        D = desc[i]
        D[0:9] = (im.data_ptr(), H, W, VH, VW, py, px, bg, int(bool(flip)))
        r0, nr = t, h                                              # canvas rows the second pass reads
        if VH != oh:
            vb, vk, vks = _device_tables(VH, oh, device, filter)
            D[15:18] = (vb.data_ptr(), vk.data_ptr(), vks)
            bh = _host_bounds(VH, oh, filter)
            r0 = int(bh[t, 0])
            nr = int(bh[t + h - 1, 0] + bh[t + h - 1, 1]) - r0
        if VW != ow:
            hb, hk, hks = _device_tables(VW, ow, device, filter)
            D[12:15] = (hb.data_ptr(), hk.data_ptr(), hks)
            D[9] = mid_off                                         # offset now, pointer once the scratch exists
            mid_off += (nr * cw * 3 + 63) // 64 * 64
            max_rows = max(max_rows, nr)
        D[10:12] = (r0, nr)
        D[18:20] = (l, t)
    scratch = torch.empty(max(mid_off, 64), dtype=torch.uint8, device=device)
    desc[:, 9] += scratch.data_ptr()
    ddesc = _RING.upload(desc, device)
    m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    with torch.cuda.device(device):
        rc = lib.visrep_preprocess_u8_batch(_lib.ptr(ddesc), n, max_rows, ch, cw, m, s, _lib.ptr(out), _lib.F32 if out.dtype == torch.float32 else _lib.BF16,
                                            _lib.stream_ptr())
    _lib.check(rc, "visrep_preprocess_u8_batch")
    ddesc.record_stream(torch.cuda.current_stream(device))
    scratch.record_stream(torch.cuda.current_stream(device))
    return out


def _host_bounds(n_in, n_out, filter="bicubic"):
    key = (n_in, n_out, filter)
    if key not in _HOST_BOUNDS:
        _HOST_BOUNDS[key] = pil_coeffs(n_in, n_out, filter)[0].reshape(-1, 2)
    return _HOST_BOUNDS[key]


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
        return {"pixel_values": self.preprocess_padded(images, None)}

    @torch.no_grad()
    def preprocess_padded(self, images, pad_background):
        """preprocess() of expand2square(image, pad_background) (llava/mm_utils.py:78-95 `image_aspect_ratio == 'pad'`) without building the
        square: padding, both resize passes, crop and normalisation of the whole batch are two launches (preprocess_batch)."""
        dev = []
        for im in images:
            if isinstance(im, torch.Tensor):                                     # RGB u8 [H, W, 3] already in HBM (device_jpeg.DeviceJpegDecoder)
                dev.append(im.to(self.device))
            else:
                dev.append(torch.from_numpy(np.array(im.convert("RGB"))).to(self.device, non_blocking=True))   # PIL image: decoded on the host
        geo = []
        for d in dev:
            H, W = int(d.shape[0]), int(d.shape[1])
            if pad_background is not None:
                H = W = max(H, W)
            geo.append(self.geometry(W, H))
        return preprocess_batch(dev, [g[0] for g in geo], [g[1] for g in geo], self.mean, self.std, self.dtype, pad_background)

    __call__ = preprocess
