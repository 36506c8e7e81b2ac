"""JPEG files -> RGB u8 tensors in HBM: the decode step of the reference's image loaders (PIL `Image.open(path)
.convert('RGB')`, C_score/extract_feature.py:65-66, llava/mm_utils.py:78-95, llava/feature/extract.py:198-214), bit-exact with PIL.

Host threads parse + Huffman-decode (visrep_jpeg_entropy_decode: a serial bit stream per file; ctypes releases the GIL, so a thread
pool scales over the host cores), ONE pinned upload carries the batch's quantised coefficients, ONE launch pair
(visrep_jpeg_reconstruct) does dequantisation + IDCT + chroma upsampling + YCbCr -> RGB for every image of the batch.  Files this decoder
does not take (progressive, CMYK, arithmetic coding, unusual chroma layouts - `why` comes from visrep_last_error) are decoded by PIL like
before and uploaded; `DeviceJpegDecoder.stats` counts both routes, nothing is silent.
"""
from __future__ import annotations

import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib

JD_FIELDS = 32            # int64 fields per image descriptor (csrc/jpeg_decode.hip: JD_*)


def parse_info(data: bytes):
    """(info, why): headers of a JPEG stream; why = None when this decoder takes the file, else the reason it does not."""
    lib = _lib.load()
    info = _lib.JpegInfo()
    rc = lib.visrep_jpeg_info(data, len(data), C.byref(info))
    if rc != 0 or info.unsupported:
        return info, _lib.last_error() or "unsupported JPEG"
    return info, None


def entropy_decode(data: bytes, info=None):
    """Host half: (info, coef int16 [coef_count], qtab uint16 [ncomp, 64]); raises ValueError for files this decoder does not take."""
    lib = _lib.load()
    if info is None:
        info, why = parse_info(data)
        if why:
            raise ValueError(why)
    coef = np.empty(info.coef_count, np.int16)
    qtab = np.empty((info.ncomp, 64), np.uint16)
    rc = lib.visrep_jpeg_entropy_decode(data, len(data), coef.ctypes.data_as(C.c_void_p), qtab.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(_lib.last_error())
    return info, coef, qtab


def _align(x: int, a: int = 64) -> int:
    return (x + a - 1) // a * a


class DeviceJpegDecoder:
    """decode(files_or_bytes) -> list of uint8 [H, W, 3] device tensors (views of one batch buffer)."""

    def __init__(self, device=None, threads: Optional[int] = None):
        self.lib = _lib.require_gpu()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.pool = ThreadPoolExecutor(max_workers=threads or min(32, os.cpu_count() or 1))
        self.stats = {"device": 0, "pil": 0, "pil_reasons": {}}
        # two pinned staging buffers (coefficients + tables + descriptors), alternated: the upload of batch k may still be in flight
        # while batch k + 1 is assembled.  Assembly is plain numpy memcpy - torch CPU copies wake the intra-op pool, which on a
        # 256-core host costs more than the Huffman decode (profiles/round1_pipeline.md).
        self._stage = [None, None]
        self._stage_ev = [None, None]
        self._turn = 0

    def _staging(self, nbytes: int):
        k = self._turn
        self._turn ^= 1
        if self._stage_ev[k] is not None:
            self._stage_ev[k].synchronize()
        if self._stage[k] is None or self._stage[k].numel() < nbytes:
            self._stage[k] = torch.empty(max(nbytes, 1 << 24) * 5 // 4, dtype=torch.uint8).pin_memory()
        return k, self._stage[k]

    @staticmethod
    def _host(item):
        data = item if isinstance(item, (bytes, bytearray, memoryview)) else open(item, "rb").read()
        info, why = parse_info(data)
        if why:
            return ("pil", item, why)
        try:
            return ("dev",) + entropy_decode(data, info)
        except ValueError as e:                                    # corrupt entropy stream: let PIL have its say (it raises or repairs)
            return ("pil", item, str(e))

    def submit(self, items: Sequence):
        """Start the host half (parse + Huffman decode, one file per pool thread) of a batch; hand the result to finish()."""
        return [self.pool.submit(self._host, it) for it in items]

    def decode(self, items: Sequence) -> List[torch.Tensor]:
        return self.finish(self.submit(items))

    @torch.no_grad()
    def finish(self, futures) -> List[torch.Tensor]:
        """Device half of a submitted batch: one upload, one launch pair; returns uint8 [H, W, 3] tensors in submission order."""
        parts = [f.result() for f in futures]
        out: List[Optional[torch.Tensor]] = [None] * len(parts)
        dev_idx = [i for i, p in enumerate(parts) if p[0] == "dev"]
        if dev_idx:
            desc = np.zeros((len(dev_idx), JD_FIELDS), np.int64)
            coef_off = plane_off = rgb_off = 0
            max_blocks = max_pixels = 0
            for j, i in enumerate(dev_idx):
                info = parts[i][1]
                D = desc[j]
                nb = 0
                for c in range(info.ncomp):
                    D[0 + c], D[3 + c] = coef_off, plane_off
                    D[6 + c], D[9 + c], D[12 + c], D[15 + c] = info.blocks_w[c], info.blocks_h[c], info.comp_w[c], info.comp_h[c]
                    n = info.blocks_w[c] * info.blocks_h[c]
                    coef_off += n * 64
                    plane_off = _align(plane_off + n * 64)
                    nb += n
                D[18], D[19], D[20], D[21], D[22], D[23], D[24] = info.width, info.height, info.ncomp, info.hmax, info.vmax, rgb_off, 192 * j
                rgb_off = _align(rgb_off + info.width * info.height * 3)
                max_blocks, max_pixels = max(max_blocks, nb), max(max_pixels, info.width * info.height)
            # one staging buffer = [coefficients int16 | tables uint16 [n, 3, 64] | descriptors int64 [n, 32]], one upload
            q_off = _align(coef_off * 2)
            d_off = _align(q_off + len(dev_idx) * 384)
            total = d_off + desc.nbytes
            k, stage = self._staging(total)
            hb = stage.numpy()
            hcoef = hb[: coef_off * 2].view(np.int16)
            hq = hb[q_off: q_off + len(dev_idx) * 384].view(np.uint16).reshape(len(dev_idx), 3, 64)
            hq[:] = 0
            for j, i in enumerate(dev_idx):
                info, coef, qtab = parts[i][1:]
                np.copyto(hcoef[int(desc[j, 0]): int(desc[j, 0]) + coef.size], coef)
                hq[j, : info.ncomp] = qtab
            hb[d_off: d_off + desc.nbytes] = desc.view(np.uint8).reshape(-1)
            dbuf = stage[:total].to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._stage_ev[k] = ev
            dcoef, dq, ddesc = dbuf[: coef_off * 2], dbuf[q_off: q_off + len(dev_idx) * 384], dbuf[d_off: d_off + desc.nbytes]
            planes = torch.empty(max(plane_off, 64), dtype=torch.uint8, device=self.device)
            rgb = torch.empty(max(rgb_off, 64), dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.visrep_jpeg_reconstruct(_lib.ptr(dcoef), _lib.ptr(dq), _lib.ptr(ddesc), len(dev_idx), max_blocks, max_pixels,
                                                            _lib.ptr(planes), _lib.ptr(rgb), _lib.stream_ptr()), "visrep_jpeg_reconstruct")
            for j, i in enumerate(dev_idx):
                info = parts[i][1]
                o = int(desc[j, 23])
                out[i] = rgb[o: o + info.width * info.height * 3].view(info.height, info.width, 3)
            self.stats["device"] += len(dev_idx)
        for i, p in enumerate(parts):
            if p[0] == "pil":
                import io
                from PIL import Image
                src = p[1]
                img = Image.open(io.BytesIO(bytes(src)) if isinstance(src, (bytes, bytearray, memoryview)) else src).convert("RGB")
                out[i] = torch.from_numpy(np.asarray(img).copy()).to(self.device)
                self.stats["pil"] += 1
                self.stats["pil_reasons"][p[2]] = self.stats["pil_reasons"].get(p[2], 0) + 1
        return out
