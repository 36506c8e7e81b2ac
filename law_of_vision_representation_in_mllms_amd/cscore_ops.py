"""Device entry points of the C score (visrep_cscore_transfer / visrep_pck_count).  See csrc/cscore.hip."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def lin_table(P: int, device) -> torch.Tensor:
    # utils_correspondence.py:239-242: torch.tensor(np.linspace(-1, 1, P)).float()
    return torch.tensor(np.linspace(-1, 1, P)).float().to(device)


def pack_rows(img1: np.ndarray, img2: np.ndarray, patch_idx: np.ndarray, nkp: np.ndarray, rows: int = 32):
    """Pack the key points of pairs that share a TARGET image into groups of at most `rows` MFMA tile rows (host side, numpy).

    Pairs are visited target by target (stable order), inside a target largest first, each into the first open group of that target with
    room (first-fit decreasing); a pair's key points are never split.  Returns (rows_tab int32 [n_groups, rows, 4] = (pair, k, source
    image, source patch index) with pair = -1 on empty rows, tgt int32 [n_groups]).  Groups of one target are consecutive, which is what
    the kernel's XCD mapping wants (the second group of a target finds the map in L2)."""
    n = len(nkp)
    order = np.lexsort((-nkp.astype(np.int64), img2.astype(np.int64)))          # by target, then larger pairs first
    tab, tgt = [], []
    open_groups = []                                                           # (index into tab, rows used) of the current target
    cur = None
    for z in order:
        K = int(nkp[z])
        if K <= 0:
            continue
        if cur != int(img2[z]):
            cur, open_groups = int(img2[z]), []
        slot = next((g for g in open_groups if g[1] + K <= rows), None)
        if slot is None:
            tab.append(np.full((rows, 4), -1, np.int32))
            tgt.append(cur)
            slot = [len(tab) - 1, 0]
            open_groups.append(slot)
        t = tab[slot[0]]
        r0 = slot[1]
        t[r0:r0 + K, 0] = z
        t[r0:r0 + K, 1] = np.arange(K)
        t[r0:r0 + K, 2] = img1[z]
        t[r0:r0 + K, 3] = patch_idx[z, :K]
        slot[1] += K
    if not tab:
        return np.zeros((0, rows, 4), np.int32), np.zeros((0,), np.int32)
    return np.stack(tab), np.asarray(tgt, np.int32)


_PACK_CACHE: "dict[tuple, tuple]" = {}


def packed_rows_on(device, img1, img2, patch_idx, nkp, rows: int = 32):
    """(rows_tab, tgt) of pack_rows as device tensors, memoised on the pair list's CONTENT: an evaluation visits the same (category, map
    size) pair lists once per setting - 13 times in the sweep - and the packing is host work (~5 us per pair).

    Callers that hold the pair list on the host (C_score.pck_train._compute_pck does) pass numpy arrays / CPU tensors: nothing is copied
    off the device and no stream is synchronised.  Device tensors are accepted and downloaded (one sync) - pass `packed=` to transfer()
    to keep that out of a timed region.  The key is a 128-bit digest of the four arrays' bytes plus their shapes and dtypes (Python's
    64-bit hash() of the bytes could collide silently)."""
    import hashlib
    arrs = [np.ascontiguousarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t) for t in (img1, img2, patch_idx, nkp)]
    a1, a2, aidx, ank = arrs
    if aidx.ndim != 2 or a1.ndim != 1 or a2.shape != a1.shape or ank.shape != a1.shape or aidx.shape[0] != a1.shape[0]:
        raise ValueError("packed key-point tiles: img1 / img2 / nkp must be [n] and patch_idx [n, kmax]")
    if ank.size and (int(ank.min()) < 0 or int(ank.max()) > min(aidx.shape[1], rows)):
        raise ValueError(f"packed key-point tiles: nkp must lie in [0, min(kmax = {aidx.shape[1]}, {rows})], got [{int(ank.min())}, {int(ank.max())}]")
    h = hashlib.blake2b(digest_size=16)
    for a in arrs:
        h.update(str((a.shape, a.dtype.str)).encode())
        h.update(a.tobytes())
    key = (str(device), rows, h.hexdigest())
    hit = _PACK_CACHE.get(key)
    if hit is None:
        tab, tgt = pack_rows(a1, a2, aidx, ank, rows)
        hit = (torch.from_numpy(tab).to(device), torch.from_numpy(tgt).to(device))
        if len(_PACK_CACHE) >= 256:
            _PACK_CACHE.pop(next(iter(_PACK_CACHE)))
        _PACK_CACHE[key] = hit
    return hit


@torch.no_grad()
def transfer(bank: torch.Tensor, img1: torch.Tensor, img2: torch.Tensor, patch_idx: torch.Tensor, nkp: torch.Tensor, P: int,
             window: int = 5, soft_eval: bool = True, beta: float = 0.02, anno_size: int = 840, split: int = 0,
             layout: str = "cp", sort_pairs: bool = True, packed=None) -> torch.Tensor:
    """Keypoint transfer for a batch of pairs.

    bank [n_images, C, P*P] fp32 (the reference's on-disk [1, C, P, P] maps, flattened); img1/img2/nkp int32 [n];
    patch_idx int32 [n, kmax] (kmax <= 32).  Returns xy fp32 [n, kmax, 2] in the annotation frame.
    split > 0: the bank holds two encoders concatenated on the channel axis ([0, split) and [split, C)), normalised
    separately, concatenated and re-normalised (pck_train_two.py:24-36).
    layout "pc": bank is position-major [n_images, P*P, C] - the towers' own [N, C] token layout; a keypoint's descriptor is one
    contiguous row, which is what the kernel wants (C and split multiples of 4).
    packed (layout "pc" only; default on): key points of the pairs of one target image share 32-row MFMA tiles (pack_rows) - ~2.3x fewer
    tiles and target-map passes on SPair-shaped pair lists; the per-row arithmetic is the unpacked kernel's, bit for bit.  True / None = pack
    here (memoised on the pair list; device-resident index tensors are downloaded for that - one stream sync per call);
    a (rows_tab, tgt) tuple from packed_rows_on() = use that packing (no host work, no sync); False = one tile per pair.
    sort_pairs only concerns the one-tile-per-pair route (the packed route always groups by target image): asking for
    sort_pairs=False together with packing is refused instead of silently ignored.
    """
    # window < 0 (round 6): utils_correspondence.py:321-324 selects apply_gaussian_kernel (sigma = -window), hard-wired to 60 x 60 maps in the
    # reference (np.linspace(0, 59, 60), l.285-288); the kernels weight with the same Gaussian over integer patch coordinates on any grid
    lib = _lib.require_gpu()
    if layout not in ("cp", "pc"):
        raise ValueError("layout must be 'cp' ([n, C, P*P]) or 'pc' ([n, P*P, C])")
    if bank.dtype != torch.float32 or bank.dim() != 3 or bank.shape[2 if layout == "cp" else 1] != P * P:
        raise ValueError("bank must be fp32 [n_images, C, P*P] (layout 'cp') or [n_images, P*P, C] (layout 'pc')")
    C_ = bank.shape[1 if layout == "cp" else 2]
    bank = bank.contiguous()
    n, kmax = patch_idx.shape
    dev = bank.device
    stride = anno_size / P
    if packed is None:
        packed = layout == "pc"
    if packed is not False:
        if layout != "pc":
            raise ValueError("packed key-point tiles need the position-major layout 'pc'")
        if not sort_pairs:
            raise ValueError("sort_pairs=False has no meaning on the packed route (tiles are grouped by target image); pass packed=False")
        if kmax > 32:
            raise ValueError(f"at most 32 key points per pair (patch_idx has {kmax} columns)")
        tab_d, tgt_d = packed if isinstance(packed, tuple) else packed_rows_on(dev, img1, img2, patch_idx, nkp)
        xy = torch.zeros(n, kmax, 2, dtype=torch.float32, device=dev)
        lin = lin_table(P, dev)          # a NAMED tensor: a temporary would return its block to the allocator before the launch is enqueued
        if tgt_d.shape[0]:
            rc = lib.visrep_cscore_transfer_packed(_lib.ptr(bank), _lib.ptr(tab_d), _lib.ptr(tgt_d), _lib.ptr(lin), _lib.ptr(xy), int(tgt_d.shape[0]), kmax,
                                                   P, C_, int(split), int(window), int(soft_eval), float(beta), float(stride), float(stride // 2),
                                                   _lib.stream_ptr())
            _lib.check(rc, "visrep_cscore_transfer_packed")
        return xy
    i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
    img1, img2, patch_idx, nkp = i32(img1), i32(img2), i32(patch_idx), i32(nkp)
    # Launch order = pairs grouped by TARGET image.  One workgroup handles one pair and streams the whole target map (P^2 x C fp32,
    # 1 - 2.4 MB) but only K rows of the source map; a bank of ~1,800 maps (1.9 - 4.3 GB) does not fit the 32 MB of L2, and in dataset
    # order nearly every pair re-fetches its target map from HBM (rocprofv3 PMC, profiles/round2_scores_pmc.md: L2 hit rate 6 %,
    # 22.5 GB of fabric traffic per launch).  An image is the target of ~7 pairs: run together they fetch it once.
    order = torch.argsort(img2.to(torch.int64), stable=True) if sort_pairs and n > 1 else None
    if order is not None:
        img1, img2, patch_idx, nkp = (t.index_select(0, order).contiguous() for t in (img1, img2, patch_idx, nkp))
    xy = torch.zeros(n, kmax, 2, dtype=torch.float32, device=dev)
    lin = lin_table(P, dev)
    rc = lib.visrep_cscore_transfer(_lib.ptr(bank), _lib.ptr(img1), _lib.ptr(img2), _lib.ptr(patch_idx), _lib.ptr(nkp),
                                    _lib.ptr(lin), _lib.ptr(xy), n, kmax, P, C_, int(split), int(window), int(soft_eval),
                                    float(beta), float(stride), float(stride // 2), 0 if layout == "cp" else 1, _lib.stream_ptr())
    _lib.check(rc, "visrep_cscore_transfer")
    if order is not None:
        out = torch.empty_like(xy)
        out.index_copy_(0, order, xy)                               # back to the caller's pair order
        return out
    return xy


@torch.no_grad()
def pck_counts(xy: torch.Tensor, kps1: torch.Tensor, kps2: torch.Tensor, thresholds: torch.Tensor, nkp: torch.Tensor,
               alphas=(0.1, 0.05, 0.01)) -> torch.Tensor:
    """counts int32 [n, 4] = hits@alpha0..2, n_visible (pck_train.py:101,149-163)."""
    lib = _lib.require_gpu()
    dev = xy.device
    n, kmax, _ = xy.shape
    kps1 = kps1.to(device=dev, dtype=torch.float32).contiguous()
    kps2 = kps2.to(device=dev, dtype=torch.float32).contiguous()
    thr = thresholds.to(device=dev, dtype=torch.float64).contiguous()
    nkp = nkp.to(device=dev, dtype=torch.int32).contiguous()
    counts = torch.zeros(n, 4, dtype=torch.int32, device=dev)
    a = (C.c_float * 3)(*[float(np.float32(x)) for x in alphas])
    xy = xy.contiguous()
    rc = lib.visrep_pck_count(_lib.ptr(xy), _lib.ptr(kps1), _lib.ptr(kps2), _lib.ptr(thr), _lib.ptr(nkp), n, kmax, a,
                              _lib.ptr(counts), _lib.stream_ptr())
    _lib.check(rc, "visrep_pck_count")
    return counts


@torch.no_grad()
def mutual_nn_distance(bank: torch.Tensor, img1: torch.Tensor, img2: torch.Tensor, P: int, chunk: int = 2048, eps: float = 1e-10) -> torch.Tensor:
    """get_distance_mutual_nn (utils_correspondence.py:54-73) for a batch of pairs: the mean Euclidean distance of the mutual nearest
    neighbours of the two L2-normalised descriptor sets (pck_train.py:24-29 normalisation).  bank: position-major fp32 [n_images, P*P, C]
    RAW maps; returns fp32 [n_pairs] (nan for a pair without mutual neighbours, as torch's empty mean).  Pairs are processed in chunks:
    the [chunk, P^2, P^2] Gram buffer (exact-fp32 MFMA, visrep_gram_pairs_f32) is the only large temporary."""
    lib = _lib.require_gpu()
    if bank.dtype != torch.float32 or bank.dim() != 3 or bank.shape[1] != P * P:
        raise ValueError("bank must be fp32 [n_images, P*P, C] (position-major)")
    bank = bank.contiguous()
    dev = bank.device
    n_img, PP, C_ = bank.shape
    if PP % 4 or C_ % 4 or PP > 4096:
        raise ValueError("mutual_nn_distance needs P*P and C multiples of 4 and P <= 64")
    i1 = img1.to(device=dev, dtype=torch.int32).contiguous()
    i2 = img2.to(device=dev, dtype=torch.int32).contiguous()
    n = i1.shape[0]
    rn = torch.empty(n_img, PP, dtype=torch.float32, device=dev)
    _lib.check(lib.visrep_row_rnorm_f32(_lib.ptr(bank), n_img * PP, C_, float(eps), _lib.ptr(rn), _lib.stream_ptr()), "visrep_row_rnorm_f32")
    out = torch.empty(n, dtype=torch.float32, device=dev)
    chunk = max(1, min(chunk, (2 << 30) // (4 * PP * PP)))                  # the Gram buffer stays <= 2 GiB (60 x 60 maps: 51.8 MB per pair)
    gram = torch.empty(min(chunk, max(n, 1)), PP, PP, dtype=torch.float32, device=dev)
    for s in range(0, n, chunk):
        a, b = i1[s:s + chunk].contiguous(), i2[s:s + chunk].contiguous()
        m = a.shape[0]
        _lib.check(lib.visrep_gram_pairs_f32(_lib.ptr(bank), _lib.ptr(a), _lib.ptr(b), m, PP, C_, _lib.ptr(gram), _lib.stream_ptr()), "visrep_gram_pairs_f32")
        r1, r2 = rn.index_select(0, a.long()).contiguous(), rn.index_select(0, b.long()).contiguous()
        _lib.check(lib.visrep_mutual_nn_distance(_lib.ptr(gram), _lib.ptr(r1), _lib.ptr(r2), m, PP, float(eps), C.c_void_p(out.data_ptr() + 4 * s), _lib.stream_ptr()),
                   "visrep_mutual_nn_distance")
    return out


@torch.no_grad()
def masked_nn_distance(desc1: torch.Tensor, desc2: torch.Tensor, mask1: torch.Tensor, mask2: torch.Tensor, resolution: int = 64) -> torch.Tensor:
    """The mask-based flip distance of ADAPT_FLIP without MUTUAL_NN (utils_correspondence.py:22-52 get_distance, called at pck_train.py:122-124)
    on the device.  desc* [1 | -, P^2, C] L2-normalised descriptors of one image each (the reference hard-codes P = 60; any square grid here),
    mask* [H, W] binary masks (convert_to_binary_mask).  Lines 23-37 - bilinear resize of the maps and nearest resize of the masks to
    resolution^2, masking, `== 0 -> -100000` - are torch glue on the device; the search of lines 41-50 (4096 x 4096 x C differences) is
    visrep_masked_nn_min_f32.  Returns a 0-dim fp32 tensor (nan for an empty source mask, like the reference's empty mean)."""
    import torch.nn.functional as F
    if mask1 is None or mask2 is None:
        raise AttributeError("'NoneType' object has no attribute 'unsqueeze' (get_distance needs the `_mask.png` of both images: "
                             "prepare_feature_paths_and_load found none, as in the reference, utils_correspondence.py:27)")
    d1, d2 = desc1.reshape(-1, desc1.shape[-1]), desc2.reshape(-1, desc2.shape[-1])
    P = int(round(d1.shape[0] ** 0.5))
    if P * P != d1.shape[0] or d2.shape != d1.shape:
        raise ValueError("descriptors must be two [P*P, C] maps of one square grid")
    lib = _lib.require_gpu()
    dev = d1.device
    R, C_ = int(resolution), d1.shape[1]
    m1 = F.interpolate(mask1.to(dev).float()[None, None], size=(R, R), mode="nearest")[0, 0]
    m2 = F.interpolate(mask2.to(dev).float()[None, None], size=(R, R), mode="nearest")[0, 0]
    f1 = F.interpolate(d1.float().t().reshape(1, C_, P, P), size=(R, R), mode="bilinear")[0] * m1[None]
    f2 = F.interpolate(d2.float().t().reshape(1, C_, P, P), size=(R, R), mode="bilinear")[0] * m2[None]
    f1 = torch.where(f1 == 0, torch.full_like(f1, -100000.0), f1).reshape(C_, R * R).t()
    f2 = torch.where(f2 == 0, torch.full_like(f2, -100000.0), f2).reshape(C_, R * R).t()
    pad = (-C_) % 4                                                   # zero channels on both sides add nothing to a difference
    src = f1[m1.reshape(-1) == 1]
    if pad:
        src, f2 = F.pad(src, (0, pad)), F.pad(f2, (0, pad))
    src, tgt = src.contiguous(), f2.contiguous()
    if src.shape[0] == 0:
        return torch.tensor(float("nan"), device=dev)
    d2min = torch.empty(src.shape[0], dtype=torch.float32, device=dev)
    _lib.check(lib.visrep_masked_nn_min_f32(_lib.ptr(src), _lib.ptr(tgt), src.shape[0], tgt.shape[0], src.shape[1], _lib.ptr(d2min), _lib.stream_ptr()),
               "visrep_masked_nn_min_f32")
    return d2min.sqrt().mean()


# ------------------------------------------------------------------------------------------------ host twins (explicit device "cpu" only)
@torch.no_grad()
def transfer_cpu(bank: torch.Tensor, img1, img2, patch_idx, nkp, P: int, window: int = 5, soft_eval: bool = True, beta: float = 0.02,
                 anno_size: int = 840, split: int = 0, layout: str = "cp", threads: int = 0) -> torch.Tensor:
    """transfer() on HOST cores (visrep_cscore_transfer_cpu, csrc/host_twins.hip: plain C++ fp32) - the `*_cpu` twin of SURVEY §8b.  CPU
    tensors in, xy fp32 [n, kmax, 2] out; same arguments and semantics as transfer(); never used as a fallback."""
    lib = _lib.load()
    if layout not in ("cp", "pc"):
        raise ValueError("layout must be 'cp' ([n, C, P*P]) or 'pc' ([n, P*P, C])")
    if bank.device.type != "cpu" or bank.dtype != torch.float32 or bank.dim() != 3 or bank.shape[2 if layout == "cp" else 1] != P * P:
        raise ValueError("bank must be a CPU fp32 tensor [n_images, C, P*P] (layout 'cp') or [n_images, P*P, C] (layout 'pc')")
    C_ = bank.shape[1 if layout == "cp" else 2]
    bank = bank.contiguous()
    i32 = lambda t: torch.as_tensor(t).to(device="cpu", dtype=torch.int32).contiguous()
    img1, img2, patch_idx, nkp = i32(img1), i32(img2), i32(patch_idx), i32(nkp)
    n, kmax = patch_idx.shape
    stride = anno_size / P
    lin = torch.tensor(np.linspace(-1, 1, P)).float()
    xy = torch.zeros(n, kmax, 2, dtype=torch.float32)
    rc = lib.visrep_cscore_transfer_cpu(_lib.ptr(bank), _lib.ptr(img1), _lib.ptr(img2), _lib.ptr(patch_idx), _lib.ptr(nkp), _lib.ptr(lin), _lib.ptr(xy),
                                        n, kmax, P, C_, int(split), int(window), int(soft_eval), float(beta), float(stride), float(stride // 2),
                                        0 if layout == "cp" else 1, int(threads))
    _lib.check(rc, "visrep_cscore_transfer_cpu")
    return xy


@torch.no_grad()
def pck_counts_cpu(xy: torch.Tensor, kps1: torch.Tensor, kps2: torch.Tensor, thresholds: torch.Tensor, nkp: torch.Tensor,
                   alphas=(0.1, 0.05, 0.01)) -> torch.Tensor:
    """pck_counts() on the host (visrep_pck_count_cpu): counts int32 [n, 4] = hits@alpha0..2, n_visible."""
    lib = _lib.load()
    n, kmax, _ = xy.shape
    f32 = lambda t: t.to(device="cpu", dtype=torch.float32).contiguous()
    xy, kps1, kps2 = f32(xy), f32(kps1), f32(kps2)
    thr = thresholds.to(device="cpu", dtype=torch.float64).contiguous()
    nk = nkp.to(device="cpu", dtype=torch.int32).contiguous()
    counts = torch.zeros(n, 4, dtype=torch.int32)
    a = (C.c_float * 3)(*[float(np.float32(x)) for x in alphas])
    _lib.check(lib.visrep_pck_count_cpu(_lib.ptr(xy), _lib.ptr(kps1), _lib.ptr(kps2), _lib.ptr(thr), _lib.ptr(nk), n, kmax, a, _lib.ptr(counts)),
               "visrep_pck_count_cpu")
    return counts
