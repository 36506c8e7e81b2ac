"""C-score oracle (CPU).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Restates the zero-shot GeoAware-SC evaluation arithmetic of the reference:

  * normalize_feats                     C_score/pck_train.py:24-29
  * two-encoder normalize_feats         C_score/pck_train_two.py:24-36
  * kpts_to_patch_idx                   C_score/utils/utils_correspondence.py:384-388
  * calculate_keypoint_transformation   C_score/utils/utils_correspondence.py:345-382
  * get_flow / soft_argmax / softmax_with_temperature /
    unnormalise_and_convert_mapping_to_flow
                                        C_score/utils/utils_correspondence.py:297-337,234-256,226-232,258-277
  * get_distance (mask-based)           C_score/utils/utils_correspondence.py:22-52
  * per-image PCK                       C_score/pck_train.py:101,142-163
  * per-keypoint PCK                    C_score/pck_train.py:210-226
  * weighted aggregation                C_score/utils/logger.py:22-72

Only the K keypoint rows of the [P^2, P^2] similarity are evaluated (the
reference computes all rows then gathers, utils_correspondence.py:360,367-368);
row results are independent so this is the same arithmetic.

SOFT_EVAL_WINDOW < 0 is the Gaussian-kernel soft-argmax (apply_gaussian_kernel, utils_correspondence.py:278-295; round 6).

Window soft-argmax semantics that must be preserved: entries
outside the (2w+1)^2 window (clamped at the borders) are ZERO, not -inf, and
still take part in the softmax over all P^2 targets.
"""
from __future__ import annotations

import numpy as np
import torch


def normalize_feats(feat: torch.Tensor, epsilon: float = 1e-10) -> torch.Tensor:
    # pck_train.py:24-29 — feat [..., P^2, C]
    norms = torch.linalg.norm(feat, dim=-1, keepdim=True)
    return feat / (norms + epsilon)


def normalize_feats_two(feat: torch.Tensor, split: int, epsilon: float = 1e-10) -> torch.Tensor:
    # pck_train_two.py:24-36 — normalise the two encoders' channels separately, concat, renormalise
    a, b = feat[..., :split], feat[..., split:]
    a = a / (torch.linalg.norm(a, dim=-1, keepdim=True) + epsilon)
    b = b / (torch.linalg.norm(b, dim=-1, keepdim=True) + epsilon)
    f = torch.cat([a, b], dim=-1)
    return f / (torch.linalg.norm(f, dim=-1, keepdim=True) + epsilon)


def kpts_to_patch_idx(kps: torch.Tensor, num_patches: int, anno_size: int = 840) -> np.ndarray:
    # utils_correspondence.py:384-388 (numpy float64 * float32 -> float64, truncation to int32)
    y, x = kps[:, 1].numpy(), kps[:, 0].numpy()
    yp = (num_patches / anno_size * y).astype(np.int32)
    xp = (num_patches / anno_size * x).astype(np.int32)
    return num_patches * yp + xp


def descriptors_from_map(feat_map: torch.Tensor, num_patches: int) -> torch.Tensor:
    # pck_train.py:38-39,53 — [1, C, P, P] -> [1, P^2, C], L2-normalised
    desc = feat_map.reshape(1, 1, -1, num_patches ** 2).permute(0, 1, 3, 2)
    return normalize_feats(desc[0])


def window_soft_argmax_rows(sim_rows: torch.Tensor, P: int, window: int, beta: float = 0.02):
    """Rows of get_flow(): sim_rows [K, P^2] -> (x, y) patch coords [K] each (float32)."""
    K = sim_rows.shape[0]
    corr = sim_rows.clone()
    if window > 0:
        am = torch.argmax(corr, dim=-1)
        mx, my = am % P, am // P
        xs = torch.arange(P)
        inx = (xs[None, :] >= (mx[:, None] - window).clamp(0, P - 1)) & (xs[None, :] <= (mx[:, None] + window).clamp(0, P - 1))
        iny = (xs[None, :] >= (my[:, None] - window).clamp(0, P - 1)) & (xs[None, :] <= (my[:, None] + window).clamp(0, P - 1))
        mask = (iny[:, :, None] & inx[:, None, :]).reshape(K, P * P).to(corr.dtype)
        corr = corr * mask
    elif window < 0:
        # "kernel soft-argmax" (get_flow :321-324 -> apply_gaussian_kernel :278-295): every similarity is weighted by a Gaussian (sigma = -window
        # patches) around the row's argmax target before the softmax.  The reference builds the kernel's coordinates with linspace(0, 59, 60),
        # i.e. it only runs on 60 x 60 maps; integer patch coordinates 0 .. P - 1 are the same thing at P = 60 and the generalisation elsewhere.
        am = torch.argmax(corr, dim=-1)
        mx, my = (am % P).float(), (am // P).float()
        xs = torch.arange(P).float()
        sigma = float(-window)
        g = torch.exp(-((xs[None, None, :] - mx[:, None, None]) ** 2 + (xs[None, :, None] - my[:, None, None]) ** 2) / (2 * sigma ** 2))    # [K, ty, tx]
        corr = corr * g.reshape(K, P * P)
    M = corr.max(dim=1, keepdim=True).values
    e = torch.exp((corr - M) / beta)
    p = (e / e.sum(dim=1, keepdim=True)).view(K, P, P)            # [K, ty, tx]
    lin = torch.tensor(np.linspace(-1, 1, P)).float()
    gx = (p.sum(dim=1) * lin[None, :]).sum(dim=1)                  # marginal over ty, expectation over tx
    gy = (p.sum(dim=2) * lin[None, :]).sum(dim=1)
    x = (gx + 1) * (P - 1) / 2.0
    y = (gy + 1) * (P - 1) / 2.0
    return x, y


def keypoint_transfer(desc1: torch.Tensor, desc2: torch.Tensor, patch_idx, P: int,
                      anno_size: int = 840, soft_eval: bool = True, window: int = 5,
                      beta: float = 0.02) -> torch.Tensor:
    """calculate_keypoint_transformation: desc* [1, P^2, C] (normalised) -> [K, 2] (x, y)."""
    idx = torch.as_tensor(np.asarray(patch_idx), dtype=torch.long)
    rows = desc1[0][idx] @ desc2[0].t()                             # [K, P^2]
    stride = anno_size / P
    if soft_eval:
        x, y = window_soft_argmax_rows(rows, P, window, beta)
        x, y = x.clamp(0, P - 1), y.clamp(0, P - 1)
    else:
        nn = torch.max(rows, dim=-1).indices
        y, x = nn // P, nn % P
    nn_x = x * stride + stride // 2
    nn_y = y * stride + stride // 2
    return torch.stack([nn_x, nn_y]).permute(1, 0)


def pair_pck(pred_xy: torch.Tensor, kps1: torch.Tensor, kps2: torch.Tensor, threshold,
             alphas=(0.1, 0.05, 0.01), anno_size: int = 840):
    """pck_train.py:101,149-163 — returns (correct[3] per-image means, n_visible, hits[3, n_vis])."""
    vis = kps1[:, 2] * kps2[:, 2] > 0
    gt = kps2[vis][:, [1, 0]]
    prd = pred_xy[vis][:, [1, 0]]
    alpha = torch.tensor(list(alphas))
    err = (gt - prd).norm(dim=-1).unsqueeze(0).repeat(len(alphas), 1)
    if threshold is not None:
        bbox = torch.tensor(threshold, dtype=torch.float64).repeat(int(vis.sum()))
        hits = err < alpha.unsqueeze(-1) * bbox.unsqueeze(0)
    else:
        hits = err < alpha.unsqueeze(-1) * anno_size
    return hits.float().mean(dim=-1), int(vis.sum()), hits


def category_pck(feats, pairs, kps, thresholds, P, anno_size=840, window=5, soft_eval=True):
    """compute_pck (pck_train.py:57-245) for one category on in-memory features.

    feats: list of [1, C, P, P] maps (one per file slot, 2 per pair); kps [2N, K, 3].
    Returns (per-kpt correct[3]+[n_kpts], img_correct[3]+[N], preds list).
    """
    N = len(pairs)
    img_acc = [[], [], []]
    all_hits = []
    preds = []
    for i in range(N):
        k1, k2 = kps[2 * i], kps[2 * i + 1]
        d1 = descriptors_from_map(feats[2 * i], P)
        d2 = descriptors_from_map(feats[2 * i + 1], P)
        idx = kpts_to_patch_idx(k1, P, anno_size)
        pred = keypoint_transfer(d1, d2, idx, P, anno_size, soft_eval, window)
        preds.append(pred)
        thr = None if thresholds is None else thresholds[i]
        c, nv, hits = pair_pck(pred, k1, k2, thr, anno_size=anno_size)
        for a in range(3):
            img_acc[a].append(c[a].item())
        all_hits.append(hits)
    img_correct = torch.tensor(img_acc).mean(dim=-1).tolist() + [N]
    hits = torch.cat(all_hits, dim=1)
    n_k = hits.shape[1]
    correct = (hits.sum(dim=-1) / n_k).tolist() + [n_k]
    return correct, img_correct, preds


def weighted_pcks(per_cat_values, weights):
    # logger.py:61-72
    v = np.asarray(per_cat_values, dtype=np.float64)   # [n_cat, 3]
    return tuple(np.average(v[:, j], weights=weights) for j in range(3))


# ---------------------------------------------------------------------------------------------------- ADAPT_FLIP (pck_train.py:111-126)
def mutual_nn_distance(desc1: torch.Tensor, desc2: torch.Tensor) -> torch.Tensor:
    """get_distance_mutual_nn, utils_correspondence.py:54-73: desc* [1, P^2, C] normalised descriptors -> mean distance of the
    mutual nearest neighbours (0-dim tensor; nan if there are none)."""
    d = torch.cdist(desc1, desc2)[0]
    nn12, nn21 = torch.argmin(d, dim=1), torch.argmin(d, dim=0)
    mutual = nn21[nn12] == torch.arange(d.shape[0])
    return torch.min(d, dim=1)[0][mutual].mean()


def masked_nn_distance(desc1: torch.Tensor, desc2: torch.Tensor, mask1: torch.Tensor, mask2: torch.Tensor, resolution: int = 64) -> torch.Tensor:
    """utils_correspondence.py:22-52 get_distance (the flip decision of ADAPT_FLIP without MUTUAL_NN, pck_train.py:122-124): descriptors
    [1, P^2, C] -> [C, P, P] maps (the reference hard-codes P = 60; here P = sqrt(P^2)), bilinearly resized to resolution^2 (align_corners =
    False), the binary masks [H, W] nearest-resized to the same grid; features outside a mask - and every element that is exactly 0, inside or
    outside (line 36-37 compares elementwise) - become -100000; for every SOURCE cell inside its mask the Euclidean distance to the nearest
    target cell (difference first, then the norm: line 47); the mean of those distances (nan when the source mask is empty, as in the
    reference: the mean of an empty tensor)."""
    import torch.nn.functional as F
    P = int(round(desc1.shape[-2] ** 0.5))
    R = resolution
    m1 = F.interpolate(mask1.float()[None, None], size=(R, R), mode="nearest")[0, 0]
    m2 = F.interpolate(mask2.float()[None, None], size=(R, R), mode="nearest")[0, 0]
    f1 = F.interpolate(desc1.reshape(-1, desc1.shape[-1]).t().reshape(1, -1, P, P).float(), size=(R, R), mode="bilinear")[0]
    f2 = F.interpolate(desc2.reshape(-1, desc2.shape[-1]).t().reshape(1, -1, P, P).float(), size=(R, R), mode="bilinear")[0]
    f1, f2 = f1 * m1[None], f2 * m2[None]
    f1 = torch.where(f1 == 0, torch.full_like(f1, -100000.0), f1)
    f2 = torch.where(f2 == 0, torch.full_like(f2, -100000.0), f2)
    s2d, t2d = f1.reshape(f1.shape[0], -1).t(), f2.reshape(f2.shape[0], -1).t()
    src = s2d[m1.reshape(-1) == 1]
    if src.shape[0] == 0:
        return torch.tensor(float("nan"))
    # lines 44-49 loop over the source cells with norm(tgt - src[i]); cdist without the matmul shortcut is the same difference-first
    # arithmetic for all of them at once (the Gram form would cancel the 1e10-sized squares of the -100000 entries)
    return torch.cdist(src, t2d, compute_mode="donot_use_mm_for_euclid_dist").min(dim=1).values.mean()


def permute_indices(flip_list, vis=None):
    """utils_geoware.py:151-189: twins of a group trade places when (vis is None or) the whole group is visible."""
    flat = []
    for item in flip_list:
        flat += item if isinstance(item, list) else [item]
    idx = list(range(max(flat) + 1))
    for item in flip_list:
        if isinstance(item, list) and (vis is None or all(bool(vis[i]) for i in item)):
            for n, i in enumerate(item):
                idx[i] = item[(n + 1) % len(item)]
    return idx


def flip_keypoints(kps: torch.Tensor, img_size, permute_list=None) -> torch.Tensor:
    # utils_geoware.py:199-204
    out = kps.detach().clone()
    out[:, 0] = img_size - out[:, 0]
    return out if permute_list is None else out[permute_list]


def adapt_flip_prediction(pred, pred_flip, kps1, kps2, flip_dist, orig_dist, permute_list, anno_size=840):
    """optimized_kps_1_to_2, utils_geoware.py:269-279."""
    vis = kps1[:, 2] * kps2[:, 2] > 0
    masked = kps1 * vis.unsqueeze(-1).float()
    flipped = flip_keypoints(masked, anno_size, permute_indices(permute_list, None))
    vis_flip = flipped[:, 2] * kps2[:, 2] * kps1[:, 2] > 0
    out = pred.clone()
    if flip_dist < orig_dist:
        out[vis_flip] = pred_flip[vis_flip]
    return out


# ---------------------------------------------------------------------------------------------------- supervised post-processor
def aggregation_network(x: torch.Tensor, sd: dict, feature_dims, num_norm_groups: int, eps: float = 1e-5) -> torch.Tensor:
    """AggregationNetwork.forward in evaluation mode (projection_network.py:89-125) over one BottleneckBlock per channel group
    (model_utils/resnet.py:174-286): x [B, sum(dims), H, W], sd = the module's state_dict (fp32 tensors)."""
    import torch.nn.functional as F
    mix = torch.softmax(sd["mixing_weights"].float(), dim=0)
    out, start = None, 0

    def conv_gn(t, pre, pad):
        t = F.conv2d(t, sd[f"{pre}.weight"], None, 1, pad)
        return F.group_norm(t, num_norm_groups, sd[f"{pre}.norm.weight"], sd[f"{pre}.norm.bias"], eps)
    for i in range(mix.shape[0]):
        l = i % len(feature_dims)
        pre = f"bottleneck_layers.{l}.0"
        feats = x[:, start:start + feature_dims[l]]
        start += feature_dims[l]
        h = F.relu(conv_gn(feats, f"{pre}.conv1", 0))
        h = F.relu(conv_gn(h, f"{pre}.conv2", 1))
        h = conv_gn(h, f"{pre}.conv3", 0)
        sc = conv_gn(feats, f"{pre}.shortcut", 0) if f"{pre}.shortcut.weight" in sd else feats
        y = mix[i] * F.relu(h + sc)
        out = y if out is None else out + y
    return out
