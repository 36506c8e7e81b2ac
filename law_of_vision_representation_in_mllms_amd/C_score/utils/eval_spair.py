"""Post-hoc metrics over result.pkl — C_score/utils/eval_spair.py: re-reads the annotations of every scored pair and
recomputes per-image / per-keypoint PCK, optionally restricted to the geometry-aware key points.  logger.log_geo_stats
uses convert_all_results + get_img_result(geo=True) for the "Weighted Per image geo-aware PCK" line (logger.py:88-91).
Host code on K <= 30 points per pair, as in the reference; thresholds here are float32 products (eval_spair.py:119,328),
unlike compute_pck's float64 ones."""
import json

import numpy as np
import torch

from .utils_dataset import preprocess_kps_pad
from .utils_geoware import AP10K_GEO_AWARE, SPAIR_FLIP, SPAIR_GEO_AWARE, geo_aware_points

ALPHA = (0.1, 0.05, 0.01)


def _anno(image_path):
    with open(image_path.replace("JPEGImages", "ImageAnnotation").replace("jpg", "json")) as f:
        return json.load(f)


def _spair_kps(anno, size):
    k = torch.zeros(30, 3)
    for i in range(30):
        pt = anno["kps"][str(i)]
        if pt is not None:
            k[i, :2] = torch.Tensor(pt).float()
            k[i, 2] = 1
    return preprocess_kps_pad(k, anno["image_width"], anno["image_height"], size)


def load_spair_data(path1, path2, size=256):
    a1, a2 = _anno(path1), _anno(path2)
    source_kps = _spair_kps(a1, size)[0]
    target_kps, _, _, trg_scale = _spair_kps(a2, size)
    bb = np.asarray(a2["bndbox"])
    thresholds = max(bb[3] - bb[1], bb[2] - bb[0]) * trg_scale
    az1, az2 = a1["azimuth_id"], a2["azimuth_id"]
    mirror = 1 if (az1 <= 3) != (az2 <= 3) else 0
    return source_kps, target_kps, thresholds, az1, az2, mirror


def load_ap10k_data(path1, path2, size=256):
    out = []
    for a in (_anno(path1), _anno(path2)):
        k = torch.tensor(a["keypoints"]).view(-1, 3).float()
        k[:, -1] /= 2
        out.append(preprocess_kps_pad(k, a["width"], a["height"], size) + (np.asarray(a["bbox"]),))
    (source_kps, *_), (target_kps, _, _, trg_scale, tb) = out
    return source_kps, target_kps, max(tb[3], tb[2]) * trg_scale


def _convert(result, dataset):
    rows = []
    for item in result:
        src_fn, trg_fn = item["src_fn"], item["trg_fn"]
        category = src_fn.split("/")[-2]
        pred = torch.tensor(item["src_kpts_pred"][:, [1, 0]]).float()
        row = {"src_fn": src_fn, "trg_fn": trg_fn, "category": category}
        if dataset == "ap10k":
            src_kps, trg_kps, thr = load_ap10k_data(src_fn, trg_fn, item["resize_resolution"])
            groups, n_slots = AP10K_GEO_AWARE, 17
        else:
            src_kps, trg_kps, thr, az1, az2, mirror = load_spair_data(src_fn, trg_fn, item["resize_resolution"])
            groups, n_slots = SPAIR_GEO_AWARE[category], 30
            row.update(az=min(abs(az1 - az2), 8 - abs(az1 - az2)), mirror=mirror)
        vis = src_kps[:, 2] * trg_kps[:, 2] > 0
        if dataset != "ap10k":
            # eval_spair.py:164-175: mutually visible members of the left/right groups of which the SOURCE image shows at least two
            # (the AP-10k converter has no flip groups: flip=True raises KeyError there, as in the reference)
            row.update(flip_idx=geo_aware_points(SPAIR_FLIP[category], vis, src_kps[:, 2] > 0))
        row.update(src_kps=src_kps[:, [1, 0]], gt_kps=trg_kps[:, [1, 0]], pred_kps=pred, thresholds=torch.tensor(thr).float(),
                   used_points=[i for i in range(n_slots) if vis[i]], geo_aware_idx=geo_aware_points(groups, vis, trg_kps[:, 2] > 0))
        rows.append(row)
    return rows


def convert_all_results(result):
    return _convert(result, "spair")


def convert_all_results_ap10k(result):
    return _convert(result, "ap10k")


def _selected(all_results, cls, az):
    return [r for r in all_results if (cls is None or r["category"] == cls) and (az is None or r["az"] == az)]


def _hits(item, idx):
    alpha = torch.tensor(ALPHA)
    err = torch.abs(item["gt_kps"][idx] - item["pred_kps"][idx]).norm(dim=-1)
    return err.unsqueeze(0) < alpha.unsqueeze(1) * item["thresholds"].repeat(len(idx)).unsqueeze(0)


def get_std_result(all_results, cls=None, geo=False, flip=False, az=None):
    """Key-point-level PCK over the selected pairs -> (correct[3], n_keypoints); geo / flip restrict it to the geometry-aware / the
    left-right group key points (eval_spair.py:333-336: geo wins when both are set)."""
    hits = [_hits(r, r["geo_aware_idx"] if geo else r["flip_idx"] if flip else r["used_points"]) for r in _selected(all_results, cls, az)]
    hits = torch.cat(hits, dim=1)
    return hits.sum(dim=-1).float() / hits.shape[1], hits.shape[1]


def get_img_result(all_results, cls=None, geo=False, flip=False, az=None):
    """Mean over pairs of the per-pair PCK -> (correct[3], n_pairs counted); geo skips pairs without geometry-aware points."""
    per_img = []
    for r in _selected(all_results, cls, az):
        idx = r["geo_aware_idx"] if geo else r["flip_idx"] if flip else r["used_points"]
        if (geo or flip) and len(idx) == 0:
            continue
        h = _hits(r, idx)
        per_img.append(h.sum(dim=-1).float() / len(idx))
    if not per_img:
        return torch.zeros(3), 0
    return torch.stack(per_img, dim=0).mean(dim=0), len(per_img)
