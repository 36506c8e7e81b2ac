"""C score (zero-shot SPair-71k PCK) on MI355X — drop-in for the evaluation half of C_score/pck_train.py.

Same public functions and argument meaning:
  normalize_feats(args, feat, epsilon=1e-10)                                          pck_train.py:24-29
  prepare_feature_paths_and_load / get_patch_descriptors                               pck_train.py:31-55
  compute_pck(args, save_path, aggre_net, files, kps, category=None, used_points=None, thresholds=None)
        -> (correct, geo_score, out_results, img_correct)                              pck_train.py:57-245
  eval(args, aggre_net, save_path, split='val') -> (pck_010, pck_005, pck_001, total_out_results)   :315-340
  main(args) + the same argparse flags / yaml keys                                     :342-442

What is different underneath: compute_pck loads every DISTINCT image's feature map once into one device-resident bank
(the reference re-`torch.load`s both maps for every pair and decodes/resizes both JPEGs only to discard them), then
runs ONE keypoint-transfer launch and ONE PCK-count launch for all pairs of the category (csrc/cscore.hip).  With
torch.distributed initialised the pairs of a category are sharded over ranks in contiguous blocks and the hit counters
are all-reduced (RCCL) — per-image means stay exact because counts, not means, are reduced.

COMPUTE_GEOAWARE_METRICS (pck_train.py:68-80,169-192,231-243) reuses the count launch on the geometry-aware key points.
Not built (fail loudly): training (`DO_EVAL` false), ADAPT_FLIP.
"""
import argparse
import os
import pickle
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import cscore_ops
from .model_utils.projection_network import DummyAggregationNetwork
from .utils.logger import get_logger, load_config, log_geo_stats, log_weighted_pcks, update_geo_stats, update_stats
from .utils.utils_correspondence import (calculate_keypoint_transformation, convert_to_binary_mask, get_distance, get_distance_mutual_nn,  # noqa: F401 (API parity)
                                         kpts_to_patch_idx, kpts_to_patch_idx_batch)
from .utils.utils_dataset import get_dataset_info, load_eval_data
from .utils.utils_geoware import AP10K_GEO_AWARE, SPAIR_GEO_AWARE, filtered_groups, geo_aware_points, renumber_used_points

device = 'cuda' if torch.cuda.is_available() else 'cpu'
logger = get_logger()


def normalize_feats(args, feat, epsilon=1e-10):
    norms = torch.linalg.norm(feat, dim=-1)[:, :, None]
    return feat / (norms + epsilon)


def _feature_path(img_path, flip, ensemble, model):
    # pck_train.py:33-37
    feature_base = img_path.replace('JPEGImages', 'features').replace('.jpg', '')
    suffix_flip = '_flip' if flip else ''
    ensemble_folder = f'features_ensemble{ensemble}' if ensemble > 1 else 'features'
    return f"{feature_base}_{model}{suffix_flip}.pt".replace('features', ensemble_folder)


def _mask_path(img_path, flip):
    # pck_train.py:33-36: `<feature_base>_mask[_flip].png` (next to the features, never in the ensemble folder)
    return f"{img_path.replace('JPEGImages', 'features').replace('.jpg', '')}_mask{'_flip' if flip else ''}.png"


def _load_mask(img_path, flip):
    path = _mask_path(img_path, flip)
    return convert_to_binary_mask(path) if os.path.exists(path) else None          # pck_train.py:40-43


def prepare_feature_paths_and_load(aggre_net, img_path, flip, ensemble, num_patches, device, model):
    desc = torch.load(_feature_path(img_path, flip, ensemble, model), map_location="cpu").to(device)
    desc = aggre_net(desc).reshape(1, 1, -1, num_patches ** 2).permute(0, 1, 3, 2)
    return desc, _load_mask(img_path, flip)


def get_patch_descriptors(args, aggre_net, num_patches, files, pair_idx, flip=False, flip2=False, img1=None, img2=None,
                          device='cuda'):
    img1_desc, mask1 = prepare_feature_paths_and_load(aggre_net, files[pair_idx * 2], flip, args.ENSEMBLE, num_patches, device, args.MODEL)
    img2_desc, mask2 = prepare_feature_paths_and_load(aggre_net, files[pair_idx * 2 + 1], flip2, args.ENSEMBLE, num_patches, device, args.MODEL)
    return normalize_feats(args, img1_desc[0]), normalize_feats(args, img2_desc[0]), mask1, mask2


_renumber_used_points = renumber_used_points


def _dist():
    d = torch.distributed
    return d if d.is_available() and d.is_initialized() else None


def build_feature_bank(args, aggre_net, files, num_patches, dev, models=None, flip=False):
    """Distinct images of a category -> ([n_img, C, P^2] fp32 device bank, per-file-slot bank index).

    `models`: feature-file suffixes; two of them (pck_train_two.py) are concatenated on the channel axis - the raw maps,
    the per-encoder normalisation happens in the transfer kernel (split = channels of the first one)."""
    models = (args.MODEL,) if models is None else tuple(models)
    uniq, slot = {}, []
    for f in files:
        if f not in uniq:
            uniq[f] = len(uniq)
        slot.append(uniq[f])
    def load(f):
        per_model = [aggre_net(torch.load(_feature_path(f, flip, args.ENSEMBLE, m), map_location="cpu")).reshape(-1, num_patches ** 2).float()
                     for m in models]
        return torch.cat(per_model, 0) if len(per_model) > 1 else per_model[0]
    # each distinct map is read ONCE (the reference reloads both maps of every pair, pck_train.py:31-39), a few files in flight
    with ThreadPoolExecutor(max_workers=8) as pool:
        maps = list(pool.map(load, uniq))
    return torch.stack(maps).to(dev), np.asarray(slot, dtype=np.int32)


def compute_pck(args, save_path, aggre_net, files, kps, category=None, used_points=None, thresholds=None, bank=None):
    return _compute_pck(args, save_path, aggre_net, files, kps, category, used_points, thresholds, bank, models=(args.MODEL,))


def _compute_pck(args, save_path, aggre_net, files, kps, category, used_points, thresholds, bank, models, local=False):
    """local: this rank evaluates ALL pairs of the category by itself (sweep.py: categories are owned by ranks) - no pair sharding, no collective."""
    adapt_flip = bool(getattr(args, "ADAPT_FLIP", False))
    if adapt_flip and len(models) != 1:
        raise NotImplementedError("ADAPT_FLIP is defined by pck_train.py only (pck_train_two.py has no flip branch that reads two encoders)")
    if getattr(args, "TOTAL_SAVE_RESULT", 0):
        raise NotImplementedError("TOTAL_SAVE_RESULT > 0 asks for the matplotlib match visualisations (utils_visualization.py), which are not built")
    geo = bool(getattr(args, "COMPUTE_GEOAWARE_METRICS", False))
    P = args.NUM_PATCHES
    N = len(files) // 2
    dev = torch.device(device)
    split = 0
    if bank is None:
        bank_t, slot = build_feature_bank(args, aggre_net, files, P, dev, models)
        if len(models) == 2:      # channels of the first encoder: read off one of its maps
            split = int(aggre_net(torch.load(_feature_path(files[0], False, args.ENSEMBLE, models[0]), map_location="cpu")).reshape(-1, P * P).shape[0])
    else:
        bank_t, slot = bank[0], bank[1]
        split = int(bank[2]) if len(bank) > 2 else 0
    layout = bank[3] if bank is not None and len(bank) > 3 else "cp"
    if layout == "cp" and bank_t.shape[1] % 4 == 0 and split % 4 == 0:
        # position-major [n, P^2, C]: a keypoint's descriptor becomes one contiguous row (see csrc/cscore.hip); one transpose of
        # the category's bank replaces a strided line gather per (pair, keypoint, channel)
        bank_t, layout = bank_t.transpose(1, 2).contiguous(), "pc"
    kps = kps.float()
    K = kps.shape[1]
    if K > 32:
        raise ValueError("at most 32 keypoints per pair are supported")
    d = None if local else _dist()
    rank, world = (d.get_rank(), d.get_world_size()) if d else (0, 1)
    lo, hi = (N * rank) // world, (N * (rank + 1)) // world                   # contiguous pair block of this rank
    k1, k2 = kps[0::2], kps[1::2]
    idx = kpts_to_patch_idx_batch(args, k1, P).astype(np.int32) if N else np.zeros((0, K), np.int32)     # = kpts_to_patch_idx per pair, one numpy op
    nkp = torch.full((N,), K, dtype=torch.int32)
    sl = slice(lo, hi)
    _check_patch_idx(idx[sl], P)
    src, trg = torch.from_numpy(slot[0::2][sl].copy()), torch.from_numpy(slot[1::2][sl].copy())
    # the packing of the key-point rows is host work on host arrays (memoised per pair list): built here, once per category, from the numpy
    # index tables - transfer() then neither downloads nor hashes anything
    packed = cscore_ops.packed_rows_on(bank_t.device, slot[0::2][sl], slot[1::2][sl], idx[sl], nkp[sl].numpy()) if layout == "pc" and hi > lo else None
    xy = cscore_ops.transfer(bank_t, src, trg, torch.from_numpy(idx[sl]), nkp[sl], P, window=args.SOFT_EVAL_WINDOW,
                             soft_eval=bool(args.SOFT_EVAL), anno_size=args.ANNO_SIZE, split=split, layout=layout, packed=packed)
    if adapt_flip:
        xy = _adapt_flip(args, aggre_net, files, kps, category, used_points, bank, bank_t, layout, slot, sl, src, trg, xy, P, dev, models)
    alphas = shown_alphas = (0.1, 0.05, 0.01) if args.EVAL_DATASET != 'pascal' else (0.1, 0.05, 0.15)
    if thresholds is not None:
        thr = torch.tensor(thresholds, dtype=torch.float64)
    else:   # alpha * ANNO_SIZE is a float32 product in the reference (pck_train.py:160,222): hand the kernel that rounded
        thr = torch.ones(N, dtype=torch.float64)                              # product as its alpha, with a unit threshold
        alphas = tuple(float(np.float32(a) * np.float32(args.ANNO_SIZE)) for a in alphas)
    counts = cscore_ops.pck_counts(xy, k1[sl], k2[sl], thr[sl], nkp[sl], alphas)
    cnt = torch.zeros(N, 8 if geo else 4, dtype=torch.int32, device=counts.device)
    cnt[sl, :4] = counts
    if geo:
        # geometry-aware subset (pck_train.py:68-80,169-192,231-243): the same count launch with every other key point's
        # visibility cleared in the source annotation, so hits and totals are over the geo-aware points only
        table = AP10K_GEO_AWARE if args.EVAL_DATASET == 'ap10k' else SPAIR_GEO_AWARE.get(category)
        if table is None:
            raise ValueError(f"no geometry-aware key point groups for category {category!r}")
        groups = filtered_groups(table, (torch.arange(K) if used_points is None else used_points).tolist())
        vis_all, vis2_all = (k1[:, :, 2] * k2[:, :, 2] > 0).numpy(), (k2[:, :, 2] > 0).numpy()
        geo_mask = torch.zeros(N, K)
        for i in range(N):
            geo_mask[i, geo_aware_points(groups, vis_all[i], vis2_all[i])] = 1
        k1g = k1.clone()
        k1g[:, :, 2] *= geo_mask
        cnt[sl, 4:] = cscore_ops.pck_counts(xy, k1g[sl], k2[sl], thr[sl], nkp[sl], alphas)
    pred = torch.zeros(N, K, 2, dtype=torch.float32, device=xy.device)
    pred[sl] = xy
    if d:
        d.all_reduce(cnt)
        d.all_reduce(pred)
    cnt = cnt.cpu().numpy()
    pred = pred.cpu()
    used = torch.arange(K) if used_points is None else used_points.cpu()
    # renumber_used_points (utils_geoware.py) for every pair at once: zeros [N, 30, 2] with the used columns filled, then one row view per pair
    if _renumber_used_points is renumber_used_points:
        full = torch.zeros(N, 30, pred.shape[2])
        full[:, used] = pred
        full = full.numpy()
        preds30 = [full[i] for i in range(N)]
    else:                                                                       # a patched-in renumbering (tests): keep the per-pair call
        preds30 = [_renumber_used_points(pred[i], used).numpy() for i in range(N)]
    out_results = [{"src_fn": files[2 * i], "trg_fn": files[2 * i + 1], "src_kpts_pred": preds30[i], "resize_resolution": args.ANNO_SIZE}
                   for i in range(N)]
    # per-image PCK: float32 mean of 0/1 hits per pair (pck_train.py:158), then float32 mean over pairs (:205-206)
    with np.errstate(invalid="ignore", divide="ignore"):
        per_img = cnt[:, :3].astype(np.float32) / cnt[:, 3:4].astype(np.float32)
    if not args.KPT_RESULT:
        img_correct = torch.from_numpy(per_img.T.copy()).mean(dim=-1).tolist()
        img_correct.append(N)
    else:
        img_correct = None
    n_kpts = int(cnt[:, 3].sum())
    correct = (torch.from_numpy(cnt[:, :3].sum(0).astype(np.int64)) / n_kpts).tolist()
    correct.append(n_kpts)
    shown = correct[:3] if args.KPT_RESULT else img_correct[:3]
    if rank == 0:
        logger.info(f'{category}...' + ' | '.join(f'PCK-Transfer@{a:.2f}: {v * 100:.2f}%' for a, v in zip(_f32(shown_alphas), shown)))
    geo_score = []
    if geo:
        n_geo = int(cnt[:, 7].sum())
        geo_pairs = int((cnt[:, 7] > 0).sum())
        correct_geo = (torch.from_numpy(cnt[:, 4:7].sum(0).astype(np.int64)) / n_geo).tolist()
        geo_score = [geo_pairs / N, n_geo / n_kpts, *correct_geo, n_geo]
        if rank == 0:
            logger.info(' | '.join(f'PCK-Transfer_geo-aware@{a:.2f}: {v * 100:.2f}%' for a, v in zip(_f32(shown_alphas), correct_geo[:3])))
            logger.info(f'Geo-aware occurance count: {geo_pairs}, with ratio {geo_pairs / N * 100:.2f}%; total count ratio {n_geo / n_kpts * 100:.2f}%')
    return correct, geo_score, out_results, img_correct


def _adapt_flip(args, aggre_net, files, kps, category, used_points, bank, bank_t, layout, slot, sl, src, trg, xy, P, dev, models):
    """ADAPT_FLIP of compute_pck (pck_train.py:82-94,111-126) for this rank's pair block: a second keypoint transfer from the MIRRORED
    source image's features (`<img>_<MODEL>_flip.pt`) with the mirrored, left/right-permuted key points, the mutual-nearest-neighbour
    distance of both source variants to the target (get_distance_mutual_nn) - or, without MUTUAL_NN, the mask-based get_distance - and per
    pair the reference's choice between the two."""
    from .utils.utils_geoware import AP10K_FLIP, SPAIR_FLIP, flip_keypoints, flip_permutation, optimized_kps_1_to_2, permute_indices
    if layout != "pc":
        raise NotImplementedError("ADAPT_FLIP needs channel counts that are multiples of 4 (position-major bank)")
    table = AP10K_FLIP if args.EVAL_DATASET == 'ap10k' else SPAIR_FLIP[category]
    K = kps.shape[1]
    used = torch.arange(K) if used_points is None else used_points
    permute_list = flip_permutation(table, used.tolist(), K)
    lo, hi = sl.start, sl.stop
    n_img = bank_t.shape[0]
    if bank is not None and len(bank) > 4:
        flip_bank = bank[4]                                                    # caller-provided mirrored maps, same slots as the bank
        src_f = src + n_img
    else:
        # only SOURCE images are ever mirrored (get_patch_descriptors(flip=True) flips img1, pck_train.py:82-94): read the `_flip.pt`
        # files of this rank's distinct sources - a dataset with flip features for the sources only evaluates like in the reference
        flip_bank, fslot = build_feature_bank(args, aggre_net, files[0::2][lo:hi], P, dev, models, flip=True)
        flip_bank = flip_bank.transpose(1, 2).contiguous()
        src_f = torch.from_numpy(fslot.astype(np.int64)).to(src.dtype) + n_img
    both = torch.cat([bank_t, flip_bank.to(bank_t.device)], 0)                 # mirrored maps behind the bank
    k1, k2 = kps[0::2], kps[1::2]
    idx_f = np.zeros((hi - lo, K), np.int32)
    for n, i in enumerate(range(lo, hi)):
        vis = k1[i][:, 2] * k2[i][:, 2] > 0
        flipped = flip_keypoints(k1[i], args.ANNO_SIZE, permute_indices(permute_list, vis))
        if flipped.shape[0] != K:            # the flip table does not cover every key-point column (the reference's indexing breaks too)
            raise ValueError(f"flip table of {category!r} covers {flipped.shape[0]} of the {K} key points")
        idx_f[n] = kpts_to_patch_idx(args, flipped, P)
    _check_patch_idx(idx_f, P)
    nkp = torch.full((hi - lo,), K, dtype=torch.int32)
    xy_f = cscore_ops.transfer(both, src_f, trg, torch.from_numpy(idx_f), nkp, P, window=args.SOFT_EVAL_WINDOW,
                               soft_eval=bool(args.SOFT_EVAL), anno_size=args.ANNO_SIZE, layout="pc")
    if getattr(args, "MUTUAL_NN", False):
        d_orig = cscore_ops.mutual_nn_distance(both, src, trg, P).cpu()
        d_flip = cscore_ops.mutual_nn_distance(both, src_f, trg, P).cpu()
    else:
        # pck_train.py:122-124: the mask-based get_distance (utils_correspondence.py:22-52) on the normalised descriptors of the pair and the
        # `_mask.png` / `_mask_flip.png` files next to the features (round 6; the reference hard-codes 60 x 60 maps, this takes any square grid).
        # Masks of a caller-provided bank: bank[5] = (masks, flipped source masks), lists indexed like the bank's / the flip bank's images.
        if bank is not None and len(bank) > 5:
            masks, fmasks = bank[5]
            mask_of = lambda i, which: (masks[int(slot[2 * i + which])] if masks is not None else None)
            fmask_of = lambda n: (fmasks[int(src_f[n]) - n_img] if fmasks is not None else None)
        else:
            mask_of = lambda i, which: _load_mask(files[2 * i + which], False)
            fmask_of = lambda n: _load_mask(files[2 * (lo + n)], True)
        norm = lambda m: m / (torch.linalg.norm(m, dim=-1, keepdim=True) + 1e-10)              # normalize_feats (pck_train.py:24-29)
        d_orig, d_flip = torch.empty(hi - lo), torch.empty(hi - lo)
        for n, i in enumerate(range(lo, hi)):
            t_desc, m2 = norm(both[int(trg[n])]), mask_of(i, 1)
            d_orig[n] = cscore_ops.masked_nn_distance(norm(both[int(src[n])]), t_desc, mask_of(i, 0), m2).item()
            d_flip[n] = cscore_ops.masked_nn_distance(norm(both[int(src_f[n])]), t_desc, fmask_of(n), m2).item()
    out = xy.clone()
    xy_c, xyf_c = xy.cpu(), xy_f.cpu()
    for n, i in enumerate(range(lo, hi)):
        vis = k1[i][:, 2] * k2[i][:, 2] > 0
        out[n] = optimized_kps_1_to_2(args, xy_c[n], xyf_c[n], k1[i], k2[i], d_flip[n], d_orig[n], vis, permute_list).to(out.device)
    return out


def _check_patch_idx(idx, P):
    """A key point on the far border of the annotation frame (x or y = ANNO_SIZE, e.g. x = 0 mirrored) maps to patch column / row P: the
    reference's tensor indexing raises IndexError there (utils_correspondence.py:360); the kernel indexes unclamped, so refuse first."""
    if idx.size and (idx.min() < 0 or idx.max() >= P * P):
        bad = int(idx.max() if idx.max() >= P * P else idx.min())
        raise IndexError(f"index {bad} is out of bounds for dimension 2 with size {P * P}")


def _f32(alphas):
    return torch.tensor(alphas).tolist()          # the reference prints alpha.tolist() of a float32 tensor


def eval(args, aggre_net, save_path, split='val', _compute=None):
    compute_pck_fn = _compute or compute_pck
    aggre_net.eval()
    data_dir, categories, split = get_dataset_info(args, split)
    total_out_results, pcks, pcks_05, pcks_01, weights, kpt_weights = ([] for _ in range(6))
    geo = bool(getattr(args, "COMPUTE_GEOAWARE_METRICS", False))
    geo_lists = [[] for _ in range(6)]
    for cat in categories:
        files, kps, thresholds, used_points = load_eval_data(args, data_dir, cat, split)
        compute_args = (save_path, aggre_net, files, kps, cat, used_points)
        pck, correct_geo, out_results, img_correct = (compute_pck_fn(args, *compute_args, thresholds=thresholds) if args.BBOX_THRE
                                                      else compute_pck_fn(args, *compute_args))
        total_out_results.extend(out_results)
        update_stats(args, pcks, pcks_05, pcks_01, weights, kpt_weights, pck, img_correct)
        if geo:
            update_geo_stats(*geo_lists, correct_geo)
    pck_010, pck_005, pck_001 = log_weighted_pcks(args, logger, pcks, pcks_05, pcks_01, weights)
    if geo:
        log_geo_stats(args, *geo_lists, kpt_weights, total_out_results)
    aggre_net.train()
    return pck_010, pck_005, pck_001, total_out_results


def main(args, _eval=None):
    torch.manual_seed(args.SEED)
    np.random.seed(args.SEED)
    args.BBOX_THRE = not (args.IMG_THRESHOLD or args.EVAL_DATASET == 'pascal')
    if args.SAMPLE == 0:
        args.SAMPLE = None
    save_path = f'./results_{args.EVAL_DATASET}/pck_train_{args.NOTE}_sample_{args.EPOCH}_{args.SAMPLE}_lr_{args.LR}'
    os.makedirs(save_path, exist_ok=True)
    from .. import dist_env
    owned = dist_env.init_from_env()                                  # under torchrun: pairs of every category sharded over ranks
    d = _dist()
    lead = d is None or d.get_rank() == 0
    try:
        if lead:
            get_logger(save_path + '/result.log')                     # one log file, one result.pkl: rank 0's
        else:
            import logging
            get_logger().setLevel(logging.WARNING)
        logger.info(args)
        if args.DUMMY_NET:
            aggre_net = DummyAggregationNetwork()
        else:   # pck_train.py:347,356-363: GeoAware-SC's supervised post-processor over [SD s5, s4, s3, DINOv2] channel groups
            from .model_utils.projection_network import AggregationNetwork
            aggre_net = AggregationNetwork(feature_dims=[640, 1280, 1280, 768], projection_dim=args.PROJ_DIM, device=device)
            if args.LOAD is not None:
                aggre_net.load_pretrained_weights(torch.load(args.LOAD, map_location="cpu"))
                logger.info(f'Load model from {args.LOAD}')
        if not args.DO_EVAL:
            raise NotImplementedError("training is out of scope of the scoring path; run with DO_EVAL (configs/eval_zero_shot_spair.yaml)")
        with torch.no_grad():
            pck_010, pck_005, pck_001, result = (_eval or eval)(args, aggre_net, save_path, split='test')
        if lead:
            with open(save_path + '/result.pkl', 'wb') as f:
                pickle.dump(result, f)
    finally:
        dist_env.finalize(owned)
    return pck_010, pck_005, pck_001


def build_parser(two=False):
    p = argparse.ArgumentParser()
    p.add_argument('--config', type=str, default=None)
    p.add_argument('--SEED', type=int, default=42)
    p.add_argument('--NOTE', type=str, default='')
    p.add_argument('--SAMPLE', type=int, default=0)
    p.add_argument('--TEST_SAMPLE', type=int, default=20)
    p.add_argument('--TOTAL_SAVE_RESULT', type=int, default=0)
    p.add_argument('--IMG_THRESHOLD', action='store_true', default=False)
    p.add_argument('--ANNO_SIZE', type=int, default=840)
    p.add_argument('--LR', type=float, default=1.25e-3)
    p.add_argument('--EPOCH', type=int, default=1)
    p.add_argument('--TRAIN_DATASET', type=str, default='spair')
    p.add_argument('--ENSEMBLE', type=int, default=1)
    p.add_argument('--DO_EVAL', action='store_true', default=False)
    p.add_argument('--DUMMY_NET', action='store_true', default=False)
    p.add_argument('--EVAL_DATASET', type=str, default='spair')
    p.add_argument('--COMPUTE_GEOAWARE_METRICS', action='store_true', default=False)
    p.add_argument('--KPT_RESULT', action='store_true', default=False)
    p.add_argument('--ADAPT_FLIP', action='store_true', default=False)
    p.add_argument('--MUTUAL_NN', action='store_true', default=False)
    p.add_argument('--SOFT_EVAL', action='store_true', default=False)
    p.add_argument('--SOFT_EVAL_WINDOW', type=int, default=7)
    if two:                                                      # pck_train_two.py:444-445
        p.add_argument('--MODEL1', type=str, default='clip')
        p.add_argument('--MODEL2', type=str, default='dino')
    else:
        p.add_argument('--MODEL', type=str, default='clip')
    p.add_argument('--NUM_PATCHES', type=int, default=7)
    p.add_argument('--AP10K_EVAL_SUBSET', type=str, default='intra-species')   # intra-species | cross-species | cross-family
    # flags of the training half (pck_train.py:404-422): accepted so that the reference's command lines and yaml files parse;
    # training itself is not built (main() raises without DO_EVAL)
    for name, typ, default in (('WD', float, 1e-3), ('BZ', int, 1), ('SCHEDULER', str, None), ('SCHEDULER_P1', float, 0.3),
                               ('EVAL_EPOCH', int, 5000), ('LOAD', str, None), ('DENSE_OBJ', int, 1), ('GAUSSIAN_AUGMENT', float, 0.1),
                               ('FEAT_MAP_DROPOUT', float, 0.2), ('PROJ_DIM', int, 768), ('SELF_CONTRAST_WEIGHT', float, 0),
                               ('SOFT_TRAIN_WINDOW', int, 0)):
        p.add_argument(f'--{name}', type=typ, default=default)
    for name in ('NOT_WANDB', 'PAIR_AUGMENT'):
        p.add_argument(f'--{name}', action='store_true', default=False)
    p.add_argument('--DATA_DIR', type=str, default=None)       # extension: dataset root (default ./data/<set>, as the reference)
    return p


def parse_args(argv=None, two=False):
    args = build_parser(two).parse_args(argv)
    if args.config is not None:
        d = vars(args)
        d.update(load_config(args.config))
        args = argparse.Namespace(**d)
    return args


if __name__ == '__main__':
    main(parse_args())
