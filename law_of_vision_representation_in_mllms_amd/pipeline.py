"""Disk-free scoring pipelines: towers -> features -> scores without the per-image `torch.save` files.

The reference writes every image's projected features to `<bench>/<enc>/tensor_{k}.pt` (llava_arch.py:229-248) and every dense map
to `<img>_<model>.pt` (extract_feature.py:125-129), then re-reads them - per pair, for the C score (pck_train.py:31-39).  Here the
features stay in HBM (288 GB holds every feature set of the paper's sweeps at once) and the score kernels consume them in place:

    a_scores_from_features   {encoder: [n, N_e, D]} device tensors -> {encoder: A score}; same arithmetic and the same python-float
                             reduction order as A_score.compute (compute.py:41-85), so the numbers are identical to the file route
    a_scores_from_stacks     PIL images -> each VisionEncoderStack's processor + tower + mm_projector -> the above
    c_score_from_tower       SPair-71k images -> dense feature bank per category straight from a tower or featurizer ->
                             C_score.pck_train.compute_pck(bank=...) (pck_train.py:57-245, 315-340)
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import numpy as np
import torch

from .A_score import compute as AC
from .C_score import pck_train as PT
from .C_score.utils.logger import log_weighted_pcks, update_stats
from .C_score.utils.utils_dataset import get_dataset_info, load_eval_data


def a_scores_from_features(features: Dict[str, torch.Tensor], refs=("clip336", "clip224"), verbose=True) -> Dict[str, float]:
    """features[name]: [n, N_e, D] on the GPU (one row block per image, all encoders over the SAME n images)."""
    if any(r not in features for r in refs):
        raise ValueError("Failed to load tensors from 'clip336' or 'clip224' subfolder")          # the reference's error (compute.py:34-35)
    results = {}
    scales = {}                                                     # row factors: once per feature set, not once per (encoder, reference)

    def scale_of(key, n):
        if key not in scales:
            scales[key] = AC._row_scales(features[key])
        sc = scales[key]
        return None if sc is None else sc[:n]
    for name, other in features.items():
        totals = []
        for r in refs:
            ref = features[r]
            n = min(other.shape[0], ref.shape[0])
            same = other.dtype == ref.dtype
            s = AC._score_batch(other[:n], ref[:n], scale_of(name, n) if same else None, scale_of(r, n) if same else None).double().cpu()
            totals.append(sum(float(s[i]) for i in range(n)) / n)
        results[name] = (totals[0] + totals[1]) / 2
    if verbose:
        for name, v in results.items():
            print(f'Average cosine similarity between clip224+clip336 and {name}: {v}')
    return results


@torch.no_grad()
def a_scores_from_stacks(stacks: Dict[str, object], images: Sequence, batch: int = 64, limit: int = 100, verbose=True) -> Dict[str, float]:
    """stacks[name]: a llava_arch.VisionEncoderStack (tower(s) + mm_projector); images: PIL images.  The first `limit` images
    are scored (the reference dumps tensor_1 .. tensor_100, llava_arch.py:241-247)."""
    images = list(images)[:limit]
    feats = {}
    for name, stack in stacks.items():
        tower = stack.get_vision_tower()
        proc = tower.image_processor
        chunks = []
        for s in range(0, len(images), batch):
            px = proc.preprocess(images[s:s + batch], return_tensors="pt")["pixel_values"]
            chunks.append(stack.encode_images(px))
        feats[name] = torch.cat(chunks, 0)
    return a_scores_from_features(feats, verbose=verbose)


def _to_map(f: torch.Tensor) -> torch.Tensor:
    """tower tokens [B, N, C] (kept as they are) or featurizer maps [B, C, h, w] -> position-major [B, P^2, C] fp32."""
    if f.dim() == 3:
        return f.float()
    return f.reshape(f.shape[0], f.shape[1], -1).permute(0, 2, 1).float()


@torch.no_grad()
def feature_bank(extract: Callable[[Sequence[str]], torch.Tensor], files: Sequence[str], batch: int = 32):
    """Distinct images of `files` -> ([n_img, P^2, C] fp32 device bank in the towers' own token layout, per-file-slot bank index)."""
    uniq, slot = {}, []
    for f in files:
        if f not in uniq:
            uniq[f] = len(uniq)
        slot.append(uniq[f])
    paths = list(uniq)
    maps = [_to_map(extract(paths[s:s + batch])) for s in range(0, len(paths), batch)]
    return torch.cat(maps, 0).contiguous(), np.asarray(slot, dtype=np.int32)


@torch.no_grad()
def c_score_from_tower(args, extract: Callable[[Sequence[str]], torch.Tensor], save_path: str = ".", split: str = "test",
                       extract2: Optional[Callable] = None, batch: int = 32):
    """pck_train.eval with the per-category feature bank built in HBM from `extract(image paths) -> features`.

    extract2: a second encoder for the two-encoder score (pck_train_two.py): the banks are concatenated on the channel axis."""
    aggre_net = PT.DummyAggregationNetwork()
    data_dir, categories, split = get_dataset_info(args, split)
    total_out_results, pcks, pcks_05, pcks_01, weights, kpt_weights = ([] for _ in range(6))
    for cat in categories:
        files, kps, thresholds, used_points = load_eval_data(args, data_dir, cat, split)
        bank, slot = feature_bank(extract, files, batch)
        split_c = 0
        if extract2 is not None:
            bank2, _ = feature_bank(extract2, files, batch)
            split_c = bank.shape[2]
            bank = torch.cat([bank, bank2], 2).contiguous()
        if bank.shape[1] != args.NUM_PATCHES ** 2:
            raise ValueError(f"feature maps have {bank.shape[1]} positions, NUM_PATCHES = {args.NUM_PATCHES}")
        layout = "pc"
        if bank.shape[2] % 4 or split_c % 4:                      # channel counts the position-major kernel cannot take
            bank, layout = bank.transpose(1, 2).contiguous(), "cp"
        pck, _, out_results, img_correct = PT._compute_pck(args, save_path, aggre_net, files, kps, cat, used_points,
                                                           thresholds if args.BBOX_THRE else None, (bank, slot, split_c, layout), models=("fused",))
        total_out_results.extend(out_results)
        update_stats(args, pcks, pcks_05, pcks_01, weights, kpt_weights, pck, img_correct)
    pck_010, pck_005, pck_001 = log_weighted_pcks(args, PT.logger, pcks, pcks_05, pcks_01, weights)
    return pck_010, pck_005, pck_001, total_out_results
