"""SPair-71k pair loading for the C score — host-side restatement of C_score/utils/utils_dataset.py:13-35 (preprocess_kps_pad),
:208-274 (load_spair_data), :115-123 (load_eval_data), :125-147 (get_dataset_info).  PF-Pascal / AP-10k loaders are later
rows."""
import json
import os
from glob import glob

import numpy as np
import torch


def preprocess_kps_pad(kps, img_width, img_height, size):
    """Key points of an image whose long side is resized to `size` and which is then centre-padded to a square
    (same arithmetic as the reference: np.around for the short side, int() truncation for the offset)."""
    out = kps.clone()
    long_side = max(img_width, img_height)
    scale = size / long_side
    out[:, :2] *= scale
    offs = [0, 0]                                          # (x, y)
    if img_width != img_height:
        axis = 1 if img_height < img_width else 0         # landscape pads y, portrait pads x
        short = img_height if axis == 1 else img_width
        offs[axis] = int((size - int(np.around(size * short / long_side))) / 2)
        out[:, axis] += offs[axis]
    out *= out[:, 2:3].clone()                             # invisible key points -> (0, 0, 0)
    return out, offs[0], offs[1], scale


def load_spair_data(path="data/SPair-71k", size=256, category='cat', split='test', subsample=None):
    np.random.seed(42)
    pairs = sorted(glob(f'{path}/PairAnnotation/{split}/*:{category}.json'))
    if subsample is not None and subsample > 0:
        pairs = [pairs[ix] for ix in np.random.choice(len(pairs), subsample)]
    files, thresholds, kps = [], [], []
    category_anno = list(glob(f'{path}/ImageAnnotation/{category}/*.json'))[0]
    with open(category_anno) as f:
        num_kps = len(json.load(f)['kps'])
    for pair in pairs:
        with open(pair) as f:
            data = json.load(f)
        assert category == data["category"]
        source_fn = f'{path}/JPEGImages/{category}/{data["src_imname"]}'
        target_fn = f'{path}/JPEGImages/{category}/{data["trg_imname"]}'
        source_bbox = np.asarray(data["src_bndbox"])    # (x1, y1, x2, y2)
        target_bbox = np.asarray(data["trg_bndbox"])
        out = []
        for fn, wh in ((source_fn, data["src_imsize"][:2]), (target_fn, data["trg_imsize"][:2])):
            with open(fn.replace('JPEGImages', 'ImageAnnotation').replace('jpg', 'json')) as f:
                kpts = json.load(f)['kps']
            k = torch.zeros(num_kps, 3)
            for i in range(30):
                point = kpts[str(i)]
                if point is None:
                    k[i, :3] = 0
                else:
                    k[i, :2] = torch.Tensor(point).float()
                    k[i, 2] = 1
            out.append(preprocess_kps_pad(k, wh[0], wh[1], size))
        (source_kps, _, _, src_scale), (target_kps, _, _, trg_scale) = out
        if split == 'test' or split == 'val':
            thresholds.append(max(target_bbox[3] - target_bbox[1], target_bbox[2] - target_bbox[0]) * trg_scale)
        elif split == 'trn':
            thresholds.append(max(source_bbox[3] - source_bbox[1], source_bbox[2] - source_bbox[0]) * src_scale)
            thresholds.append(max(target_bbox[3] - target_bbox[1], target_bbox[2] - target_bbox[0]) * trg_scale)
        kps += [source_kps, target_kps]
        files += [source_fn, target_fn]
    kps = torch.stack(kps)
    used_kps, = torch.where(kps[:, :, 2].any(dim=0))
    kps = kps[:, used_kps, :]
    return files, kps, thresholds, used_kps


def load_eval_data(args, path, category, split):
    if args.EVAL_DATASET in ('ap10k', 'pascal'):
        raise NotImplementedError("only the SPair-71k loader is built so far (SURVEY.md §8f N4)")
    return load_spair_data(path, args.ANNO_SIZE, category, split, args.TEST_SAMPLE)


def get_dataset_info(args, split):
    if args.EVAL_DATASET in ('ap10k', 'pascal'):
        raise NotImplementedError("only the SPair-71k loader is built so far (SURVEY.md §8f N4)")
    data_dir = getattr(args, "DATA_DIR", 'data/SPair-71k')
    categories = sorted(os.listdir(os.path.join(data_dir, 'ImageAnnotation')))
    return data_dir, categories, split
