"""Pair loading for the C score — host-side restatement of C_score/utils/utils_dataset.py:13-35 (preprocess_kps_pad),
:208-274 (load_spair_data), :151-204 (load_ap10k_data), :278-371 (load_pascal_data, evaluation splits), :115-123
(load_eval_data), :125-147 (get_dataset_info).  The PF-Pascal training split reads Matlab annotation files and belongs to
training, which is not built."""
import json
import os
from glob import glob

import numpy as np
import torch

PASCAL_CLASSES = ('aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog', 'horse',
                  'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor')


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
    return _finish(files, kps, thresholds)


def _finish(files, kps, thresholds):
    kps = torch.stack(kps)
    used_kps, = torch.where(kps[:, :, 2].any(dim=0))
    return files, kps[:, used_kps, :], thresholds, used_kps


def load_ap10k_data(path="data/ap-10k", size=840, category='cat', split='test', subsample=20):
    """AP-10k pairs of one species / family / 'all': the pair file names the two per-image annotation files (cwd-relative, as
    written by the dataset preparation script); 17 key points with COCO visibility 0/2 -> 0/1; bbox is (l, t, w, h)."""
    np.random.seed(42)
    pairs = sorted(glob(f'{path}/PairAnnotation/{split}/*:{category}.json'))
    if subsample is not None and subsample > 0:
        pairs = [pairs[ix] for ix in np.random.choice(len(pairs), subsample)]
    files, kps, thresholds = [], [], []
    for pair in pairs:
        with open(pair) as f:
            data = json.load(f)
        sides = []
        for key in ("src_json_path", "trg_json_path"):
            with open(data[key]) as f:
                anno = json.load(f)
            k = torch.tensor(anno["keypoints"]).view(-1, 3).float()
            k[:, -1] /= 2
            k, _, _, scale = preprocess_kps_pad(k, anno["width"], anno["height"], size)
            bbox = np.asarray(anno["bbox"])
            sides.append((k, max(bbox[3], bbox[2]) * scale))
            files.append(data[key].replace("json", "jpg").replace('ImageAnnotation', 'JPEGImages'))
        if 'test' in split:
            thresholds.append(sides[1][1])
        elif 'trn' in split:
            thresholds += [sides[0][1], sides[1][1]]
        kps += [sides[0][0], sides[1][0]]
    return _finish(files, kps, thresholds)


def load_pascal_data(path="data/PF-dataset-PASCAL", size=256, category='cat', split='test', subsample=None):
    """PF-Pascal pairs of one class from `<split>_pairs_pf_pascal.csv` (source, target, class id, then ';'-joined XA, YA, XB, YB);
    up to 20 key points per image; image sizes come from the JPEG headers; no bbox thresholds (PCK is relative to ANNO_SIZE)."""
    import pandas as pd
    from PIL import Image
    if split.startswith('train'):
        raise NotImplementedError("the PF-Pascal training split (Matlab annotations) belongs to training, which is not built")
    np.random.seed(42)
    table = pd.read_csv(f'{path}/{split}_pairs_pf_pascal.csv')
    rows = table.iloc[np.where(table.iloc[:, 2].values.astype("int") - 1 == PASCAL_CLASSES.index(category))[0], :]

    def points(xs, ys):
        X, Y = np.fromstring(xs, sep=";"), np.fromstring(ys, sep=";")
        pts = np.zeros((20, 3), np.float32)
        pts[:, :2] = -1
        pts[:len(X), 0], pts[:len(X), 1], pts[:len(X), 2] = X, Y, 1
        return torch.from_numpy(pts)
    files, kps = [], []
    for i in range(len(rows)):
        for name, cols in ((rows.iloc[i, 0], (3, 4)), (rows.iloc[i, 1], (5, 6))):
            fn = f'{path}/../{name}'
            w, h = Image.open(fn).size
            kps.append(preprocess_kps_pad(points(rows.iloc[i, cols[0]], rows.iloc[i, cols[1]]), w, h, size)[0])
            files.append(fn)
    files, kps, _, used_kps = _finish(files, kps, None)
    return files, kps, None, used_kps


def load_eval_data(args, path, category, split):
    loader = {'ap10k': load_ap10k_data, 'pascal': load_pascal_data}.get(args.EVAL_DATASET, load_spair_data)
    return loader(path, args.ANNO_SIZE, category, split, args.TEST_SAMPLE)


def get_dataset_info(args, split):
    """(data_dir, categories, split) — `args.DATA_DIR` overrides the reference's hard-wired ./data/<set> location."""
    override = getattr(args, "DATA_DIR", None)
    if args.EVAL_DATASET == 'pascal':
        data_dir = override or 'data/PF-dataset-PASCAL'
        categories = sorted(os.listdir(os.path.join(data_dir, 'Annotations')))
    elif args.EVAL_DATASET == 'ap10k':
        data_dir = override or 'data/ap-10k'
        anno = os.path.join(data_dir, 'ImageAnnotation')
        families = os.listdir(anno)
        subset = args.AP10K_EVAL_SUBSET
        categories = []
        if subset == 'intra-species':
            categories = [species for fam in families for species in os.listdir(os.path.join(anno, fam))]
        elif subset == 'cross-species':
            categories = [fam for fam in families if len(os.listdir(os.path.join(anno, fam))) > 1]
            split += '_cross_species'
        elif subset == 'cross-family':
            categories = ['all']
            split += '_cross_family'
        categories = sorted(categories)
    else:
        data_dir = override or 'data/SPair-71k'
        categories = sorted(os.listdir(os.path.join(data_dir, 'ImageAnnotation')))
    return data_dir, categories, split
