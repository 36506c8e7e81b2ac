"""Geometry-aware keypoint groups of the C score — C_score/utils/utils_geoware.py:6-54 (the SPair-71k / AP-10k tables of
keypoints that have a same-semantics twin elsewhere on the object: left/right wing tip, the four paws ...), :137-149
(renumber_indices) and :263-267 (renumber_used_points); for ADAPT_FLIP evaluation (pck_train.py:82-94,111-126) the left/right flip
tables :55-76,102-113, permute_indices :151-189, flip_keypoints :199-204 and optimized_kps_1_to_2 :269-279.  The PIL rotation helpers
belong to training, which this repo does not build.

The tables are dataset facts; they are written as "a+b" group strings and checked against the reference's own lists
by tests/test_host_cscore.py (fixture tests/golden/spair_host.npz `geo.table.*`)."""
import torch


def _groups(spec):
    """'0 1 4+5 6+7' -> [0, 1, [4, 5], [6, 7]]"""
    out = []
    for tok in spec.split():
        ids = [int(t) for t in tok.split("+")]
        out.append(ids if len(ids) > 1 else ids[0])
    return out


SPAIR_GEO_AWARE = {cat: _groups(spec) for cat, spec in {
    "aeroplane": "0 1 2 3 4+5 6+7 8+9 10+11 12+13 14+15 16+17 18+19 20+21 22 23 24",
    "bicycle": "0 1 2+3 4 5 6+7 8 9+10 13",
    "bird": "0 1+2 3 4 5 6 7 8 9 10+11 12+13 14+15 16",
    "boat": "0 1+2 3+4 5+6 7+8 9+10 11+12 13",
    "bottle": "0+1 2+3 4+5 6+7 8+9",
    "bus": "0+1 2+3+5+6 4+7 10+13+20+23 11+14+21+24 12+15+22+25 16+17+26+27 18+19+28+29",
    "car": "0+1 2+3+6+7 4+8 5+9 10+13+20+23 11+14+21+24 12+15+22+25 16+17+26+27 18+19+28+29",
    "cat": "0 1 2+3 4 5 6 7 8 9+10+11+12 13 14",
    "chair": "0+1 2+3 4+5+6+7 8+9 10+11 12+13",
    "cow": "0 1 2+3 4 5 6 7 8 9+10+11+12 13 14 15+16+17+18 19+20",
    "dog": "0 1 2+3 4 5 6 7 8 9+10+11+12 13 14 15",
    "horse": "0 1 2+3 4 5 6 7 8 9 10+11+12+13 14 15 16+17+18+19",
    "motorbike": "0+1 2+3 4 5 6 7 8 9 10 11 12",
    "person": "0 1 2 3 4 5 6 7 8+9 10+11 12+13 14+15 16+17 18+19",
    "pottedplant": "0+1+2+3 4+5 6+8 7",
    "sheep": "0 1 2+3 4 5 6 7 8 9+10+11+12 13 14 15+16+17+18 19+20",
    "train": "0+1 2+3 4+5 6+7 8+9 10+11 12+13 14+15 16+17",
    "tvmonitor": "0+2+4+6 1+3+5+7 8+10+12+14 9+11+13+15",
}.items()}

AP10K_GEO_AWARE = _groups("0 1 2 3 4 5+8 6+9+12+15 7+10+13+16 11+14")


# which key points trade places when the image is mirrored (a group = the left / right twins); singletons stay
SPAIR_FLIP = {cat: _groups(spec) for cat, spec in {
    "aeroplane": "0 1 2 3 4+5 6+7 8+9 10+11 12+13 14+15 16+17 18+19 20+21 22 23 24",
    "bicycle": "0 1 2+3 4 5 6+7 8 9+10 11",
    "bird": "0 1+2 3 4+5 6 7+8 9 10+11 12+13 14+15 16",
    "boat": "0 1+2 3+4 5+6 7+8 9+10 11+12 13",
    "bottle": "0+1 2+3 4+5 6+7 8+9",
    "bus": "0+1 2+3 5+6 4 7 8+18 11+21 9+19 12+22 10+20 13+23 14+15 24+25 16+17 26+27",
    "car": "0+1 2+3 4 5 6+7 8 9 10+20 13+23 11+21 14+24 12+22 15+25 16+17 26+27 18+19 28+29",
    "cat": "0+1 2+3 4+5 6+7 8 9+10 11+12 13 14",
    "chair": "0+1 2+3 4+5 6+7 8+9 10+11 12+13",
    "cow": "0+1 2+3 4+5 6+7 8 9+10 11+12 13 14 15+16 17+18 19+20",
    "dog": "0+1 2+3 4+5 6 7 8 9+10 11+12 13 14 15",
    "horse": "0+1 2+3 4+5 6+7 8 9 10+11 12+13 14 15 16+17 18+19",
    "motorbike": "0+1 2+3 4 5 6 7 8 9 10 11 12",
    "person": "0+1 2+3 4 5 6 7 8+9 10+11 12+13 14+15 16+17 18+19",
    "pottedplant": "0+2 1 3 4+5 6+8 7",
    "sheep": "0+1 2+3 4+5 6+7 8 9+10 11+12 13 14 15+16 17+18 19+20",
    "train": "0+1 2+3 4+5 6+7 8+9 10+11 12+13 14+15 16+17",
    "tvmonitor": "0+2 4+6 1 5 3+7 8+10 12+14 9 13 11+15",
}.items()}
AP10K_FLIP = _groups("0+1 2 3 4 5+8 6+9 12+15 7+10 13+16 11+14")


def flip_permutation(table, used_points, n_kps):
    """The permute list compute_pck builds (pck_train.py:82-94): the table as it is when it covers exactly the category's key-point
    columns, otherwise restricted to the used key points and renumbered to their column index."""
    if sum(len(i) if isinstance(i, list) else 1 for i in table) == n_kps:
        return table
    return filtered_groups(table, used_points)


def permute_indices(flip_list, vis=None):
    """Index map of a mirror flip: every group rotates by one place (twins swap), provided - when `vis` is given - all of the group's
    key points are visible; everything else stays (utils_geoware.py:151-189)."""
    flat = [i for item in flip_list for i in (item if isinstance(item, list) else [item])]
    indices = list(range(max(flat) + 1))
    for item in flip_list:
        if isinstance(item, list) and (vis is None or all(bool(vis[i]) for i in item)):
            for pos, i in enumerate(item):
                indices[i] = item[(pos + 1) % len(item)]
    return indices


def flip_keypoints(keypoints, img_size, permute_list=None):
    """Key points of the mirrored image: x -> img_size - x, rows re-ordered by the flip permutation (utils_geoware.py:199-204)."""
    out = keypoints.detach().clone()
    out[:, 0] = img_size - out[:, 0]
    return out if permute_list is None else out[permute_list]


def optimized_kps_1_to_2(args, kps_1_to_2, kps_1_to_2_flip, img1_kps, img2_kps, flip_dist, original_dist, vis, permute_list):
    """Adaptive flip (utils_geoware.py:269-279): when the mirrored source is the closer one, the predictions of the key points that stay
    mutually visible after the flip are taken from the mirrored pass."""
    masked = img1_kps * vis.unsqueeze(-1).float()
    flipped = flip_keypoints(masked, args.ANNO_SIZE, permute_indices(permute_list, None))
    vis_flip = flipped[:, 2] * img2_kps[:, 2] * img1_kps[:, 2] > 0
    if flip_dist < original_dist:
        kps_1_to_2 = kps_1_to_2.clone()
        kps_1_to_2[vis_flip] = kps_1_to_2_flip[vis_flip]
    return kps_1_to_2


def renumber_indices(lst, counter=[0]):
    """Nested list -> same nesting with 0,1,2,... in traversal order (positions among the category's used key points)."""
    out = []
    for item in lst:
        if isinstance(item, list):
            out.append(renumber_indices(item, counter))
        else:
            out.append(counter[0])
            counter[0] += 1
    return out


def renumber_used_points(kpts, idx):
    out = torch.zeros(30, kpts.shape[1])
    out[idx] = kpts
    return out


def filtered_groups(groups, used_points):
    """The table restricted to the key points a category uses, renumbered to their column index (pck_train.py:68-80)."""
    used = set(int(u) for u in used_points)
    kept = []
    for item in groups:
        item = [item] if isinstance(item, int) else item
        members = [i for i in item if i in used]
        if members:
            kept.append(members)
    return renumber_indices(kept, counter=[0])


def geo_aware_points(groups, vis, vis2):
    """Key points counted as geometry-aware for one pair (pck_train.py:169-180 / eval_spair.py): mutually visible members of a
    group of which the TARGET image shows at least two."""
    picked = []
    for item in groups:
        item = [item] if isinstance(item, int) else item
        if sum(bool(vis2[i]) for i in item) >= 2:
            picked += [i for i in item if bool(vis[i])]
    return picked
