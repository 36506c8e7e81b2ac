"""Keypoint-transfer entry points of the C score on MI355X — drop-ins for the two functions pck_train.py imports from
C_score/utils/utils_correspondence.py: kpts_to_patch_idx (:384-388) and calculate_keypoint_transformation (:345-382)."""
import numpy as np
import torch

from ... import cscore_ops


def kpts_to_patch_idx(args, img1_kps, num_patches):
    # identical host arithmetic (numpy float64 * float32, truncation to int32)
    img1_y, img1_x = img1_kps[:, 1].cpu().numpy(), img1_kps[:, 0].cpu().numpy()
    img1_y_patch = (num_patches / args.ANNO_SIZE * img1_y).astype(np.int32)
    img1_x_patch = (num_patches / args.ANNO_SIZE * img1_x).astype(np.int32)
    return num_patches * img1_y_patch + img1_x_patch


def kpts_to_patch_idx_batch(args, kps, num_patches):
    """kpts_to_patch_idx for a stack of key-point sets [N, K, 3] -> int32 [N, K] in ONE numpy expression: the same elementwise float32
    arithmetic as N calls (python-float factor x float32 array, truncation to int32), without 12,234 tiny tensor -> numpy round trips per
    category list (0.4 s of host time per setting of the sweep, tools/diag/c_leg_profile.py)."""
    a = kps.cpu().numpy()
    y_patch = (num_patches / args.ANNO_SIZE * a[:, :, 1]).astype(np.int32)
    x_patch = (num_patches / args.ANNO_SIZE * a[:, :, 0]).astype(np.int32)
    return num_patches * y_patch + x_patch


def calculate_keypoint_transformation(args, img1_desc, img2_desc, img1_patch_idx, num_patches):
    """img*_desc: [1, P^2, C] patch descriptors (normalised or not — the kernel applies normalize_feats itself, and
    re-normalising unit rows is the identity to fp32 rounding).  Returns Tensor[K, 2] (x, y) on the descriptors' device.
    More than 32 keypoints are processed in chunks of 32 (kernel limit)."""
    dev = img1_desc.device if img1_desc.is_cuda else torch.device("cuda")
    P = num_patches
    bank = torch.stack([img1_desc[0].t(), img2_desc[0].t()]).to(device=dev, dtype=torch.float32).contiguous()   # [2, C, P^2]
    idx = torch.as_tensor(np.asarray(img1_patch_idx), dtype=torch.int32)
    outs = []
    for s in range(0, len(idx), 32):
        part = idx[s:s + 32]
        xy = cscore_ops.transfer(bank, torch.tensor([0]), torch.tensor([1]), part[None], torch.tensor([len(part)]), P,
                                 window=getattr(args, "SOFT_EVAL_WINDOW", 0), soft_eval=bool(getattr(args, "SOFT_EVAL", False)),
                                 anno_size=args.ANNO_SIZE)
        outs.append(xy[0, :len(part)])
    return torch.cat(outs, 0)


def convert_to_binary_mask(img_path, threshold=127, angle=None):
    """utils_correspondence.py:6-20: a mask image -> float32 {0, 1} tensor [H, W] (grey level > threshold)."""
    import numpy as np
    from PIL import Image
    img = Image.open(img_path).convert('L')
    if angle is not None:
        img = img.rotate(angle)
    return (torch.from_numpy(np.array(img)) > threshold).float()


def get_distance(feature1, feature2, mask1, mask2, RESOLUTION=64):
    """utils_correspondence.py:22-52: the mask-based flip distance (ADAPT_FLIP without MUTUAL_NN, pck_train.py:122-124).  feature* [1, P^2, C]
    normalised descriptors, mask* [H, W] binary masks -> 0-dim tensor.  The reference reshapes to 60 x 60 whatever it is given (so it only
    runs on 60 x 60 maps); this one takes any square grid and equals it at 60 (tests/golden/maskdist.npz).  cscore_ops.masked_nn_distance."""
    from ... import cscore_ops
    return cscore_ops.masked_nn_distance(feature1.cuda(), feature2.cuda(), mask1, mask2, RESOLUTION)


def get_distance_mutual_nn(feature1, feature2):
    """utils_correspondence.py:54-73: feature* [1, P^2, C] descriptors -> mean cdist over the mutual nearest neighbours (a 0-dim tensor).
    Runs visrep_gram_pairs_f32 + visrep_mutual_nn_distance; the descriptors are L2-normalised there (idempotent for the reference's
    already-normalised inputs)."""
    from ... import cscore_ops
    f1, f2 = feature1.reshape(-1, feature1.shape[-1]).float(), feature2.reshape(-1, feature2.shape[-1]).float()
    bank = torch.stack([f1, f2]).cuda()
    P = int(round(f1.shape[0] ** 0.5))
    return cscore_ops.mutual_nn_distance(bank, torch.tensor([0]), torch.tensor([1]), P)[0]
