"""Two-encoder C score on MI355X - drop-in for the evaluation half of C_score/pck_train_two.py.

The reference scores encoder COMBINATIONS (CLIP@224+DINOv2, CLIP@336+DINOv2, ... - 2 of the paper's 13 settings) by
loading both feature maps of every image (`<base>_{MODEL1}.pt`, `<base>_{MODEL2}.pt`, pck_train_two.py:37-53), L2-
normalising each encoder's channels separately, concatenating and normalising again (pck_train_two.py:24-36); everything
after that is pck_train.py.  Same here: the bank holds the two raw maps concatenated on the channel axis and
`visrep_cscore_transfer(split = C1)` applies both normalisations inside the Gram kernel, so no normalised copy of the
descriptors is ever written.  Same argparse flags / yaml keys (MODEL1, MODEL2) as the reference.
"""
import torch

from . import pck_train as _one
from .model_utils.projection_network import DummyAggregationNetwork  # noqa: F401 (API parity)
from .pck_train import _feature_path, logger  # noqa: F401

device = _one.device


def normalize_feats(args, feat1, feat2, epsilon=1e-10):
    # pck_train_two.py:24-36 (the reference only defines `feats` under DUMMY_NET; the zero-shot score always sets it)
    if not args.DUMMY_NET:
        raise NotImplementedError("two-encoder normalisation is only defined for DUMMY_NET (pck_train_two.py:25)")
    a = feat1 / (torch.linalg.norm(feat1, dim=-1)[:, :, None] + epsilon)
    b = feat2 / (torch.linalg.norm(feat2, dim=-1)[:, :, None] + epsilon)
    feats = torch.cat([a, b], dim=-1)
    return feats / (torch.linalg.norm(feats, dim=-1)[:, :, None] + epsilon)


def prepare_feature_paths_and_load(aggre_net, img_path, flip, ensemble, num_patches, device, model1, model2):
    descs = []
    for model in (model1, model2):
        d = torch.load(_feature_path(img_path, flip, ensemble, model), map_location="cpu").to(device)
        descs.append(aggre_net(d).reshape(1, 1, -1, num_patches ** 2).permute(0, 1, 3, 2))
    return descs[0], descs[1], None


def get_patch_descriptors(args, aggre_net, num_patches, files, pair_idx, flip=False, flip2=False, img1=None, img2=None,
                          device='cuda'):
    a1, a2, mask1 = prepare_feature_paths_and_load(aggre_net, files[pair_idx * 2], flip, args.ENSEMBLE, num_patches, device, args.MODEL1, args.MODEL2)
    b1, b2, mask2 = prepare_feature_paths_and_load(aggre_net, files[pair_idx * 2 + 1], flip2, args.ENSEMBLE, num_patches, device, args.MODEL1, args.MODEL2)
    return normalize_feats(args, a1[0], a2[0]), normalize_feats(args, b1[0], b2[0]), mask1, mask2


def compute_pck(args, save_path, aggre_net, files, kps, category=None, used_points=None, thresholds=None, bank=None):
    if not args.DUMMY_NET:
        raise NotImplementedError("two-encoder normalisation is only defined for DUMMY_NET (pck_train_two.py:25)")
    return _one._compute_pck(args, save_path, aggre_net, files, kps, category, used_points, thresholds, bank,
                             models=(args.MODEL1, args.MODEL2))


def eval(args, aggre_net, save_path, split='val'):
    return _one.eval(args, aggre_net, save_path, split, _compute=compute_pck)


def main(args):
    return _one.main(args, _eval=eval)


def parse_args(argv=None):
    return _one.parse_args(argv, two=True)


if __name__ == '__main__':
    main(parse_args())
