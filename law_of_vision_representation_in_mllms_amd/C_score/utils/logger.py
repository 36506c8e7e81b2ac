"""Config / stats / log lines of the C score — C_score/utils/logger.py:8-98 without the loguru dependency."""
import logging
import sys

import numpy as np
import yaml

logger = logging.getLogger("visrep.cscore")


def load_config(config_path):
    with open(config_path, 'r') as f:
        return yaml.safe_load(f)


def get_logger(output_file=None):
    if not logger.handlers:
        h = logging.StreamHandler(sys.stderr)
        h.setFormatter(logging.Formatter("[%(asctime)s] %(message)s", "%Y-%m-%d %H:%M:%S"))
        logger.addHandler(h)
        logger.setLevel(logging.INFO)
    if output_file:
        fh = logging.FileHandler(output_file)
        fh.setFormatter(logging.Formatter("[%(asctime)s] %(message)s", "%Y-%m-%d %H:%M:%S"))
        logger.addHandler(fh)
    return logger


def update_stats(args, pcks, pcks_05, pcks_01, weights, kpt_weights, pck, img_correct):
    src = pck if args.KPT_RESULT else img_correct
    pcks.append(src[0])
    pcks_05.append(src[1])
    pcks_01.append(src[2])
    weights.append(src[3])
    kpt_weights.append(pck[3])


def log_weighted_pcks(args, logger, pcks, pcks_05, pcks_01, weights):
    pck_010 = np.average(pcks, weights=weights)
    pck_005 = np.average(pcks_05, weights=weights)
    pck_001 = np.average(pcks_01, weights=weights)
    if not args.KPT_RESULT and args.TRAIN_DATASET == "spair":
        logger.info(f"Weighted Per image PCK0.10: {pck_010 * 100:.2f}%, image PCK0.05: {pck_005 * 100:.2f}%, image PCK0.01: {pck_001 * 100:.2f}%")
    else:
        logger.info(f"Weighted Per kpt PCK0.10: {pck_010 * 100:.2f}%, kpt PCK0.05: {pck_005 * 100:.2f}%, kpt PCK0.01: {pck_001 * 100:.2f}")
    return pck_010, pck_005, pck_001


def update_geo_stats(geo_aware, geo_aware_count, pcks_geo, pcks_geo_05, pcks_geo_01, weights_geo, correct_geo):
    """correct_geo = compute_pck's geo_score: [pairs with geo points / N, geo points / key points, PCK@3 alphas, n geo points]."""
    for lst, v in zip((geo_aware, geo_aware_count, pcks_geo, pcks_geo_05, pcks_geo_01, weights_geo), correct_geo):
        lst.append(v)


def log_geo_stats(args, geo_aware, geo_aware_count, pcks_geo, pcks_geo_05, pcks_geo_01, weights_geo, weights_kpt, total_out_results):
    from .eval_spair import convert_all_results, get_img_result
    avg_geo_aware = np.average(geo_aware) * 100
    avg_geo_aware_count = np.average(geo_aware_count, weights=weights_kpt) * 100
    logger.info(f"Average images geo-aware occurrence: {avg_geo_aware:.2f}%, Average points geo-aware occurrence: {avg_geo_aware_count:.2f}%")
    if not args.KPT_RESULT and args.TRAIN_DATASET == "spair":     # per-image numbers come from re-reading the annotations
        g10, g05, g01 = get_img_result(convert_all_results(total_out_results), geo=True)[0].tolist()
        logger.info(f"Weighted Per image geo-aware PCK0.10: {g10*100:.2f}%, image PCK0.05: {g05*100:.2f}%, image PCK0.01: {g01*100:.2f}%")
        return g10, g05, g01
    g10, g05, g01 = (np.average(p, weights=weights_geo) * 100 for p in (pcks_geo, pcks_geo_05, pcks_geo_01))
    logger.info(f"Weighted Per kpts geo-aware PCK0.10: {g10:.2f}%, kpts PCK0.05: {g05:.2f}%, kpts PCK0.01: {g01:.2f}%")
    return g10, g05, g01
