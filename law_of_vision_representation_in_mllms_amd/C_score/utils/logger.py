"""Config / stats / log lines of the C score — C_score/utils/logger.py:8-72 without the loguru dependency."""
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
