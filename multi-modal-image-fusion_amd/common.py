# -*- coding: utf-8 -*-
"""Training plumbing -- API mirror of the reference's common.py (argument parsers :23-71, setup_seed
:84-93, setup_dist :96-102, reduce_value :105-113, AverageMeter :116-133, WarmupLR :136-166,
Logger / make_logger :169-210, save_result :74-81): same names, positional orders, keywords, defaults and return
values, so call sites written against the reference (`WarmupLR(optimizer, 0.001, len(loader))`,
`log_dir, logger = make_logger(BASE_DIR)`) behave the same.  Only plumbing lives here; the hot path is in mmif/."""
import argparse
import logging
import os
import random
from datetime import datetime

import numpy as np
import torch
import torch.distributed as dist

from data.transform import denorm, norm

__all__ = ['get_train_args', 'get_test_args', 'save_result', 'setup_seed', 'setup_dist', 'reduce_value', 'AverageMeter',
           'WarmupLR', 'Logger', 'make_logger', 'norm', 'denorm']


# flag tables: (name, default, type, help).  The reference's flags (common.py:23-71) keep their names, defaults and -- including the
# `type=bool` ones, which treat any non-empty string as True -- their argparse behaviour; the last four are this engine's additions
# (the reference picks model / precision by editing the scripts).
_MODELS = 'PFNetv1 | PFNetv2 | DenseFuse | VIFNet | NestFuse | RFNNest | DeepFuse | DBNet | SEDRFuse | IFCNN | DIFNet | PMGI | UNFusion | MAFusion | Res2Fusion'
_EXTRA = [('model', 'PFNetv1', str, _MODELS),
          ('dtype', 'fp32', str, 'feature-map storage: fp32 (parity) | bf16 (throughput)'),
          ('synthetic', 0, int, '>0: train on this many synthetic random pairs (no dataset needed)'),
          ('graph', False, bool, 'replay forward + losses + backward of the training batch shape as one hipGraph (launch-bound small batches)')]
_TRAIN = [('lr', 1e-4, float, 'learning rate'), ('bs', 16, int, 'batch size'), ('epoch', 12, int, 'num of epochs'),
          ('use_patches', True, bool, 'enable to train with patches'), ('warmup', False, bool, 'enable to warm up lr'),
          ('clip_grad', True, bool, 'enable to clip grad norm'), ('local_rank', 0, int, 'node rank for distribution'),
          ('local_world_size', 1, int, 'num of gpus for distribution'), ('data', 'roadscene', str, 'dataset folder name')]
_TEST = [('use_gpu', True, bool, 'enable to test on gpu'), ('data', 'roadscene', str, 'dataset folder name'),
         ('ckpt', '2023-02-26_23-15', str, 'checkpoint folder name')]


def _parse(description, table):
    ap = argparse.ArgumentParser(description=description)
    for name, default, kind, text in table + _EXTRA:
        ap.add_argument('--' + name, default=default, type=kind, help=text)
    return ap.parse_args()


def get_train_args():
    return _parse('Training', _TRAIN)


def get_test_args():
    return _parse('Testing', _TEST)


def save_result(pred, img1=None, img2=None):
    """uint8 HWC image of the prediction (or of img1 | img2 | pred side by side); denorm truncates, as the reference's does"""
    if img1 is not None and img2 is not None:
        return np.concatenate(tuple(map(denorm, (img1, img2, pred))), axis=1)
    return denorm(pred)


def setup_seed(seed=0, benchmark=False, deterministic=True):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    # the HIP kernels are deterministic by construction (two-stage reductions, no float atomics)


def setup_dist(rank=0, world_size=1):
    """One process per GPU over RCCL ('nccl' on ROCm); rendezvous on 127.0.0.1; honours torchrun's env."""
    from mmif.dist import setup_dist as _setup
    if 'LOCAL_RANK' in os.environ:  # launched by torch.distributed.run
        return _setup()
    return _setup(rank, world_size)


def reduce_value(value, world_size=1, average=True):
    if world_size > 1:
        with torch.no_grad():
            dist.all_reduce(value)
            if average:
                value /= world_size
    return value


class AverageMeter(object):
    """running (sample-weighted) mean: .val last value, .avg mean, .sum, .count (reference common.py:116-133)"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def is_empty(self):
        return not self.count

    def update(self, val, n=1):
        self.val, self.sum, self.count = val, self.sum + val * n, self.count + n
        self.avg = self.sum / self.count


class WarmupLR(torch.optim.lr_scheduler._LRScheduler):
    """lr = base_lr * factor(iter): `warmup_factor` ('constant') or the line from warmup_factor to 1 ('linear') while
    iter < warmup_iters, then 1 (reference common.py:136-166, same positional order)"""

    def __init__(self, optimizer, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear", last_epoch=-1, verbose=False):
        self.warmup_factor = warmup_factor
        self.warmup_iters = warmup_iters
        self.warmup_method = warmup_method
        super(WarmupLR, self).__init__(optimizer, last_epoch)   # torch >= 2.7 dropped the `verbose` argument

    def get_lr(self):
        f = self._get_warmup_factor_at_iter(self.warmup_method, self.last_epoch, self.warmup_iters, self.warmup_factor)
        return [base_lr * f for base_lr in self.base_lrs]

    @staticmethod
    def _get_warmup_factor_at_iter(method, iter, warmup_iters, warmup_factor):
        if iter >= warmup_iters:
            return 1.0
        if method == 'constant':
            return warmup_factor
        elif method == 'linear':
            alpha = iter / warmup_iters
            return warmup_factor + (1.0 - warmup_factor) * alpha
        else:
            raise ValueError("only supported ['constant', 'linear'] method")


class Logger(object):
    def __init__(self, log_path):
        log_name = os.path.basename(log_path)
        log_dir = os.path.dirname(log_path)
        if not os.path.exists(log_dir):
            os.makedirs(log_dir)
        self.log_name = log_name if log_name else "train.log"
        self.log_path = log_path

    def init_logger(self):
        logger = logging.getLogger(self.log_name)
        logger.setLevel(logging.INFO)
        fh = logging.FileHandler(self.log_path, "w")
        fh.setLevel(logging.INFO)
        fh.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
        ch = logging.StreamHandler()
        ch.setLevel(logging.INFO)
        logger.addHandler(fh)
        logger.addHandler(ch)
        return logger


def make_logger(root_dir):
    """<root_dir>/../checkpoints/<YYYY-mm-dd_HH-MM>/train.log -> (log_dir, logger)   (reference common.py:200-210)"""
    time_str = datetime.strftime(datetime.now(), "%Y-%m-%d_%H-%M")
    log_dir = os.path.join(root_dir, '..', 'checkpoints', time_str)
    logger = Logger(os.path.join(log_dir, "train.log")).init_logger()
    return log_dir, logger
