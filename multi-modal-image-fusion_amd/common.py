# -*- coding: utf-8 -*-
"""Training plumbing -- API mirror of the reference's common.py (argument parsers :23-71, setup_seed
:84-93, setup_dist :96-102, reduce_value :105-113, AverageMeter :116-133, WarmupLR :136-166,
make_logger :169-210, save_result :74-81).  Only plumbing lives here; the hot path is in mmif/."""
import argparse
import logging
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist

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
_TRAIN = [('lr', None, float, 'learning rate'), ('bs', None, int, 'batch size'), ('epoch', None, int, 'number of epoch'),
          ('use_patches', True, bool, 'use patches or random crop'), ('warmup', False, bool, 'use warmup lr'),
          ('clip_grad', True, bool, 'clip grad norm'), ('local_rank', 0, int, 'node rank for distributed training'),
          ('local_world_size', 1, int, 'number of gpus for distributed training'), ('data', 'polar', str, 'dataset folder name')]
_TEST = [('use_gpu', True, bool, 'use gpu or cpu'), ('data', 'polar', str, 'dataset folder name'), ('ckpt', None, str, 'checkpoint folder name')]


def _parse(description, table):
    ap = argparse.ArgumentParser(description=description)
    for name, default, kind, text in table + _EXTRA:
        ap.add_argument('--' + name, default=default, type=kind, help=text)
    return ap.parse_args()


def get_train_args():
    return _parse('Training', _TRAIN)


def get_test_args():
    return _parse('Testing', _TEST)


def norm(img, mode=None):
    """data/transform.py:15-29 -- mode None: /255"""
    img = np.asarray(img, dtype=np.float32)
    if mode is None:
        return img / 255.0
    if mode == 'min-max':
        return (img - img.min()) / max(img.max() - img.min(), 1e-7)
    if mode == 'z-score':
        return (img - img.mean()) / max(img.std(), 1e-7)
    raise ValueError("only supported [None, 'min-max', 'z-score'] mode")


def denorm(img):
    """data/transform.py:32-35 -- clip to [0,1], *255, uint8 HWC"""
    arr = img.detach().float().cpu().numpy() if torch.is_tensor(img) else np.asarray(img)
    arr = np.clip(arr, 0.0, 1.0) * 255.0
    if arr.ndim == 3:
        arr = arr.transpose(1, 2, 0)
    return arr.round().astype(np.uint8)


def save_result(pred, img1=None, img2=None):
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
    """linear warm-up from start_factor*lr to lr over warmup_iters (reference common.py:136-166)"""

    def __init__(self, optimizer, warmup_iters, start_factor=0.001, last_epoch=-1):
        self.warmup_iters = max(1, warmup_iters)
        self.start_factor = start_factor
        super(WarmupLR, self).__init__(optimizer, last_epoch)

    def get_lr(self):
        t = min(self.last_epoch, self.warmup_iters) / self.warmup_iters
        f = self.start_factor + (1.0 - self.start_factor) * t
        return [base * f for base in self.base_lrs]


class Logger(object):
    def __init__(self, path_log):
        self.log_name = os.path.basename(path_log) or 'root'
        self.out_path = path_log
        os.makedirs(os.path.dirname(self.out_path), exist_ok=True)

    def init_logger(self):
        logger = logging.getLogger(self.log_name)
        logger.setLevel(level=logging.INFO)
        if not logger.handlers:
            fh = logging.FileHandler(self.out_path, 'a')
            fh.setFormatter(logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s'))
            ch = logging.StreamHandler()
            logger.addHandler(fh)
            logger.addHandler(ch)
        return logger


def make_logger(out_dir):
    time_str = time.strftime('%Y-%m-%d_%H-%M')
    log_dir = os.path.join(out_dir, time_str)
    os.makedirs(log_dir, exist_ok=True)
    return Logger(os.path.join(log_dir, 'train.log')).init_logger(), log_dir
