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


def _common_extra(parser):
    # additions of this engine (the reference hard-codes both by editing list indices in the script)
    parser.add_argument('--model', default='PFNetv1', type=str, help='PFNetv1 | PFNetv2 | DenseFuse | VIFNet | NestFuse | RFNNest | DeepFuse | DBNet | SEDRFuse | IFCNN | DIFNet | PMGI | UNFusion | MAFusion | Res2Fusion')
    parser.add_argument('--dtype', default='fp32', type=str, help='feature-map storage: fp32 (parity) | bf16 (throughput)')
    parser.add_argument('--synthetic', default=0, type=int, help='>0: train on this many synthetic random pairs (no dataset needed)')
    parser.add_argument('--graph', default=False, type=bool,
                        help='replay forward + losses + backward of the training batch shape as one hipGraph (launch-bound small batches)')


def get_train_args():
    parser = argparse.ArgumentParser(description='Training')
    parser.add_argument('--lr', default=None, type=float, help='learning rate')
    parser.add_argument('--bs', default=None, type=int, help='batch size')
    parser.add_argument('--epoch', default=None, type=int, help='number of epoch')
    parser.add_argument('--use_patches', default=True, type=bool, help='use patches or random crop')
    parser.add_argument('--warmup', default=False, type=bool, help='use warmup lr')
    parser.add_argument('--clip_grad', default=True, type=bool, help='clip grad norm')
    parser.add_argument('--local_rank', default=0, type=int, help='node rank for distributed training')
    parser.add_argument('--local_world_size', default=1, type=int, help='number of gpus for distributed training')
    parser.add_argument('--data', default='polar', type=str, help='dataset folder name')
    _common_extra(parser)
    return parser.parse_args()


def get_test_args():
    parser = argparse.ArgumentParser(description='Testing')
    parser.add_argument('--use_gpu', default=True, type=bool, help='use gpu or cpu')
    parser.add_argument('--data', default='polar', type=str, help='dataset folder name')
    parser.add_argument('--ckpt', default=None, type=str, help='checkpoint folder name')
    _common_extra(parser)
    return parser.parse_args()


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
    def __init__(self):
        self.reset()

    def is_empty(self):
        return self.count == 0

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
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
