# -*- coding: utf-8 -*-
"""Model training -- entry point mirroring the reference's train.py (same flags, log lines and
checkpoint names; train_model() follows train.py:37-133 step for step) on the MI355X HIP engine.

    python train.py --data roadscene [--lr --bs --epoch --use_patches --warmup --clip_grad]
    python -m torch.distributed.run --nproc-per-node 8 train.py --data roadscene      # data parallel

Differences, all deliberate: the model is chosen with --model (the reference edits a
list index), CUDA_VISIBLE_DEVICES is not overridden, ranks come from torchrun's environment as well as
--local_rank/--local_world_size, gradients are all-reduced by ONE RCCL collective inside the fused
optimiser (mmif.optim) instead of DDP buckets + four scalar all-reduces, and --synthetic N trains on N
random pairs when the image datasets / cv2 are not available.
"""
import os
import sys
import time

BASE_DIR = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, BASE_DIR)

import torch
import torch.distributed as dist
from torch.optim.lr_scheduler import MultiStepLR
from torch.utils.data import DataLoader, DistributedSampler, TensorDataset

from common import *
from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss, unit_gradient
from core.model import *
from data._io import imwrite
from mmif import engine as E
from mmif.dist import broadcast_parameters
from mmif.optim import FusedClipAdam


_fused = {}


def _losses(loss_fn1, loss_fn2, loss_fn3, img1, img2, imgf):
    """loss1 + loss2 + loss3 of the reference's step (train.py:64-69); with SSIMLoss('ssim') as ONE device call (core.loss.FusionLoss:
    the same kernels, their sum and the sum of their gradients without torch glue kernels)"""
    if getattr(loss_fn1, 'mode', None) == 'ssim' and not getattr(loss_fn1, 'use_padding', False):
        key = (id(loss_fn1), id(loss_fn2), id(loss_fn3))
        fl = _fused.get(key)
        if fl is None:
            fl = _fused[key] = FusionLoss(loss_fn1, loss_fn2, loss_fn3, 'max', 'max')
        total = fl(img1, img2, imgf)
        return total, fl.values[1], fl.values[2], fl.values[3]
    l1, l2, l3 = loss_fn1(img1, img2, imgf), loss_fn2(img1, img2, imgf, mode='max'), loss_fn3(img1, img2, imgf, mode='max')
    return l1 + l2 + l3, l1, l2, l3


graphed = None   # GraphedStep of the training batch shape (--graph True)


def train_model(model, data_loader, loss_fn1, loss_fn2, loss_fn3, epoch, mode='train', save_dir=None):
    global graphed
    use_graph = bool(getattr(args, 'graph', False))
    loss = AverageMeter()
    epoch_idx = epoch + 1
    torch.cuda.synchronize(device)
    start_time = time.time()
    imgf = img1 = img2 = None
    for it, (img1, img2) in enumerate(data_loader):
        iter_idx = it + 1
        img1 = img1.to(device, non_blocking=True)
        img2 = img2.to(device, non_blocking=True)
        if mode == 'train' and graphed is not None and graphed.matches(img1, img2):
            # launch-bound batches: forward + losses + backward replayed as one hipGraph (mmif/graph.py)
            graphed(img1, img2)
            imgf = graphed.imgf
            total_loss, loss1, loss2, loss3 = optimizer.reduced_scalars.unbind(0)
            if args.warmup and epoch < 1:
                warmup_scheduler.step()
        elif mode == 'train':
            if use_graph and graphed is None:
                from mmif.graph import GraphedStep
                graphed = GraphedStep(model, lambda a, b, f: _losses(loss_fn1, loss_fn2, loss_fn3, a, b, f), optimizer, img1, img2)
            optimizer.zero_grad(set_to_none=True)
            imgf = model(img1, img2)
            total_loss, loss1, loss2, loss3 = _losses(loss_fn1, loss_fn2, loss_fn3, img1, img2, imgf)
            if hasattr(optimizer, "stage_scalars"):   # (data parallel) the loss values ride in the early gradient all-reduce
                optimizer.stage_scalars([total_loss, loss1, loss2, loss3])
            total_loss.backward(unit_gradient(total_loss))   # (= .backward(): a cached ones tensor instead of a fill kernel per step)
            # clip_grad_norm_(5) + Adam + (distributed) gradient & loss all-reduce: one fused step
            optimizer.step(scalars=[total_loss, loss1, loss2, loss3])
            total_loss, loss1, loss2, loss3 = optimizer.reduced_scalars.unbind(0)
            if args.warmup and epoch < 1:
                warmup_scheduler.step()
        else:
            with torch.no_grad():
                imgf = model(img1, img2)
                total_loss, loss1, loss2, loss3 = _losses(loss_fn1, loss_fn2, loss_fn3, img1, img2, imgf)
            if is_distributed:
                vals = torch.stack([total_loss, loss1, loss2, loss3])
                total_loss, loss1, loss2, loss3 = reduce_value(vals, world_size).unbind(0)
        loss.update(total_loss.item(), len(imgf))
        if local_rank == 0 and iter_idx % 10 == 0:
            logger.info(f'epoch: {epoch_idx:0>2}, iter: {iter_idx:0>3}, {mode} loss: {loss.avg:.4f}')
    torch.cuda.synchronize(device)
    cost_time = time.time() - start_time
    if local_rank == 0:
        logger.info(f'cost time: {cost_time:.3f}s\n')
        if save_dir is not None and imgf is not None:
            imwrite(os.path.join(save_dir, f'{epoch_idx:0>2}.png'), save_result(imgf[0], img1[0], img2[0]))
    if is_distributed:
        dist.barrier()
    return loss.avg


def make_loaders():
    per_rank = batch_size // world_size
    set_name = None if args.data in ['tno'] else 'train'     # reference train.py:181-184
    if args.use_patches:
        # patch training: the uint8 patch bank lives in HBM, batches (norm + dihedral augmentation) are produced on the
        # device (mmif.feed / csrc/feed.hip) -- no DataLoader workers, collate, pin_memory or H2D copies
        from mmif.feed import DevicePatchFeed
        if args.synthetic > 0:
            g = torch.Generator().manual_seed(0)
            def bank(n):
                return torch.randint(0, 256, (n, 64, 64), generator=g, dtype=torch.uint8)
            nv = max(batch_size, args.synthetic // 8)
            tr = DevicePatchFeed(bank(args.synthetic), bank(args.synthetic), per_rank, device, transform=True, seed=0, rank=rank,
                                 world_size=world_size, drop_last=True)
            va = DevicePatchFeed(bank(nv), bank(nv), per_rank, device, shuffle=False, rank=rank, world_size=world_size)
        else:
            from data.patches import FusionPatches
            data_dir = os.path.join(BASE_DIR, '..', 'datasets', args.data)
            assert os.path.isdir(data_dir), f'{data_dir} is not a dir (use --synthetic N to train without a dataset)'
            tr = FusionPatches(data_dir, set_name=set_name, set_type='train', transform=True).device_feed(
                per_rank, device, shuffle=True, rank=rank, world_size=world_size, drop_last=True)
            va = FusionPatches(data_dir, set_name=set_name, set_type='valid').device_feed(
                per_rank, device, shuffle=False, rank=rank, world_size=world_size)
        return tr, va, tr   # the feed is its own (distributed) sampler: set_epoch()
    if args.synthetic > 0:
        g = torch.Generator().manual_seed(0)
        def ds(n):
            return TensorDataset(torch.rand(n, 1, 256, 256, generator=g), torch.rand(n, 1, 256, 256, generator=g))
        train_set, valid_set = ds(args.synthetic), ds(max(batch_size, args.synthetic // 8))
    else:
        # whole-image training (reference train.py:193-202): data/dataset.py mirrors the reference's FusionDataset
        data_dir = os.path.join(BASE_DIR, '..', 'datasets', args.data)
        assert os.path.isdir(data_dir), f'{data_dir} is not a dir (use --synthetic N to train without a dataset)'
        from data.dataset import FusionDataset as Data
        train_set = Data(data_dir, set_name=set_name, set_type='train', transform=True, fix_size=True)
        valid_set = Data(data_dir, set_name=set_name, set_type='valid', fix_size=True)
    tr_sampler = DistributedSampler(train_set) if is_distributed else None
    va_sampler = DistributedSampler(valid_set, shuffle=False) if is_distributed else None
    tr = DataLoader(train_set, batch_size=per_rank, shuffle=tr_sampler is None, sampler=tr_sampler, num_workers=4, pin_memory=True, drop_last=True)
    va = DataLoader(valid_set, batch_size=per_rank, shuffle=False, sampler=va_sampler, num_workers=4, pin_memory=True)
    return tr, va, tr_sampler


if __name__ == '__main__':
    setup_seed(0)
    args = get_train_args()
    lr, batch_size, num_epochs = args.lr, args.bs, args.epoch
    milestones = (round(num_epochs * 2 / 3), round(num_epochs * 8 / 9))

    env_world = int(os.environ.get('WORLD_SIZE', '1'))
    is_distributed = args.local_world_size > 1 or env_world > 1
    if is_distributed:
        rank, world_size = setup_dist(args.local_rank, args.local_world_size)
        local_rank = int(os.environ.get('LOCAL_RANK', args.local_rank))
    else:
        rank, world_size, local_rank = 0, 1, 0
    assert torch.cuda.is_available(), 'the HIP engine needs a GPU'
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    E.set_compute_dtype(args.dtype)

    log_dir, logger = make_logger(BASE_DIR) if local_rank == 0 else (None, None)
    train_save_dir = valid_save_dir = None
    if local_rank == 0:   # reference train.py:172-178
        train_save_dir, valid_save_dir = os.path.join(log_dir, 'train'), os.path.join(log_dir, 'valid')
        os.makedirs(train_save_dir, exist_ok=True)
        os.makedirs(valid_save_dir, exist_ok=True)
    train_loader, valid_loader, train_sampler = make_loaders()

    model = {'PFNetv1': PFNetv1, 'PFNetv2': PFNetv2, 'DenseFuse': DenseFuse, 'VIFNet': VIFNet, 'NestFuse': NestFuse, 'RFNNest': RFNNest,
             'DeepFuse': DeepFuse, 'DBNet': DBNet, 'SEDRFuse': SEDRFuse, 'IFCNN': IFCNN, 'DIFNet': DIFNet, 'PMGI': PMGI, 'UNFusion': UNFusion, 'MAFusion': MAFusion, 'Res2Fusion': Res2Fusion}[args.model]().to(device)
    if is_distributed:
        broadcast_parameters(model, 0)  # replaces the reference's init_weights.pth + DDP constructor broadcast

    loss_fn1 = SSIMLoss('ssim', weight=1.0)
    loss_fn2 = PixelLoss('l1', weight=0.01)
    loss_fn3 = GradLoss('l1', weight=0.1).to(device)
    optimizer = FusedClipAdam(model.parameters(), lr=lr, betas=(0.9, 0.999), max_norm=5.0 if args.clip_grad else None)
    epoch_scheduler = MultiStepLR(optimizer, milestones=milestones, gamma=0.1)
    if args.warmup:
        warmup_scheduler = WarmupLR(optimizer, 0.001, len(train_loader))   # reference train.py:323

    best_loss, best_epoch = 0.0, 0
    # everything alive after set-up goes to the permanent generation: CPython's full collections then no longer walk the ~million objects
    # torch's import left behind (a 75-150 ms pause of the launch thread every few hundred steps, during which the GPU runs dry; found with
    # bench.py in round 6).  The collector itself stays on.
    import gc
    gc.collect()
    gc.freeze()
    for epoch in range(num_epochs):
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
        model.train()
        train_loss = train_model(model, train_loader, loss_fn1, loss_fn2, loss_fn3, epoch, 'train', train_save_dir)
        model.eval()
        valid_loss = train_model(model, valid_loader, loss_fn1, loss_fn2, loss_fn3, epoch, 'valid', valid_save_dir)
        epoch_scheduler.step()
        if local_rank == 0:
            logger.info(f'epoch: {epoch + 1:0>2}, train loss: {train_loss:.4f}, valid loss: {valid_loss:.4f}, lr: {optimizer.param_groups[0]["lr"]}')
            if epoch >= num_epochs // 2 and (best_loss == 0.0 or valid_loss < best_loss):
                best_loss, best_epoch = valid_loss, epoch + 1
                torch.save(model.state_dict(), os.path.join(log_dir, 'epoch_best.pth'))
    if local_rank == 0:
        torch.save(model.state_dict(), os.path.join(log_dir, 'epoch_last.pth'))
        logger.info(f'training done, best loss: {best_loss:.4f}, in epoch: {best_epoch}')
    if is_distributed:
        dist.destroy_process_group()
