# -*- coding: utf-8 -*-
"""Model testing -- entry point mirroring the reference's test.py (test_model :31-69, main :72-188): full-resolution
batch-1 fusion of <datasets>/<data>/[test/]{vis,ir} through the HIP engine, per-image SSIM ((ssim1 + ssim2) / 2 on the
fused SSIM kernel) and forward time, `NN.bmp` outputs under <checkpoint>/<data>/, result lines appended to the
checkpoint's train.log.

    python test.py --data roadscene --ckpt 2023-02-26_23-15 [--model PFNetv1 --dtype fp32|bf16]
    python test.py --synthetic 8          # 8 random 1224x1024 pairs, no dataset / checkpoint needed

Differences, deliberate: the timer is device-synchronised (the reference's measures launch time only), the model is
chosen with --model (the reference edits a list index), images are written through data/_io.py (cv2 when present).
"""
import os
import sys
import time

BASE_DIR = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, BASE_DIR)

import torch
from torch.utils.data import DataLoader

from common import *
from core.metric import calc_ssim
from core.model import *
from data._io import imwrite
from data.dataset import FusionDataset as Dataset
from mmif import engine as E

device = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')
MODELS = {'PFNetv1': PFNetv1, 'PFNetv2': PFNetv2, 'DenseFuse': DenseFuse, 'VIFNet': VIFNet, 'NestFuse': NestFuse, 'RFNNest': RFNNest,
          'DeepFuse': DeepFuse, 'DBNet': DBNet, 'SEDRFuse': SEDRFuse, 'IFCNN': IFCNN, 'DIFNet': DIFNet, 'PMGI': PMGI,
          'UNFusion': UNFusion, 'MAFusion': MAFusion, 'Res2Fusion': Res2Fusion}


def _sync():
    if device.type == 'cuda':
        torch.cuda.synchronize(device)


def test_model(model, data_loader, save_dir=None, file=None):
    timer = AverageMeter()
    ssim = AverageMeter()
    for iter, (img1, img2) in enumerate(data_loader):
        iter_idx = iter + 1
        img1 = img1.to(device, non_blocking=True)
        img2 = img2.to(device, non_blocking=True)
        _sync()
        start_time = time.time()
        with torch.no_grad():
            imgf = model(img1, img2)
        _sync()
        if iter > 0:   # the first image pays the one-off costs (reference test.py:41-48)
            timer.update(time.time() - start_time)
        with torch.no_grad():
            ssim1 = calc_ssim(img1, imgf, data_range=1.0)
            ssim2 = calc_ssim(img2, imgf, data_range=1.0)
            avg_ssim = (ssim1 + ssim2) * 0.5
            ssim.update(avg_ssim.item())
        line = f'iter: {iter_idx:0>2}, ssim: {ssim.val:.4f}, time: {timer.val * 1000:.3f}ms'
        print(line)
        if file is not None:
            file.write('\n' + line)
        if save_dir is not None:
            imwrite(os.path.join(save_dir, f'{iter_idx:0>2}.bmp'), save_result(imgf[0]))
    return ssim.avg, timer.avg


def test_set_name(data):
    """reference test.py:104-108: 'tno' keeps vis/ and ir/ at the dataset root, the others under test/"""
    return None if data in ['tno'] else 'test'


if __name__ == '__main__':
    args = get_test_args()
    assert torch.cuda.is_available(), 'the HIP engine needs a GPU'
    torch.cuda.set_device(device)
    E.set_compute_dtype(args.dtype)

    ckpt_dir = os.path.join(BASE_DIR, '..', 'checkpoints', args.ckpt)
    ckpt_path = os.path.join(ckpt_dir, 'epoch_best.pth')
    if args.synthetic > 0:
        g = torch.Generator().manual_seed(0)
        test_loader = [(torch.rand(1, 1, 1024, 1224, generator=g), torch.rand(1, 1, 1024, 1224, generator=g)) for _ in range(args.synthetic)]
        test_save_dir, log_path = None, None
    else:
        data_dir = os.path.join(BASE_DIR, '..', 'datasets', args.data)
        assert os.path.isdir(data_dir), f'{data_dir} is not a dir (use --synthetic N)'
        assert os.path.isfile(ckpt_path), f'{ckpt_path} is not a file'
        log_path = os.path.join(ckpt_dir, 'train.log')
        assert os.path.isfile(log_path)
        test_save_dir = os.path.join(ckpt_dir, args.data)
        os.makedirs(test_save_dir, exist_ok=True)
        test_set = Dataset(data_dir, set_name=test_set_name(args.data), set_type='test')
        test_loader = DataLoader(test_set, batch_size=1, shuffle=False, num_workers=4, pin_memory=True)

    print(f'model: {args.model}')
    model = MODELS[args.model]().to(device, non_blocking=True)
    params = sum([param.numel() for param in model.parameters()])
    print(f'params: {params / 1e6:.3f}M')
    if os.path.isfile(ckpt_path):
        model.load_state_dict(torch.load(ckpt_path, map_location=device), strict=False)
    model.eval()

    file = open(log_path, 'a') if log_path is not None else None
    try:
        ssim, avg_time = test_model(model, test_loader, test_save_dir, file)
        summary = f'ssim: {ssim:.4f}, time: {avg_time * 1000:.3f}ms, fps: {1.0 / max(avg_time, 1e-9):.3f}'
        print(summary)
        if file is not None:
            file.write('\n' + summary)
    finally:
        if file is not None:
            file.close()
