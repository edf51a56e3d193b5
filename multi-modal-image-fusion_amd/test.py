# -*- coding: utf-8 -*-
"""Model testing -- entry point mirroring the reference's test.py (:31-69 test_model, :72-188 main):
full-resolution batch-1 fusion through the HIP engine, per-image SSIM and wall time (device
synchronised, unlike the reference's timer), `NN.bmp` outputs when cv2 is present."""
import os
import sys
import time

BASE_DIR = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, BASE_DIR)

import torch

from common import *
from core.metric import calc_ssim
from core.model import *
from mmif import engine as E


def test_model(model, pairs, save_dir=None):
    times, ssims = [], []
    for idx, (img1, img2) in enumerate(pairs):
        img1, img2 = img1.to(device), img2.to(device)
        torch.cuda.synchronize(device)
        t0 = time.time()
        with torch.no_grad():
            imgf = model(img1, img2)
        torch.cuda.synchronize(device)
        if idx > 0:
            times.append(time.time() - t0)
        with torch.no_grad():   # reference test.py:49-52
            s = ((calc_ssim(img1, imgf, data_range=1.0) + calc_ssim(img2, imgf, data_range=1.0)) * 0.5).item()
        ssims.append(s)
        if save_dir is not None:
            try:
                import cv2
                cv2.imwrite(os.path.join(save_dir, f'{idx + 1:0>2}.bmp'), save_result(imgf[0]))
            except ImportError:
                pass
    avg_t = sum(times) / max(1, len(times))
    return sum(ssims) / max(1, len(ssims)), avg_t


if __name__ == '__main__':
    args = get_test_args()
    assert torch.cuda.is_available(), 'the HIP engine needs a GPU'
    device = torch.device('cuda', 0)
    E.set_compute_dtype(args.dtype)
    model = {'PFNetv1': PFNetv1, 'PFNetv2': PFNetv2, 'DenseFuse': DenseFuse, 'VIFNet': VIFNet, 'NestFuse': NestFuse, 'RFNNest': RFNNest,
             'DeepFuse': DeepFuse, 'DBNet': DBNet, 'SEDRFuse': SEDRFuse, 'IFCNN': IFCNN, 'DIFNet': DIFNet, 'PMGI': PMGI, 'UNFusion': UNFusion, 'MAFusion': MAFusion, 'Res2Fusion': Res2Fusion}[args.model]().to(device)
    if args.ckpt is not None:
        ckpt = os.path.join(BASE_DIR, '..', 'checkpoints', args.ckpt, 'epoch_best.pth')
        assert os.path.isfile(ckpt), f'{ckpt} is not a file'
        model.load_state_dict(torch.load(ckpt, map_location='cpu'), strict=False)
    model.eval()
    if args.synthetic > 0:
        g = torch.Generator().manual_seed(0)
        pairs = [(torch.rand(1, 1, 1024, 1224, generator=g), torch.rand(1, 1, 1024, 1224, generator=g)) for _ in range(args.synthetic)]
    else:
        from data.dataset import FusionDataset
        data_dir = os.path.join(BASE_DIR, '..', 'datasets', args.data)
        assert os.path.isdir(data_dir), f'{data_dir} is not a dir (use --synthetic N)'
        ds = FusionDataset(data_dir, 'test')
        pairs = [(a.unsqueeze(0), b.unsqueeze(0)) for a, b in ds]
    ssim, t = test_model(model, pairs)
    print(f'ssim: {ssim:.4f}, time: {t:.4f}s, fps: {1.0 / max(t, 1e-9):.2f}')
