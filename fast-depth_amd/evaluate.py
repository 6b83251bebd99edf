#!/usr/bin/env python3
"""Evaluation harness with the reference's CLI and printout (reference main.py:26-127, utils.py:12-34), for the MI355X path.

    python3 evaluate.py --evaluate [path_to_trained_model] [--gpu N] [-p FREQ] [--batch-size B] [--samples DIR|FILE.npz]

Differences from the reference, all deliberate:
  * `t_GPU` is measured with the device synchronised (the reference's synchronize() calls are commented out at main.py:69,76,
    so it prints launch latency);
  * any batch size (the reference validates with batch 1, main.py:40-41);
  * metrics come from ONE fused device reduction per batch (metrics.py of this package) instead of ~12 host syncs;
  * the reference's dataloader classes are not rebuilt (dataset absent; dataloaders/ depends on removed SciPy/NumPy APIs): samples are
    the reference's `.h5` frames (read with h5py when it is installed -- it is not in this image) or `.npz` files holding `rgb` and `depth` -- either RAW frames ([480,640,3] uint8 + [480,640] float32
    metres), which go through the reference's val_transform as ONE device gather (dataloaders/nyu.py, pinned against PIL), or
    frames already at the network resolution ([H,W,3] uint8 or float in [0,1]) -- or, by default (no --samples), `--repeat` seeded
    synthetic NYU-shaped frames (noise images with depths in [0.7, 10] m: they exercise the loop and its printout, nothing more).
A checkpoint is the reference's pickle ({'epoch','best_result','model'} or a bare module, main.py:49-57); without one, a seeded
random-weight model is evaluated (useful only as a smoke test of the loop).
"""
import argparse
import glob
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
import models  # noqa: E402
from metrics import AverageMeter, Result  # noqa: E402


def parse_command(argv=None):
    parser = argparse.ArgumentParser(description='FastDepth (MI355X path)')
    parser.add_argument('--data', metavar='DATA', default='nyudepthv2', choices=['nyudepthv2'],
                        help='dataset: nyudepthv2 (default: nyudepthv2)')
    parser.add_argument('--modality', '-m', metavar='MODALITY', default='rgb', choices=['rgb'], help='modality: rgb (default: rgb)')
    parser.add_argument('-j', '--workers', default=16, type=int, metavar='N', help='accepted for CLI compatibility; samples are read in-process')
    parser.add_argument('--print-freq', '-p', default=50, type=int, metavar='N', help='print frequency (default: 50)')
    parser.add_argument('-e', '--evaluate', default='', type=str, metavar='PATH')
    parser.add_argument('--gpu', default='0', type=str, metavar='N', help="gpu id")
    parser.add_argument('--batch-size', default=1, type=int)
    parser.add_argument('--samples', default='', help='directory of .npz samples or one .npz file (default: seeded synthetic NYU-shaped frames)')
    parser.add_argument('--repeat', default=8, type=int, help='how many synthetic frames the default (no --samples) run evaluates')
    parser.add_argument('--dtype', default='f32', choices=['f32', 'f16', 'bf16'], help='activation storage inside the engine')
    return parser.parse_args(argv)


def load_samples(args):
    files = []
    if args.samples:
        if os.path.isdir(args.samples):
            files = sorted(glob.glob(os.path.join(args.samples, '*.npz')) + glob.glob(os.path.join(args.samples, '**', '*.h5'), recursive=True))
        else:
            files = [args.samples]
    if files:
        out = []
        for f in files:
            if f.endswith('.h5'):
                # the reference's NYU-Depth-v2 files (dataloaders/dataloader.py:8-13 h5_loader: 'rgb' [3,480,640] uint8, 'depth' [480,640])
                try:
                    import h5py
                except ImportError:
                    raise RuntimeError("%s is an HDF5 sample but h5py is not installed; convert it to .npz (rgb [480,640,3] uint8, depth [480,640] float32)" % f)
                with h5py.File(f, 'r') as h5f:
                    z = {'rgb': np.transpose(np.array(h5f['rgb']), (1, 2, 0)), 'depth': np.array(h5f['depth'])}
            else:
                z = np.load(f)
            if z['rgb'].dtype == np.uint8 and z['rgb'].shape[:2] == (480, 640):
                # raw NYU frame: kept as uint8 HWC; validate() runs the reference's val_transform on the GPU (dataloaders/nyu.py)
                out.append((torch.from_numpy(z['rgb']), torch.from_numpy(z['depth'].astype(np.float32))))
                continue
            rgb = z['rgb'].astype(np.float32)
            if rgb.max() > 1.5:
                rgb = rgb / 255.0
            out.append((torch.from_numpy(rgb).permute(2, 0, 1).contiguous(), torch.from_numpy(z['depth'].astype(np.float32))[None]))
        return out
    # no --samples: synthetic NYU-shaped frames (seeded), enough to exercise the loop and its printout; the product reads nothing
    # from tests/
    g = np.random.default_rng(0)
    out = []
    for _ in range(args.repeat):
        rgb = g.random((3, 224, 224), dtype=np.float32)
        depth = (0.7 + 9.3 * g.random((1, 224, 224), dtype=np.float32)).astype(np.float32)
        out.append((torch.from_numpy(rgb), torch.from_numpy(depth)))
    return out


# printout tables: (label, Result attribute, precision) -- the text they produce is the reference's (main.py:100-119)
_PROGRESS = (("RMSE", "rmse", 2), ("MAE", "mae", 2), ("Delta1", "delta1", 3), ("REL", "absrel", 3), ("Lg10", "lg10", 3))
_SUMMARY = (("RMSE", "rmse"), ("MAE", "mae"), ("Delta1", "delta1"), ("REL", "absrel"), ("Lg10", "lg10"))


def _progress_line(i, n, gpu_time, result, average):
    head = "Test: [%d/%d]\tt_GPU=%.3f(%.3f)\n\t" % (i, n, gpu_time, average.gpu_time)
    return head + "".join("%s=%.*f(%.*f) " % (label, prec, getattr(result, attr), prec, getattr(average, attr)) for label, attr, prec in _PROGRESS)


def _summary(avg):
    return "\n*\n" + "".join("%s=%.3f\n" % (label, getattr(avg, attr)) for label, attr in _SUMMARY) + "t_GPU=%.3f\n" % avg.gpu_time


def validate(samples, model, args, device):
    """The reference's validate() loop (main.py:63-119) over in-memory samples."""
    average_meter = AverageMeter()
    model.eval()
    n_batches = (len(samples) + args.batch_size - 1) // args.batch_size
    end = time.time()
    for i in range(n_batches):
        chunk = samples[i * args.batch_size:(i + 1) * args.batch_size]
        inp = torch.stack([c[0] for c in chunk]).to(device, non_blocking=True)
        target = torch.stack([c[1] for c in chunk]).to(device, non_blocking=True)
        if inp.dtype == torch.uint8:                     # raw 480 x 640 frames: val_transform (nyu.py:48-59) as one device gather
            from dataloaders.nyu import GpuValTransform
            tf = validate.__dict__.setdefault("_tf", GpuValTransform((224, 224), device))
            inp, target = tf(inp, target)
        torch.cuda.synchronize(device)
        data_time = time.time() - end
        end = time.time()
        with torch.no_grad():
            pred = model(inp)
        torch.cuda.synchronize(device)
        gpu_time = time.time() - end
        # per-image metrics, averaged per image as the reference's batch-size-1 loop does (main.py:40-41, 80-82); times are per frame
        nb = inp.size(0)
        for result in Result.evaluate_frames(pred.data, target.data):
            average_meter.update(result, gpu_time / nb, data_time / nb, 1)
        end = time.time()
        if (i + 1) % args.print_freq == 0:
            print(_progress_line(i + 1, n_batches, gpu_time, result, average_meter.average()))
    avg = average_meter.average()
    print(_summary(avg))
    return avg


def main(argv=None):
    args = parse_command(argv)
    print(args)
    device = torch.device('cuda', int(args.gpu))
    if args.evaluate:
        assert os.path.isfile(args.evaluate), "=> no model found at '{}'".format(args.evaluate)
        print("=> loading model '{}'".format(args.evaluate))
        checkpoint = torch.load(args.evaluate, weights_only=False)
        if type(checkpoint) is dict:
            print("=> loaded best model (epoch {})".format(checkpoint.get('epoch')))
            model = checkpoint['model']
        else:
            model = checkpoint
    else:
        print("=> no checkpoint given: evaluating a seeded random-weight model (loop smoke test)")
        torch.manual_seed(0)
        model = models.MobileNetSkipAdd((224, 224), pretrained=False)
        model.decode_conv6[1].bias.data.fill_(2.8)
    model = model.to(device)
    if args.dtype != 'f32':
        model.set_compute_dtype({'f16': torch.float16, 'bf16': torch.bfloat16}[args.dtype])
    return validate(load_samples(args), model, args, device)


if __name__ == '__main__':
    main()
