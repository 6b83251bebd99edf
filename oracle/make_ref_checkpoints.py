"""Checkpoints pickled by the REFERENCE'S OWN classes (imported from /root/reference), for the drop-in proof of SURVEY.md rows b / f-4:
`torch.load` of such a file under the product's `models` / `metrics` modules must yield the product class and run.
TEST INFRASTRUCTURE; usage:  python -m oracle.make_ref_checkpoints [--full]

  tests/golden/ref_ckpt_tiny_pruned.pth.tar   COMMITTED (small): {'epoch', 'best_result', 'model'} exactly as the reference's entry point reads
                                  it (main.py:49-57): `model` = a reference `models.MobileNetSkipAdd` instance whose sub-modules carry
                                  irregular (pruned-like) widths -- encoder units are the nn.Sequential triples the reference's conv_bn / conv_dw
                                  closures build (imagenet/mobilenet.py:22-38), decoder units come from the reference's own
                                  `models.depthwise` / `models.pointwise` (models.py:61-75); `best_result` = a reference `metrics.Result`.
  tests/golden/ref_ckpt_tiny_io.npz           the input and the REFERENCE'S output for that instance (64 x 64, batch 2)
  --full: tests/golden/_ref/ref_ckpt_{unpruned,pruned}.pth.tar + ref_ckpt_pruned_io.npz   (git-ignored: 16 MB / 5 MB; they travel to the GPU box
                                  with the snapshot).  `unpruned` is the calibrated seed-0 model of make_golden.py, whose reference output is
                                  the committed tests/golden/base_s0_out.npy; `pruned` has the published pruned widths.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

from oracle import make_golden
from oracle.inputs import batch_variants

GOLD = make_golden.GOLD
TINY = ((16, 56, 88, 120, 144, 72, 104, 40, 72, 88, 96, 128, 80, 112), (200, 72, 120, 56, 16, 1))     # multiples of 8, mostly not of 32
PRUNED = ((16, 56, 88, 120, 144, 256, 408, 376, 272, 288, 296, 328, 480, 512), (200, 256, 120, 56, 16, 1))


def _enc_unit(cin, cout, stride, first):
    """What the reference's conv_bn / conv_dw closures return (imagenet/mobilenet.py:22-38), with free widths."""
    if first:
        return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))
    return nn.Sequential(nn.Conv2d(cin, cin, 3, stride, 1, groups=cin, bias=False), nn.BatchNorm2d(cin), nn.ReLU6(inplace=True),
                         nn.Conv2d(cin, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout), nn.ReLU6(inplace=True))


def pruned_reference_instance(ref_models, size, channels, seed):
    """A reference MobileNetSkipAdd whose sub-modules were replaced the way a pruning tool (NetAdapt, README.md:25) leaves them."""
    enc, dec = channels
    torch.manual_seed(seed)
    m = ref_models.MobileNetSkipAdd(size, pretrained=False)
    strides = (2, 1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1)
    cin = 3
    for i in range(14):
        setattr(m, "conv%d" % i, _enc_unit(cin, enc[i], strides[i], i == 0))
        cin = enc[i]
    skips = (enc[5], enc[3], enc[1])                       # conv5 / conv3 / conv1 outputs are added after decode_conv2 / 3 / 4
    assert (dec[1], dec[2], dec[3]) == skips, "decoder widths must match the skip tensors"
    for j in range(5):
        setattr(m, "decode_conv%d" % (j + 1), nn.Sequential(ref_models.depthwise(cin, 5), ref_models.pointwise(cin, dec[j])))
        cin = dec[j]
    m.decode_conv6 = ref_models.pointwise(cin, 1)
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, nn.Conv2d):
            n = mod.kernel_size[0] * mod.kernel_size[1] * mod.out_channels
            mod.weight.data.copy_(torch.randn(mod.weight.shape, generator=g) * (2.0 / n) ** 0.5)      # the rule of the reference's weights_init (models.py:36-50)
        if isinstance(mod, nn.BatchNorm2d):
            mod.weight.data.copy_(0.5 + torch.rand(mod.weight.shape, generator=g))
            mod.bias.data.copy_(0.3 * torch.randn(mod.bias.shape, generator=g))
            mod.running_mean.copy_(0.2 * torch.randn(mod.running_mean.shape, generator=g))
            mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
    m.decode_conv6[1].bias.data.fill_(2.0)
    return m.eval()


def main():
    ref_models, ref_metrics = make_golden.import_reference()
    best = ref_metrics.Result()
    best.update(0.01, 0.008, 0.3, 0.55, 0.4, 0.16, 0.07, 0.77, 0.94, 0.98, 0.005, 0.001)
    # reference classes are pickled by module path ("models", "metrics"): they must be importable under those names while saving
    sys.modules["models"], sys.modules["metrics"] = ref_models, ref_metrics
    tiny = pruned_reference_instance(ref_models, (64, 64), TINY, 31)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(32))
    with torch.no_grad():
        y = tiny(x)
    torch.save({"epoch": 3, "best_result": best, "model": tiny}, os.path.join(GOLD, "ref_ckpt_tiny_pruned.pth.tar"))
    np.savez(os.path.join(GOLD, "ref_ckpt_tiny_io.npz"), x=x.numpy(), y=y.numpy())
    print("tiny pruned reference checkpoint:", os.path.getsize(os.path.join(GOLD, "ref_ckpt_tiny_pruned.pth.tar")), "bytes; output range", float(y.min()), float(y.max()))
    if "--full" in sys.argv:
        out = os.path.join(GOLD, "_ref")
        os.makedirs(out, exist_ok=True)
        sample = torch.from_numpy(np.load(os.path.join(GOLD, "sample_rgb_u8.npy")).astype(np.float64) / 255.0).permute(2, 0, 1).float()[None]
        m = make_golden.calibrated(ref_models, 0, sample, "base")
        torch.save({"epoch": 0, "best_result": best, "model": m}, os.path.join(out, "ref_ckpt_unpruned.pth.tar"))
        xb = batch_variants(sample, 4, seed=0)
        with torch.no_grad():
            yb = m(xb)
        want = np.load(os.path.join(GOLD, "base_s0_out.npy"))
        assert np.array_equal(yb.numpy(), want[:4]), "the regenerated seed-0 model does not reproduce the committed golden output"
        pr = pruned_reference_instance(ref_models, (224, 224), PRUNED, 41)
        xp = batch_variants(sample, 2, seed=3)
        with torch.no_grad():
            yp = pr(xp)
        torch.save({"epoch": 9, "best_result": best, "model": pr}, os.path.join(out, "ref_ckpt_pruned.pth.tar"))
        np.savez(os.path.join(out, "ref_ckpt_pruned_io.npz"), x=xp.numpy(), y=yp.numpy())
        print("full-size reference checkpoints written to", out)


if __name__ == "__main__":
    main()
