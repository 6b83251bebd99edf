"""Drop-in `models` module for the FastDepth MobileNet-NNConv5(dw)+skip-add hot path on MI355X.

This file is the host-side mirror of the reference's operator interface for ONE path:

    reference  /root/reference/models.py:654-732   class MobileNetSkipAdd
               /root/reference/models.py:61-75     depthwise / pointwise building blocks
               /root/reference/models.py:36-50     weights_init
               /root/reference/main.py:49-57,75    how the harness obtains and calls the module

It keeps the reference's public surface -- constructor signature, attribute tree
(``conv0..conv13``, ``decode_conv1..decode_conv6``), the 228 ``state_dict`` keys, ``.train()`` /
``.eval()``, and the top-level module name ``models`` that reference-format checkpoints pickle the
class under -- while ``forward`` is *not* a chain of ``torch.nn`` calls: it hands the whole network
to the hand-written HIP engine behind the C-ABI in ``include/fastdepth_hip.h`` (one fused-kernel plan
per (batch, H, W, dtype)).  Sub-modules only *own* parameters and buffers.

There is deliberately no CPU / eager fallback here.  ``forward`` on a non-GPU tensor, or with the
HIP library missing, raises.  The CPU restatement used by the tests lives in ``oracle/`` and is never
imported by this package.
"""
import math

import torch
import torch.nn as nn

import imagenet.mobilenet as _mobilenet

__all__ = ["MobileNetSkipAdd", "MobileNetSkipConcat", "MobileNet", "NNConv", "choose_decoder", "depthwise", "pointwise", "weights_init", "PRUNED_CHANNELS"]

# Channel plan of `mobilenet-nnconv5dw-skipadd-pruned`, reconstructed from the reference's TVM tuning
# log (tvm_compile/tuning/tx2-gpu.mobilenet-nnconv5dw-skipadd-pruned.trials=2000.stop=600.log:1-38,
# SURVEY.md Appendix B): 14 encoder widths, 6 decoder widths.
PRUNED_CHANNELS = ((16, 56, 88, 120, 144, 256, 408, 376, 272, 288, 296, 328, 480, 512),
                   (200, 256, 120, 56, 16, 1))
DEFAULT_DECODER = (512, 256, 128, 64, 32, 1)


def weights_init(m):
    """He-normal init for conv layers, (1, 0) for BatchNorm affine -- same distribution and the same
    RNG consumption as reference models.py:36-50.  Like the reference it acts on the module it is
    handed (use ``module.apply(weights_init)`` for a tree)."""
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        fan = m.out_channels if isinstance(m, nn.Conv2d) else m.in_channels
        m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * fan)))
        if m.bias is not None:
            m.bias.data.zero_()
    elif isinstance(m, nn.BatchNorm2d):
        m.weight.data.fill_(1)
        m.bias.data.zero_()


def _bn_relu(channels):
    return [nn.BatchNorm2d(channels), nn.ReLU(inplace=True)]


def depthwise(in_channels, kernel_size):
    """k x k depthwise conv (pad (k-1)/2, no bias) + BN + ReLU; reference models.py:61-68."""
    if kernel_size % 2 != 1:
        raise AssertionError("parameters incorrect. kernel={}".format(kernel_size))
    return nn.Sequential(
        nn.Conv2d(in_channels, in_channels, kernel_size, stride=1, padding=kernel_size // 2,
                  bias=False, groups=in_channels), *_bn_relu(in_channels))


def pointwise(in_channels, out_channels):
    """1x1 conv (no bias) + BN + ReLU; reference models.py:70-75."""
    return nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, 1, 0, bias=False), *_bn_relu(out_channels))


class _HipForward(nn.Module):
    """Shared HIP-engine plumbing of the drop-in model classes: `forward` hands the whole network to the hand-written engine;
    there is no CPU / eager path."""

    def _engine(self):
        eng = self.__dict__.get('_fd_engine')
        if eng is None:
            from fastdepth_hip.engine import Engine  # raises loudly if libfastdepth_hip.so is absent
            eng = Engine(self)
            self.__dict__['_fd_engine'] = eng          # not a Module attribute: never pickled/state_dict'd
        return eng

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_fd_engine', None)
        return state

    def set_compute_dtype(self, dtype):
        """MI355X extension (not in the reference): storage type of the activations / pointwise weights inside the HIP
        engine -- torch.float32 (default), torch.float16 or torch.bfloat16.  The module's parameters, its input and its
        output stay float32; accumulation is fp32."""
        self._engine().set_dtype(dtype)
        return self

    def repack(self):
        """Drop cached packed weights (call after in-place edits the version counters cannot see)."""
        eng = self.__dict__.get('_fd_engine')
        if eng is not None:
            eng.invalidate()

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("fast-depth_amd: {}.forward runs on an MI355X (HIP) device only; got a {} tensor. "
                               "There is no CPU fallback in this package.".format(type(self).__name__, x.device))
        return self._engine().forward(x)


class NNConv(nn.Module):
    """Nearest-neighbour-upsampling decoder, depthwise-separable form (reference models.py:224-270 with dw=True):
    conv1..conv5 = Sequential(depthwise(C, k), pointwise(C, C/2)) for C = 1024..64, conv6 = pointwise(32, 1); the reference's
    forward interleaves a nearest x2 after conv1..conv5.  Parameter container only (the HIP engine executes it as part of
    `MobileNet`).  The dense variant (dw=False: k x k full convolutions) is outside this package's kernels."""

    def __init__(self, kernel_size, dw):
        super().__init__()
        if not dw:
            raise NotImplementedError("fast-depth_amd implements the depthwise-separable decoders ('nnconv5dw', 'nnconv3dw'); "
                                      "the dense NNConv decoder is not on the accelerated path")
        width = 1024
        for j in range(1, 6):
            setattr(self, 'conv{}'.format(j), nn.Sequential(depthwise(width, kernel_size), pointwise(width, width // 2)))
            width //= 2
        self.conv6 = pointwise(width, 1)


def choose_decoder(decoder):
    """Reference models.py:335-360, restricted to the decoders whose layers are on the accelerated path."""
    if decoder in ('nnconv5dw', 'nnconv3dw'):
        model = NNConv(int(decoder[6]), True)
    else:
        raise NotImplementedError("decoder {!r}: only 'nnconv5dw' / 'nnconv3dw' are built by fast-depth_amd "
                                  "(SURVEY.md 8(f) row f-3)".format(decoder))
    model.apply(weights_init)
    return model


class MobileNet(_HipForward):
    """MobileNet-v1 encoder + decoder WITHOUT skip connections -- `MobileNet(decoder, output_size, in_channels=3,
    pretrained=True)` as in reference models.py:420-460 (runs on the same kernels as
    MobileNetSkipAdd, with `skip = -1` everywhere).  Attribute tree and state_dict keys follow the reference:
    `mobilenet.0 .. mobilenet.13`, `decoder.conv1 .. decoder.conv6`."""

    def __init__(self, decoder, output_size, in_channels=3, pretrained=True):
        super().__init__()
        self.output_size = output_size
        if in_channels != 3:
            raise NotImplementedError("the stem kernel reads 3-channel RGB (the only modality of the reference's main.py)")
        mobilenet = _mobilenet.MobileNet()
        if pretrained:
            import os
            path = os.path.join('imagenet', 'results', 'imagenet.arch=mobilenet.lr=0.1.bs=256', 'model_best.pth.tar')
            state = torch.load(path, weights_only=False)['state_dict']
            mobilenet.load_state_dict({k[len('module.'):] if k.startswith('module.') else k: v for k, v in state.items()})
        else:
            mobilenet.apply(weights_init)
        self.mobilenet = nn.Sequential(*(mobilenet.model[i] for i in range(14)))
        self.decoder = choose_decoder(decoder)


class MobileNetSkipAdd(_HipForward):
    """MobileNet-v1 encoder + NNConv5 depthwise-separable decoder + 3 additive skips.

    ``MobileNetSkipAdd(output_size, pretrained=True)`` as in reference models.py:655.  Extra,
    keyword-only and optional: ``channels=(enc14, dec6)`` builds a pruned plan (e.g.
    ``PRUNED_CHANNELS``).  ``pretrained=True`` expects the ImageNet encoder checkpoint at the
    reference's relative path (models.py:660-670); it is not shipped, so callers in this repo use
    ``pretrained=False``.

    forward(x[B,3,H,W] float32 NCHW, H and W multiples of 32) -> y[B,1,H,W] on the same device.
    """

    def __init__(self, output_size, pretrained=True, *, channels=None):
        super().__init__()
        self.output_size = output_size
        enc, dec = (None, DEFAULT_DECODER) if channels is None else channels
        mobilenet = _mobilenet.MobileNet(channels=enc)
        if pretrained:
            import os
            path = os.path.join('imagenet', 'results', 'imagenet.arch=mobilenet.lr=0.1.bs=256', 'model_best.pth.tar')
            state = torch.load(path, weights_only=False)['state_dict']
            # checkpoints written through nn.DataParallel carry a `module.` prefix (models.py:667-669)
            mobilenet.load_state_dict({k[len('module.'):] if k.startswith('module.') else k: v
                                       for k, v in state.items()})
        else:
            mobilenet.apply(weights_init)
        for i in range(14):
            setattr(self, 'conv{}'.format(i), mobilenet.model[i])

        width = mobilenet.fc.in_features
        for j, out in enumerate(dec[:5], start=1):
            setattr(self, 'decode_conv{}'.format(j), nn.Sequential(depthwise(width, 5), pointwise(width, out)))
            width = out
        self.decode_conv6 = pointwise(width, dec[5])
        # NB: the reference calls weights_init(self.decode_convN) directly on the Sequential
        # (models.py:699-704), which matches none of the isinstance tests -> the decoder keeps
        # torch's default initialisation.  Nothing to do here; stated so nobody "fixes" it.


class MobileNetSkipConcat(_HipForward):
    """MobileNet-v1 encoder + NNConv5 depthwise-separable decoder whose three skips are CONCATENATED along the channel axis
    (reference models.py:734-814).  Same attribute names as MobileNetSkipAdd; decode_conv3/4/5 consume
    cat(up(x), skip) and are therefore wider (512 / 256 / 128 channels).  The depthwise kernels read the two channel ranges
    from their two tensors (`fd_layer_desc.concat`), forward and backward: nothing is concatenated in memory.
    Optional, keyword-only: ``channels=(enc14, dec6)`` as in MobileNetSkipAdd (dec = the six decoder OUTPUT widths)."""

    _fd_skip = "concat"                      # class attribute: survives unpickling of reference-format checkpoints

    def __init__(self, output_size, pretrained=True, *, channels=None):
        super().__init__()
        self.output_size = output_size
        enc, dec = (None, DEFAULT_DECODER) if channels is None else channels
        mobilenet = _mobilenet.MobileNet(channels=enc)
        if pretrained:
            import os
            path = os.path.join('imagenet', 'results', 'imagenet.arch=mobilenet.lr=0.1.bs=256', 'model_best.pth.tar')
            state = torch.load(path, weights_only=False)['state_dict']
            mobilenet.load_state_dict({k[len('module.'):] if k.startswith('module.') else k: v for k, v in state.items()})
        else:
            mobilenet.apply(weights_init)
        for i in range(14):
            setattr(self, 'conv{}'.format(i), mobilenet.model[i])
        widths = [mobilenet.model[i][-3].out_channels for i in range(14)]           # encoder block outputs (skips: blocks 5, 3, 1)
        width = widths[13]
        skip_of = {3: widths[5], 4: widths[3], 5: widths[1]}                          # stage j consumes cat(up(stage j-1), skip)
        for j, out in enumerate(dec[:5], start=1):
            cin = width + skip_of.get(j, 0)
            setattr(self, 'decode_conv{}'.format(j), nn.Sequential(depthwise(cin, 5), pointwise(cin, out)))
            width = out
        self.decode_conv6 = pointwise(width, dec[5])
        # as in MobileNetSkipAdd the reference's weights_init(self.decode_convN) calls are no-ops on Sequentials
