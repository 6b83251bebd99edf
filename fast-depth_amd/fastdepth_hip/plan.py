"""nn.Module tree -> list of fused layers for fd_plan_create.

Shapes are discovered from the instance's sub-modules at call time, never from constructor arguments:
reference checkpoints are whole pickled modules whose __init__ is bypassed on load (main.py:49-57),
and pruned models carry irregular widths.  The walk restates the reference's forward order
(models.py:706-732): conv0..conv13, skip taps after conv1/conv3/conv5, decode_conv1..5 each followed
by nearest x2 and (after 2/3/4) the additive skip, then decode_conv6.  Upsample + add are not layers of
their own: they become `upsample` / `skip` attributes of the layer that consumes the result.
"""
import torch.nn as nn

from . import capi


def _act_of(mod):
    if isinstance(mod, nn.ReLU6):
        return capi.FD_ACT_RELU6
    if isinstance(mod, nn.ReLU):
        return capi.FD_ACT_RELU
    if isinstance(mod, nn.Hardtanh) and mod.min_val == 0 and mod.max_val == 6:
        return capi.FD_ACT_RELU6
    raise capi.FastDepthError("unsupported activation %r" % (mod,))


def _units(seq):
    """Splits a Sequential (possibly nested) into (conv, bn, act) triples in execution order."""
    flat = []

    def walk(m):
        if isinstance(m, nn.Sequential):
            for c in m:
                walk(c)
        else:
            flat.append(m)
    walk(seq)
    if len(flat) % 3:
        raise capi.FastDepthError("expected Conv-BN-act triples, got %d modules" % len(flat))
    out = []
    for i in range(0, len(flat), 3):
        conv, bn, act = flat[i:i + 3]
        if not (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d)):
            raise capi.FastDepthError("expected Conv2d, BatchNorm2d, activation; got %r, %r" % (conv, bn))
        if conv.bias is not None or conv.dilation != (1, 1) or conv.padding != (conv.kernel_size[0] // 2,) * 2:
            raise capi.FastDepthError("conv %r is outside the FastDepth path (bias/dilation/padding)" % (conv,))
        out.append((conv, bn, _act_of(act)))
    return out


class Layer:
    __slots__ = ("conv", "bn", "desc", "name")

    def __init__(self, name, conv, bn, act, src, upsample=0, skip=-1, concat=0):
        k = conv.kernel_size[0]
        if conv.groups == 1 and k == 3:
            op = capi.FD_OP_STEM
        elif conv.groups == conv.in_channels == conv.out_channels and k in (3, 5):
            op = capi.FD_OP_DW
        elif conv.groups == 1 and k == 1:
            op = capi.FD_OP_PW
        else:
            raise capi.FastDepthError("%s: conv %r has no fused kernel on this path" % (name, conv))
        self.name, self.conv, self.bn = name, conv, bn
        self.desc = capi.LayerDesc(op, conv.in_channels, conv.out_channels, k, conv.stride[0], act, src,
                                   upsample, skip, concat)


def _layers_of_plain(model):
    """MobileNet(decoder='nnconv*dw')-shaped module (reference models.py:420-460 + NNConv.forward :244-270): encoder
    `mobilenet.0..13`, decoder `decoder.conv1..6` with a nearest x2 after conv1..conv5 and no skips."""
    layers, src = [], -1
    for i in range(14):
        for j, (conv, bn, act) in enumerate(_units(model.mobilenet[i])):
            layers.append(Layer("mobilenet.%d.%d" % (i, 3 * j), conv, bn, act, src))
            src = len(layers) - 1
    pending_up = 0
    for j in range(1, 7):
        for q, (conv, bn, act) in enumerate(_units(getattr(model.decoder, "conv%d" % j))):
            layers.append(Layer("decoder.conv%d.%d" % (j, q), conv, bn, act, src, pending_up, -1))
            pending_up = 0
            src = len(layers) - 1
        pending_up = 1 if j <= 5 else 0
    return layers


def layers_of(model):
    """MobileNetSkipAdd-shaped module -> [Layer].  Skip sources follow models.py:714-719, 724-729.
    A module with `.mobilenet` / `.decoder` (the no-skip sibling) takes the plain walk above."""
    if hasattr(model, "mobilenet") and hasattr(model, "decoder"):
        return _layers_of_plain(model)
    layers, skips = [], {}
    src = -1
    for i in range(14):
        for j, (conv, bn, act) in enumerate(_units(getattr(model, "conv%d" % i))):
            layers.append(Layer("conv%d.%d" % (i, 3 * j), conv, bn, act, src))
            src = len(layers) - 1
        if i in (1, 3, 5):
            skips[i] = src
    skip_after = {2: 5, 3: 3, 4: 1}        # decode stage -> encoder block whose output is added after its upsample
    concat = 1 if getattr(type(model), "_fd_skip", "add") == "concat" else 0      # MobileNetSkipConcat: torch.cat instead of + (models.py:803-808)
    pending_up, pending_skip = 0, -1
    for j in range(1, 7):
        for q, (conv, bn, act) in enumerate(_units(getattr(model, "decode_conv%d" % j))):
            layers.append(Layer("decode_conv%d.%d" % (j, q), conv, bn, act, src, pending_up, pending_skip, concat if pending_skip >= 0 else 0))
            pending_up, pending_skip = 0, -1
            src = len(layers) - 1
        if j <= 5:
            pending_up = 1
            pending_skip = skips[skip_after[j]] if j in skip_after else -1
    return layers
