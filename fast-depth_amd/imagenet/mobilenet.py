"""MobileNet-v1 encoder blocks, host-side mirror of the reference's `imagenet/mobilenet.py`.

Only the *module surface* lives here (plain ``torch.nn`` containers that own the parameters and
buffers); the arithmetic for the FastDepth hot path is executed by the HIP engine in
``fastdepth_hip`` and never by these modules' own ``forward``.

Reference: /root/reference/imagenet/mobilenet.py:12-63 (class MobileNet), :22-27 (conv_bn),
:29-38 (conv_dw), :40-56 (the 14-block table + AvgPool + fc).

The constructor is table driven so that pruned channel plans build the
same attribute tree.  Module creation order -- every Conv2d of every block in sequence, then the
(unused by FastDepth) 1000-way classifier -- is kept identical to the reference on purpose: the
default initialisers draw from torch's global RNG, so an identical order means that
``torch.manual_seed(s)`` followed by construction yields bit-identical parameters to the reference
(checked by tests/test_module_surface.py when /root/reference is present).
"""
import torch.nn as nn

# (out_channels, stride) of the 13 depthwise-separable units that follow the stem.
# Reference table: imagenet/mobilenet.py:41-54.
STEM_CHANNELS = 32
DW_UNITS = ((64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2),
            (512, 1), (512, 1), (512, 1), (512, 1), (512, 1), (1024, 2), (1024, 1))


def _act(relu6):
    return nn.ReLU6(inplace=True) if relu6 else nn.ReLU(inplace=True)


def stem_unit(cin, cout, stride, relu6=True):
    """3x3 dense conv + BN + activation (reference conv_bn, mobilenet.py:22-27)."""
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout), _act(relu6))


def separable_unit(cin, cout, stride, relu6=True):
    """dw3x3(stride)+BN+act, pw1x1+BN+act (reference conv_dw, mobilenet.py:29-38)."""
    return nn.Sequential(
        nn.Conv2d(cin, cin, 3, stride, 1, groups=cin, bias=False), nn.BatchNorm2d(cin), _act(relu6),
        nn.Conv2d(cin, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout), _act(relu6))


class MobileNet(nn.Module):
    """Attribute-compatible with the reference class: ``.model`` (Sequential of 14 units + AvgPool2d)
    and ``.fc``.  ``channels`` optionally overrides the 14 output widths (pruned plans)."""

    def __init__(self, relu6=True, channels=None):
        super().__init__()
        widths = [STEM_CHANNELS] + [c for c, _ in DW_UNITS] if channels is None else list(channels)
        if len(widths) != 1 + len(DW_UNITS):
            raise ValueError("channels must list 14 widths (stem + 13 separable units)")
        units = [stem_unit(3, widths[0], 2, relu6)]
        for i, (_, stride) in enumerate(DW_UNITS):
            units.append(separable_unit(widths[i], widths[i + 1], stride, relu6))
        units.append(nn.AvgPool2d(7))
        self.model = nn.Sequential(*units)
        self.fc = nn.Linear(widths[-1], 1000)

    def forward(self, x):  # ImageNet classifier path; not on the FastDepth hot path.
        x = self.model(x)
        return self.fc(x.view(-1, self.fc.in_features))
