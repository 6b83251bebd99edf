"""ctypes binding of fd_oracle.c plus the state_dict -> unit-list walk.  TEST INFRASTRUCTURE ONLY.

The unit order and the stride table restate the reference's network definition:
  encoder  /root/reference/imagenet/mobilenet.py:40-54 (strides), :22-38 (unit composition, ReLU6)
  decoder  /root/reference/models.py:683-698 (dw5x5 + pw, ReLU), forward order models.py:706-732
"""
import ctypes
import os

import numpy as np

from . import build as _build

ACT_RELU, ACT_RELU6 = 1, 2
# stride of the depthwise conv of conv1..conv13 (reference imagenet/mobilenet.py:42-54)
ENCODER_DW_STRIDES = (1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1)
BN_EPS, BN_MOMENTUM = 1e-5, 0.1   # nn.BatchNorm2d defaults, as instantiated by the reference


class _Unit(ctypes.Structure):
    _fields_ = [("cin", ctypes.c_int32), ("cout", ctypes.c_int32), ("ksize", ctypes.c_int32),
                ("stride", ctypes.c_int32), ("groups", ctypes.c_int32), ("act", ctypes.c_int32),
                ("w", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("mean", ctypes.c_void_p), ("var", ctypes.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        _lib.fdo_skipadd_forward.restype = ctypes.c_int
        _lib.fdo_skipadd_forward.argtypes = [ctypes.POINTER(_Unit), ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_float,
                                             ctypes.c_float]
        _lib.fdo_free.argtypes = [ctypes.c_void_p]
    return _lib


def unit_names():
    """(conv-prefix, bn-prefix, kind, stride, act) for the 38 units in forward order (state_dict key scheme:
    SURVEY.md Appendix E)."""
    names = [("conv0.0", "conv0.1", "stem", 2, ACT_RELU6)]
    for i, s in enumerate(ENCODER_DW_STRIDES, start=1):
        names.append(("conv%d.0" % i, "conv%d.1" % i, "dw", s, ACT_RELU6))
        names.append(("conv%d.3" % i, "conv%d.4" % i, "pw", 1, ACT_RELU6))
    for j in range(1, 6):
        names.append(("decode_conv%d.0.0" % j, "decode_conv%d.0.1" % j, "dw", 1, ACT_RELU))
        names.append(("decode_conv%d.1.0" % j, "decode_conv%d.1.1" % j, "pw", 1, ACT_RELU))
    names.append(("decode_conv6.0", "decode_conv6.1", "pw", 1, ACT_RELU))
    return names


def to_numpy_state(state_dict):
    """torch state_dict (or dict of arrays) -> dict of contiguous float32 numpy arrays (BN counters dropped)."""
    out = {}
    for k, v in state_dict.items():
        if k.endswith("num_batches_tracked"):
            continue
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        out[k] = np.array(a, dtype=np.float32, copy=True, order="C")   # private copy: train mode updates running stats in place
    return out


def _units(sd):
    arr = (_Unit * 38)()
    for u, (cp, bp, kind, stride, act) in zip(arr, unit_names()):
        w = sd[cp + ".weight"]
        u.cout, u.ksize = w.shape[0], w.shape[2]
        u.groups = w.shape[0] if kind == "dw" else 1
        u.cin = w.shape[0] if kind == "dw" else w.shape[1]
        u.stride, u.act = stride, act
        u.w = w.ctypes.data
        u.gamma, u.beta = sd[bp + ".weight"].ctypes.data, sd[bp + ".bias"].ctypes.data
        u.mean, u.var = sd[bp + ".running_mean"].ctypes.data, sd[bp + ".running_var"].ctypes.data
    return arr


def forward(state_dict, x, taps=False, train=False):
    """Runs the C restatement.  x: [N,3,H,W] float32 NCHW.  Returns y [N,1,H,W] (and the 38 per-unit
    outputs, NCHW, if taps).  With train=True BatchNorm uses batch statistics and the running stats of
    the *numpy copy* are updated in place (returned as third value)."""
    sd = to_numpy_state(state_dict)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, c, h, w = x.shape
    assert c == 3 and h % 32 == 0 and w % 32 == 0
    units = _units(sd)
    y = np.empty((n, units[37].cout, h, w), np.float32)
    tap_ptrs = (ctypes.c_void_p * 38)() if taps else None
    rc = lib().fdo_skipadd_forward(units, 38, x.ctypes.data, n, h, w, y.ctypes.data, tap_ptrs,
                                   int(train), BN_EPS, BN_MOMENTUM)
    if rc != 0:
        raise RuntimeError("fd_oracle: structural error %d" % rc)
    if not taps:
        return (y, sd) if train else y
    outs, hh = [], h
    for i, u in enumerate(units):
        hh = (hh + 2 * (u.ksize // 2) - u.ksize) // u.stride + 1
        ww = hh * w // h
        shape = (n, u.cout, hh, ww)
        buf = np.ctypeslib.as_array(ctypes.cast(tap_ptrs[i], ctypes.POINTER(ctypes.c_float)), shape=(int(np.prod(shape)),))
        outs.append(buf.reshape(shape).copy())
        lib().fdo_free(tap_ptrs[i])
        if i >= 28 and i < 37 and (i - 28) % 2 == 0:
            hh *= 2      # nearest x2 after each decode_conv1..5 (models.py:723)
    return (y, outs, sd) if train else (y, outs)
