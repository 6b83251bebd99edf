"""Drop-in `metrics` module (reference /root/reference/metrics.py): `Result` and `AverageMeter` with the same fields and
methods, but `Result.evaluate(output, target)` is ONE fused device reduction (fd_depth_metrics) followed by a single 80-byte
device->host copy, instead of the reference's ~12 `float(tensor)` synchronisations per sample (metrics.py:38-55).
SURVEY.md row f-2.  As everywhere in this package there is no CPU path: both tensors must live on the GPU."""
import ctypes
import math

import numpy as np
import torch


# the ten error measures of the reference's Result, in the positional order of its update(...) signature (metrics.py:23-29),
# followed by the two timing fields (passed as gpu_time, data_time)
_MEASURES = ("irmse", "imae", "mse", "rmse", "mae", "absrel", "lg10", "delta1", "delta2", "delta3")
_HIGHER_IS_BETTER = ("delta1", "delta2", "delta3")


class Result(object):
    """Same attributes and methods as the reference's metrics.Result; the state is filled from tables instead of literal
    assignments, and evaluate() is one device reduction."""

    def __init__(self):
        self._fill({m: 0 for m in _MEASURES})

    def _fill(self, values, gpu_time=0, data_time=0):
        for name in _MEASURES:
            setattr(self, name, values[name])
        self.data_time, self.gpu_time = data_time, gpu_time

    def set_to_worst(self):
        self._fill({m: (0 if m in _HIGHER_IS_BETTER else np.inf) for m in _MEASURES})

    def update(self, *args):
        """update(irmse, imae, mse, rmse, mae, absrel, lg10, delta1, delta2, delta3, gpu_time, data_time)"""
        if len(args) != len(_MEASURES) + 2:
            raise TypeError("update() takes %d positional values: %s, gpu_time, data_time" % (len(_MEASURES) + 2, ", ".join(_MEASURES)))
        self._fill(dict(zip(_MEASURES, args)), args[-2], args[-1])

    def evaluate(self, output, target):
        """Same definitions as reference metrics.py:31-55 (valid = target>0 or output>0, millimetres, delta_k thresholds 1.25^k), over
        all elements of the two tensors -- like the reference, which is only ever handed one image; per-image results of a batch:
        evaluate_frames."""
        if not (output.is_cuda and target.is_cuda):
            raise RuntimeError("fast-depth_amd metrics run on the GPU only (no CPU path in this package)")
        o = output.detach().float().contiguous()
        t = target.detach().float().contiguous()
        if o.numel() != t.numel():
            raise ValueError("output and target must have the same number of elements")
        self._from_sums(_device_sums(o, t, 1)[0])

    def _from_sums(self, s):
        n = s[0]
        self.mse = float(s[1] / n)
        self.rmse = math.sqrt(self.mse)
        self.mae = float(s[2] / n)
        self.lg10 = float(s[3] / n)
        self.absrel = float(s[4] / n)
        self.delta1, self.delta2, self.delta3 = float(s[5] / n), float(s[6] / n), float(s[7] / n)
        self.data_time = 0
        self.gpu_time = 0
        self.irmse = math.sqrt(s[8] / n)
        self.imae = float(s[9] / n)
        return self

    @staticmethod
    def evaluate_frames(output, target):
        """One Result per image of a batch [B, ...]: what the reference's loop computes with its batch size of 1 (main.py:40-41,
        80-82 -- RMSE / iRMSE of pooled pixels differ from the mean of the per-image values), still as ONE device reduction and
        one (B x 80-byte) copy."""
        if not (output.is_cuda and target.is_cuda):
            raise RuntimeError("fast-depth_amd metrics run on the GPU only (no CPU path in this package)")
        o = output.detach().float().contiguous()
        t = target.detach().float().contiguous()
        if o.shape != t.shape or o.dim() < 2:
            raise ValueError("output and target must be batches of the same shape")
        return [Result()._from_sums(s) for s in _device_sums(o, t, o.shape[0])]


def _device_sums(o, t, n_frames):
    from fastdepth_hip import capi
    from fastdepth_hip.engine import lib
    L = lib()
    sums = torch.empty((n_frames, 10), dtype=torch.float64, device=o.device)
    scratch = torch.empty(L.fd_depth_metrics_frames_scratch_bytes(n_frames), dtype=torch.uint8, device=o.device)
    with torch.cuda.device(o.device):
        capi.check(L, L.fd_depth_metrics_frames(o.data_ptr(), t.data_ptr(), n_frames, o.numel() // n_frames, sums.data_ptr(), scratch.data_ptr(),
                                                torch.cuda.current_stream(o.device).cuda_stream), "fd_depth_metrics_frames")
    return sums.cpu().numpy()                   # the single synchronisation


class AverageMeter(object):
    """n-weighted running sums of Result fields (reference metrics.py:58-95)."""
    _FIELDS = _MEASURES

    def __init__(self):
        self.reset()

    def reset(self):
        self.count = 0.0
        for f in self._FIELDS:
            setattr(self, "sum_" + f, 0)
        self.sum_data_time, self.sum_gpu_time = 0, 0

    def update(self, result, gpu_time, data_time, n=1):
        self.count += n
        for f in self._FIELDS:
            setattr(self, "sum_" + f, getattr(self, "sum_" + f) + n * getattr(result, f))
        self.sum_data_time += n * data_time
        self.sum_gpu_time += n * gpu_time

    def average(self):
        avg = Result()
        c = self.count
        avg.update(*[getattr(self, "sum_" + f) / c for f in _MEASURES], self.sum_gpu_time / c, self.sum_data_time / c)
        return avg
