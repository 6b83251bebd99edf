"""numpy restatement of the reference's depth metrics (parity yard-stick).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/metrics.py:31-55 (Result.evaluate) line by line; pinned by the known-answer
test on the reference's own sample triple (tests/golden/sample_*.npy)."""
import math

import numpy as np

FIELDS = ("mse", "rmse", "mae", "lg10", "absrel", "delta1", "delta2", "delta3", "irmse", "imae")


def evaluate(output, target):
    output = np.asarray(output, np.float32).reshape(-1)
    target = np.asarray(target, np.float32).reshape(-1)
    valid = (target > 0) | (output > 0)                       # metrics.py:32
    o = np.float32(1e3) * output[valid]                       # :34-35 (millimetres)
    t = np.float32(1e3) * target[valid]
    ad = np.abs(o - t)
    r = {}
    r["mse"] = float(np.mean(ad ** 2, dtype=np.float32))      # :38
    r["rmse"] = math.sqrt(r["mse"])
    r["mae"] = float(np.mean(ad, dtype=np.float32))
    with np.errstate(divide="ignore", invalid="ignore"):
        r["lg10"] = float(np.mean(np.abs(np.log(o) / np.float32(math.log(10)) - np.log(t) / np.float32(math.log(10))), dtype=np.float32))
        r["absrel"] = float(np.mean(ad / t, dtype=np.float32))
        ratio = np.maximum(o / t, t / o)                       # :44
        for k in (1, 2, 3):
            r["delta%d" % k] = float(np.mean((ratio < 1.25 ** k).astype(np.float32), dtype=np.float32))
        inv = np.abs(1 / o - 1 / t)                            # :51-55
        r["irmse"] = math.sqrt(float(np.mean(inv ** 2, dtype=np.float32)))
        r["imae"] = float(np.mean(inv, dtype=np.float32))
    return r
