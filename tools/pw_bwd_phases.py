"""Phase table of the paired 16-bit pointwise backward launch (fd_pw_bwd_h16): 100 MHz real-time stamps taken by every workgroup of ONE unit's launch
inside a full bf16 train step at batch 32 (library variant built with -DFD_PW_PROBE: tools/build_variant.py pwprobe -DFD_PW_PROBE).  Measurement aid.
usage (GPU box): python tools/pw_bwd_phases.py --lib scratch/variants/pwprobe.so"""
import argparse, ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd")); sys.path.insert(0, REPO)
import numpy as np
import torch
import models
from fastdepth_hip import capi
ap = argparse.ArgumentParser(); ap.add_argument("--lib", required=True); ap.add_argument("--batch", type=int, default=32); a = ap.parse_args()
capi.DEFAULT_LIB = os.path.abspath(a.lib)
from fastdepth_hip.train import TrainEngine
torch.manual_seed(0)
m = models.MobileNetSkipAdd((224, 224), pretrained=False).cuda().train()
eng = TrainEngine(m, dtype=torch.bfloat16)
L = eng.L
L.fd_pw_probe_select.argtypes = [ctypes.c_int] * 3; L.fd_pw_probe_select.restype = ctypes.c_int
L.fd_pw_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.fd_pw_probe_read.restype = ctypes.c_int
x = torch.rand(a.batch, 3, 224, 224, device="cuda"); t = torch.rand(a.batch, 1, 224, 224, device="cuda") * 5 + 0.5
for _ in range(3): eng.step(x, t)
torch.cuda.synchronize()
B = a.batch
units = [("conv1.3", B * 112 * 112, 64, 32), ("conv2.3", B * 56 * 56, 128, 64), ("conv3.3", B * 56 * 56, 128, 128), ("conv5.3", B * 28 * 28, 256, 256),
         ("conv8.3 (conv7-11.3)", B * 14 * 14, 512, 512), ("conv13.3", B * 7 * 7, 1024, 1024), ("decode_conv4.1", B * 56 * 56, 64, 128), ("decode_conv5.1", B * 112 * 112, 32, 64)]
SLOTS = 16384
TICK = 0.01                                                    # us per tick (s_memrealtime: 100 MHz)
med = lambda v: float(np.median(v)) * TICK if len(v) else 0.0
print("unit (M x N x K)                      | role           | workgroups | life us (median / p90) | prologue | main loop | epilogue | tail | launch span us | resident at once (sum of lives / span)")
for name, M, N, K in units:
    assert L.fd_pw_probe_select(M, N, K) == 0
    eng.step(x, t); torch.cuda.synchronize()
    buf = np.zeros((SLOTS, 8), dtype=np.int64)
    assert L.fd_pw_probe_read(buf.ctypes.data, SLOTS) == SLOTS
    d, w = buf[buf[:, 0] == 1], buf[buf[:, 0] == 2]
    d = d[d[:, 5] > 0]; w = w[w[:, 4] > 0]
    if not len(d) and not len(w):
        print("%-38s| no stamped workgroups (the unit's launch is not the paired form?)" % name); continue
    starts = np.concatenate([d[:, 1], w[:, 1]]); ends = np.concatenate([d[:, 5], w[:, 4]])
    span = (ends.max() - starts.min()) * TICK
    lives = np.concatenate([d[:, 5] - d[:, 1], w[:, 4] - w[:, 1]])
    conc = lives.sum() * TICK / span
    if len(d):
        life = d[:, 5] - d[:, 1]
        print("%-38s| backward-data  | %10d | %6.2f / %6.2f        | %8.2f | %9.2f | %8.2f | %4.2f | %14.2f | %.0f" % (
            "%s (%d x %d x %d)" % (name, M, N, K), len(d), med(life), float(np.percentile(life, 90)) * TICK, med(d[:, 2] - d[:, 1]), med(d[:, 3] - d[:, 2]), med(d[:, 4] - d[:, 3]), med(d[:, 5] - d[:, 4]), span, conc))
    if len(w):
        life = w[:, 4] - w[:, 1]
        print("%-38s| weight-gradient| %10d | %6.2f / %6.2f        | %8.2f | %9.2f | %8.2f |      |                | loop: staging + load wait %.2f, MFMA phase %.2f; first start %.2f us after the launch's first workgroup" % (
            "", len(w), med(life), float(np.percentile(life, 90)) * TICK, med(w[:, 2] - w[:, 1]), med(w[:, 3] - w[:, 2]), med(w[:, 4] - w[:, 3]), med(w[:, 6]), med(w[:, 7]), (w[:, 1].min() - starts.min()) * TICK))
