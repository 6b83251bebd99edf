"""Deterministic NYU-shaped inputs and golden-case reconstruction shared by make_golden.py and tests/.
TEST INFRASTRUCTURE ONLY."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
PKG = os.path.join(REPO, "fast-depth_amd")


def load_sample():
    """The reference's one shipped NYU-v2 validation sample (deploy/data/rgb.npy, depth.npy), stored as
    uint8 (exact: rgb == u8/255.0) + float32 depth.  Returns (x[1,3,224,224] f32, depth[1,1,224,224] f32)."""
    rgb = np.load(os.path.join(GOLD, "sample_rgb_u8.npy")).astype(np.float64) / 255.0
    depth = np.load(os.path.join(GOLD, "sample_depth.npy"))
    return torch.from_numpy(rgb).permute(2, 0, 1)[None].float(), torch.from_numpy(depth)[None, None]


def batch_variants(sample, n, seed=0):
    """sample + deterministic variants (h-flip, integer shifts, per-channel gain), SURVEY.md 8(d) config 2."""
    g = torch.Generator().manual_seed(seed)
    out = [sample[0]]
    for i in range(1, n):
        v = sample[0]
        if i % 2 == 1:
            v = v.flip(-1)
        v = torch.roll(v, shifts=(int(torch.randint(-16, 17, (1,), generator=g)),
                                  int(torch.randint(-16, 17, (1,), generator=g))), dims=(1, 2))
        gain = 0.8 + 0.4 * torch.rand(3, 1, 1, generator=g)
        out.append((v * gain).clamp(0, 1))
    return torch.stack(out)


def golden_meta():
    with open(os.path.join(GOLD, "golden.json")) as f:
        return json.load(f)


def product_models():
    """Imports the product's drop-in `models` module (fast-depth_amd/ is the import root, like the
    reference's repo root)."""
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    import models
    return models


def _sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:16]


def golden_case(name):
    """Rebuilds a golden case WITHOUT the reference: seed -> product constructor (bit-identical parameters,
    verified against the stored sha of every conv weight) + stored BatchNorm tensors.
    Returns (module in eval mode, x, reference_output, case_meta)."""
    meta = golden_meta()["cases"][name]
    models = product_models()
    torch.manual_seed(meta["seed"])
    m = models.MobileNetSkipAdd((224, 224), pretrained=False)
    sd = m.state_dict()
    for k, h in meta["conv_weight_sha"].items():
        if _sha(sd[k]) != h:
            raise AssertionError("seeded constructor no longer reproduces reference weights: " + k)
    bn = np.load(os.path.join(GOLD, name + "_bn.npz"))
    m.load_state_dict({k: torch.from_numpy(bn[k]) for k in bn.files}, strict=False)
    m.eval()
    x = batch_variants(load_sample()[0], meta["batch"], meta["seed"])
    y = torch.from_numpy(np.load(os.path.join(GOLD, name + "_out.npy")))
    return m, x, y, meta


def golden_sibling_case(name):
    """Same as golden_case for the no-skip sibling `MobileNet(decoder, ...)` (tests/golden/siblings.json,
    oracle/make_golden_siblings.py)."""
    with open(os.path.join(GOLD, "siblings.json")) as f:
        meta = json.load(f)[name]
    models = product_models()
    torch.manual_seed(meta["seed"])
    m = models.MobileNetSkipConcat((224, 224), pretrained=False) if meta["decoder"] == "skipconcat" else models.MobileNet(meta["decoder"], (224, 224), pretrained=False)
    sd = m.state_dict()
    if len(sd) != meta["keys"]:
        raise AssertionError("state_dict has %d keys, the reference has %d" % (len(sd), meta["keys"]))
    for k, h in meta["conv_weight_sha"].items():
        if _sha(sd[k]) != h:
            raise AssertionError("seeded constructor no longer reproduces reference weights: " + k)
    bn = np.load(os.path.join(GOLD, name + "_bn.npz"))
    m.load_state_dict({k: torch.from_numpy(bn[k]) for k in bn.files}, strict=False)
    m.eval()
    x = batch_variants(load_sample()[0], meta["batch"], meta["seed"])
    y = torch.from_numpy(np.load(os.path.join(GOLD, name + "_out.npy")))
    return m, x, y, meta
