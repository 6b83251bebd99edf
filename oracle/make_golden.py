"""Generates tests/golden/* by running the REFERENCE ITSELF (imported from /root/reference) in this
container.  Test infrastructure; run once, commit the outputs.  Usage: python -m oracle.make_golden

Recipe ("calibrated-synthetic weights"): torch.manual_seed(s); construct the reference
module with pretrained=False; 20 train-mode no-grad forwards on cat([sample, rand(3,3,224,224)]) to
populate BatchNorm running statistics; eval(); head BN beta=2.8, gamma=1.0 so that the output is
depth-like and non-degenerate.  Variants exercise ReLU6 saturation and non-trivial running stats.
The new repo's constructor reproduces the reference's parameters bit-for-bit from the same seed
(tests/test_module_surface.py), so only seeds, BN statistics and reference OUTPUTS are stored.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

from oracle.inputs import batch_variants

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"


def import_reference():
    tv = types.ModuleType("torchvision"); tvm = types.ModuleType("torchvision.models"); tv.models = tvm
    sys.modules["torchvision"] = tv; sys.modules["torchvision.models"] = tvm     # only ResNet classes use it
    sys.path.insert(0, REF)
    import models as ref_models, metrics as ref_metrics
    sys.path.remove(REF)
    return ref_models, ref_metrics


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:16]


def calibrated(ref_models, seed, sample, variant="base"):
    torch.manual_seed(seed)
    m = ref_models.MobileNetSkipAdd((224, 224), pretrained=False)
    if variant == "sat6":          # push encoder pre-activations past the ReLU6 clamp
        for i in range(14):
            for mod in getattr(m, "conv%d" % i):
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.weight.data.mul_(4.0)
    if variant == "affine":        # non-trivial gamma/beta everywhere
        g = torch.Generator().manual_seed(seed + 1000)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.data.copy_(0.5 + torch.rand(mod.weight.shape, generator=g))
                mod.bias.data.copy_(0.4 * torch.randn(mod.bias.shape, generator=g))
    m.train()
    g = torch.Generator().manual_seed(seed + 7)
    with torch.no_grad():
        for _ in range(20):
            m(torch.cat([sample, torch.rand(3, 3, 224, 224, generator=g)]))
    m.eval()
    m.decode_conv6[1].bias.data.fill_(2.8)
    m.decode_conv6[1].weight.data.fill_(1.0)
    return m


def main():
    ref_models, ref_metrics = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    rgb = np.load(os.path.join(REF, "deploy/data/rgb.npy"))
    depth = np.load(os.path.join(REF, "deploy/data/depth.npy"))
    pred = np.load(os.path.join(REF, "deploy/data/pred.npy"))
    rgb_u8 = np.round(rgb * 255).astype(np.uint8)
    assert np.array_equal(rgb_u8.astype(np.float64) / 255.0, rgb), "sample is not exactly uint8/255"
    np.save(os.path.join(GOLD, "sample_rgb_u8.npy"), rgb_u8)           # exact: rgb == u8/255.0
    np.save(os.path.join(GOLD, "sample_depth.npy"), depth)
    np.save(os.path.join(GOLD, "sample_tvm_pred.npy"), pred.astype(np.float32))
    sample = torch.from_numpy(rgb).permute(2, 0, 1)[None].float()
    meta = {"torch": torch.__version__, "cases": {}}

    # metrics known-answer (reference metrics.py on the reference's own triple)
    r = ref_metrics.Result(); r.evaluate(torch.from_numpy(pred), torch.from_numpy(depth)[None, None])
    meta["metrics_kat"] = {k: getattr(r, k) for k in ("mse", "rmse", "mae", "lg10", "absrel", "delta1", "delta2", "delta3", "irmse", "imae")}

    for name, seed, variant, nb in (("base_s0", 0, "base", 4), ("sat6_s1", 1, "sat6", 2), ("affine_s2", 2, "affine", 2)):
        m = calibrated(ref_models, seed, sample, variant)
        sd = m.state_dict()
        x = batch_variants(sample, nb, seed)
        taps = []
        hooks = []
        from oracle.oracle import unit_names
        mods = dict(m.named_modules())
        for cp, bp, kind, stride, act in unit_names():
            # activation module follows the BN: index +1 within the same Sequential
            parent, idx = bp.rsplit(".", 1)
            actmod = mods[parent][int(idx) + 1]
            hooks.append(actmod.register_forward_hook(lambda mod, i, o: taps.append(o.detach().clone())))
        with torch.no_grad():
            y = m(x)
        for h in hooks:
            h.remove()
        assert len(taps) == 38
        bn = {k: v.numpy() for k, v in sd.items() if ("running_" in k) or (variant != "base" and (k.endswith(".1.weight") or k.endswith(".1.bias") or k.endswith(".4.weight") or k.endswith(".4.bias")))}
        # always store every BN tensor (affine + running): small (4 x 27,842 floats)
        bn = {k: v.numpy() for k, v in sd.items() if v.dim() == 1}
        np.savez_compressed(os.path.join(GOLD, "%s_bn.npz" % name), **bn)
        np.save(os.path.join(GOLD, "%s_out.npy" % name), y.numpy())
        tapstats = []
        for t in taps:
            flat = t.flatten()
            idx = torch.linspace(0, flat.numel() - 1, 64).long()
            tapstats.append({"shape": list(t.shape), "mean": float(t.double().mean()), "absmax": float(t.abs().max()),
                             "zero_frac": float((t == 0).double().mean()), "sat6_frac": float((t == 6).double().mean()),
                             "samples": [float(v) for v in flat[idx]]})
        res = ref_metrics.Result(); res.evaluate(y[:1], torch.from_numpy(depth)[None, None])
        meta["cases"][name] = {
            "seed": seed, "variant": variant, "batch": nb,
            "conv_weight_sha": {k: sha(v) for k, v in sd.items() if v.dim() == 4},
            "out_range": [float(y.min()), float(y.max())],
            "metrics_vs_sample_depth": {k: getattr(res, k) for k in ("rmse", "mae", "absrel", "lg10", "delta1", "delta2", "delta3")},
            "taps": tapstats,
        }
        print(name, "out", tuple(y.shape), meta["cases"][name]["out_range"], "rmse", res.rmse, "d1", res.delta1,
              "sat6 max frac", max(t["sat6_frac"] for t in tapstats))

    # train-mode golden (small): B=2 (sample + h-flip), L1 vs tiled depth, fp32 and fp64 reference runs
    torch.manual_seed(3)
    m = ref_models.MobileNetSkipAdd((224, 224), pretrained=False)
    m.decode_conv6[1].bias.data.fill_(2.8)
    x = torch.cat([sample, sample.flip(-1)]); tgt = torch.from_numpy(depth)[None, None].repeat(2, 1, 1, 1); tgt[1] = tgt[1].flip(-1)
    tr = {}
    for dt in (torch.float32, torch.float64):
        mm = ref_models.MobileNetSkipAdd((224, 224), pretrained=False)
        mm.load_state_dict(m.state_dict()); mm = mm.to(dt).train()
        out = mm(x.to(dt)); loss = torch.nn.L1Loss()(out, tgt.to(dt)); loss.backward()
        tr[dt] = (float(loss), {k: p.grad.detach().double() for k, p in mm.named_parameters()},
                  {k: v.detach().double() for k, v in mm.state_dict().items() if "running_" in k})
    g32, g64 = tr[torch.float32][1], tr[torch.float64][1]
    meta["train_s3"] = {
        "seed": 3, "head_beta": 2.8, "loss_fp32": tr[torch.float32][0], "loss_fp64": tr[torch.float64][0],
        "grad_norm_fp64": {k: float(v.norm()) for k, v in g64.items()},
        "grad_relerr_ref_fp32_vs_fp64": {k: float((g32[k] - v).norm() / (v.norm() + 1e-300)) for k, v in g64.items()},
    }
    small = {k: v.float().numpy() for k, v in g64.items() if v.numel() <= 4096}      # all BN grads + small convs
    np.savez_compressed(os.path.join(GOLD, "train_s3_grads_fp64_small.npz"), **small)
    np.savez_compressed(os.path.join(GOLD, "train_s3_running_fp64.npz"), **{k: v.float().numpy() for k, v in tr[torch.float64][2].items()})
    print("train loss fp32/fp64", tr[torch.float32][0], tr[torch.float64][0])
    with open(os.path.join(GOLD, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
