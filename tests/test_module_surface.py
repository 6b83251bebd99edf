"""Drop-in surface of `models.MobileNetSkipAdd` (reference models.py:654-732, main.py:49-57): attribute tree,
state_dict key scheme, pickle path, constructor RNG compatibility, plan walk."""
import io
import os
import sys
import types

import pytest
import torch

from oracle import inputs

models = inputs.product_models()


def test_state_dict_key_scheme():
    m = models.MobileNetSkipAdd((224, 224), pretrained=False)
    sd = m.state_dict()
    assert len(sd) == 228
    assert sd["conv0.0.weight"].shape == (32, 3, 3, 3)
    assert sd["conv7.0.weight"].shape == (512, 1, 3, 3) and sd["conv7.3.weight"].shape == (512, 512, 1, 1)
    assert sd["decode_conv1.0.0.weight"].shape == (1024, 1, 5, 5) and sd["decode_conv1.1.0.weight"].shape == (512, 1024, 1, 1)
    assert sd["decode_conv6.0.weight"].shape == (1, 32, 1, 1) and "decode_conv6.1.num_batches_tracked" in sd
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 3960930
    assert m.output_size == (224, 224)


def test_pruned_plan_builds_and_counts():
    m = models.MobileNetSkipAdd((224, 224), pretrained=False, channels=models.PRUNED_CHANNELS)
    assert sum(p.numel() for p in m.parameters()) == 1360786


def test_decoder_keeps_default_init_and_encoder_is_he_normal():
    torch.manual_seed(0)
    m = models.MobileNetSkipAdd((224, 224), pretrained=False)
    assert abs(float(m.decode_conv1[0][0].weight.std()) - 0.1155) < 0.01       # kaiming-uniform default (SURVEY trap 2)
    assert abs(float(m.conv13[3].weight.std()) - (2.0 / 1024) ** 0.5) < 0.002  # weights_init normal(0, sqrt(2/(k*k*cout)))


def test_pickle_roundtrip_under_module_name_models():
    m = models.MobileNetSkipAdd((224, 224), pretrained=False)
    buf = io.BytesIO()
    torch.save({"epoch": 3, "model": m}, buf)                   # reference checkpoint format (main.py:50-54)
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    m2 = ck["model"]
    assert type(m2).__module__ == "models" and type(m2).__name__ == "MobileNetSkipAdd"
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert "_fd_engine" not in m2.__dict__


def test_forward_refuses_cpu_tensors():
    m = models.MobileNetSkipAdd((224, 224), pretrained=False).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 3, 224, 224))


def test_plan_walk_matches_reference_forward_order():
    from fastdepth_hip import capi
    from fastdepth_hip.plan import layers_of
    ls = layers_of(models.MobileNetSkipAdd((224, 224), pretrained=False))
    assert len(ls) == 38
    d = [l.desc for l in ls]
    assert (d[0].op, d[0].cin, d[0].cout, d[0].stride, d[0].act) == (capi.FD_OP_STEM, 3, 32, 2, capi.FD_ACT_RELU6)
    assert [x.stride for x in d[1:27:2]] == [1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1]     # mobilenet.py:42-54
    assert all(x.act == capi.FD_ACT_RELU6 for x in d[:27]) and all(x.act == capi.FD_ACT_RELU for x in d[27:])
    # decoder: up/skip belong to the consumer (models.py:720-729): decode_conv2 up only, 3 += x3(conv5), 4 += x2(conv3), 5 += x1(conv1)
    assert [(x.upsample, x.skip) for x in d[27:38:2]] == [(0, -1), (1, -1), (1, 10), (1, 6), (1, 2), (1, -1)]
    assert all(x.upsample == 0 and x.skip == -1 for x in d[28:37:2])
    assert [x.src for x in d] == [-1] + list(range(37))
    assert (d[37].op, d[37].cin, d[37].cout) == (capi.FD_OP_PW, 32, 1)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present (GPU box)")
def test_seeded_constructor_is_bit_identical_to_reference():
    sys.dont_write_bytecode = True
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "models" or k.startswith("imagenet")}
    tv = types.ModuleType("torchvision"); tv.models = types.ModuleType("torchvision.models")
    sys.modules.setdefault("torchvision", tv); sys.modules.setdefault("torchvision.models", tv.models)
    sys.path.insert(0, "/root/reference")
    try:
        import models as ref_models
        torch.manual_seed(7); r = ref_models.MobileNetSkipAdd((224, 224), pretrained=False)
    finally:
        sys.path.remove("/root/reference")
        for k in [k for k in sys.modules if k == "models" or k.startswith("imagenet")]:
            del sys.modules[k]
        sys.modules.update(saved)
    torch.manual_seed(7); m = models.MobileNetSkipAdd((224, 224), pretrained=False)
    a, b = r.state_dict(), m.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_no_skip_sibling_matches_reference_surface():
    """`MobileNet('nnconv5dw', ...)` (reference models.py:420-460) -- same state_dict keys, and for the same
    seed bit-identical conv weights (sha of every conv weight of the reference is stored in tests/golden/siblings.json; the
    reference's choose_decoder applies weights_init to the decoder, unlike MobileNetSkipAdd)."""
    import pytest
    from oracle import inputs
    m, x, y, meta = inputs.golden_sibling_case("nnconv5dw_s4")
    keys = list(m.state_dict())
    assert len(keys) == 228 and keys[0] == "mobilenet.0.0.weight" and "decoder.conv6.1.running_var" in keys
    assert [n for n, _ in m.named_children()] == ["mobilenet", "decoder"]
    models = inputs.product_models()
    with pytest.raises(NotImplementedError):
        models.MobileNet("deconv3", (224, 224), pretrained=False)        # decoders outside the accelerated path say so
    with pytest.raises(RuntimeError):
        m(x)                                                              # CPU tensor: no fallback


def test_skip_concat_sibling_matches_reference_surface():
    """Row f-3, second sibling: `MobileNetSkipConcat` (reference models.py:734-814) -- same keys, bit-identical seeded weights."""
    from oracle import inputs
    from fastdepth_hip.plan import layers_of
    m, x, y, meta = inputs.golden_sibling_case("skipconcat_s6")
    assert m.decode_conv3[0][0].in_channels == 512 and m.decode_conv5[1][0].in_channels == 128
    descs = [l.desc for l in layers_of(m)]
    cat = [(d.cin, d.concat, d.skip) for d in descs if d.concat]
    assert [c[0] for c in cat] == [512, 256, 128] and all(c[2] >= 0 for c in cat)
