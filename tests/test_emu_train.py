"""CPU-emulated train step kernels (train-mode forward with batch-statistics BatchNorm, hand-written backward, L1 loss,
fused SGD) through the C ABI, against the torch-functional oracle in fp64 with the noise-aware gradient criterion."""
import ctypes

import numpy as np
import pytest
import torch

import harness
from fastdepth_hip import capi
from test_emu_forward import G16, RAGGED, TINY, small_model


@pytest.mark.parametrize("name,plan,b", [("tiny", TINY, 2), ("tiny_sat6", TINY, 2)])     # (the ragged widths run through the layer-local test below)
def test_emulated_train_forward_backward(name, plan, b):
    m = small_model(plan[0], plan[1], seed=3)
    if name.endswith("sat6"):
        harness.saturate_encoder(m)              # encoder gamma x 4: the clamp-at-6 side of the ReLU6 masks
    g = torch.Generator().manual_seed(9)
    x = torch.rand(b, 3, 64, 64, generator=g)
    target = 2.0 + torch.rand(b, 1, 64, 64, generator=g)
    rep = harness.train_parity_report("emu", m, x, target, torch.device("cpu"), kink=1e-3 if name.endswith("sat6") else 1e-4)
    # (sat6: the pre-activations are 4x larger and, at this size, agree with the oracle to ~4e-4 of their scale only -- see below)
    # 64x64 inputs leave 2x2 pixels x batch 2 = 8 samples per channel at the deepest BatchNorms: the train-mode forward is
    # ill-conditioned there (pre-activations agree to ~4e-4 only), and the backward inherits that -> looser bound than on the
    # full-size GPU test
    # (the saturating variant has twice as many kinks per unit: its end-to-end bound is looser still; the sharp statement about the
    # clamp-at-6 mask is the layer-local test below, at 2e-5)
    harness.assert_train_parity(rep, tol=1e-2 if name.endswith("sat6") else 5e-3)
    if name.endswith("sat6"):
        assert harness.LAST_SAT6_FRAC > 0.005, harness.LAST_SAT6_FRAC


def test_emulated_l1_loss_and_sgd():
    L = harness.get_lib("emu")
    g = torch.Generator().manual_seed(1)
    pred, tgt = torch.rand(2, 1, 32, 32, generator=g), torch.rand(2, 1, 32, 32, generator=g)
    pred[0, 0, 0, :4] = tgt[0, 0, 0, :4]                       # exact ties: gradient 0 (torch sgn semantics)
    dpred, loss = torch.empty_like(pred), torch.zeros(1)
    scratch = torch.empty(L.fd_l1_loss_scratch_bytes(pred.numel()), dtype=torch.uint8)
    capi.check(L, L.fd_l1_loss(pred.data_ptr(), tgt.data_ptr(), dpred.data_ptr(), loss.data_ptr(), pred.numel(), scratch.data_ptr(), None), "fd_l1_loss")
    p = pred.clone().requires_grad_(True)
    ref = torch.nn.L1Loss()(p, tgt); ref.backward()
    assert float(loss) == pytest.approx(float(ref), rel=1e-6)
    assert torch.equal(dpred, p.grad)
    # masked form (valid = target > 0): invalid rows / pixels, an exact tie on a valid pixel, then the all-invalid case
    from oracle import torch_ref
    tm = tgt.clone()
    tm[0, 0, 5:9] = 0.0; tm[1, 0, :, ::3] = 0.0; tm[1, 0, 7, 7] = -1.0
    dm, lm = torch.full_like(pred, float("nan")), torch.zeros(1)
    capi.check(L, L.fd_l1_loss_masked(pred.data_ptr(), tm.data_ptr(), dm.data_ptr(), lm.data_ptr(), pred.numel(), scratch.data_ptr(), None), "fd_l1_loss_masked")
    p = pred.clone().requires_grad_(True)
    ref = torch_ref.masked_l1(p, tm); ref.backward()
    assert float(lm) == pytest.approx(float(ref), rel=1e-6)
    assert torch.equal(dm, p.grad) and float(dm[0, 0, 0, 0]) == 0.0 and float(dm[0, 0, 6].abs().max()) == 0.0
    tz = torch.zeros_like(tm)
    capi.check(L, L.fd_l1_loss_masked(pred.data_ptr(), tz.data_ptr(), dm.data_ptr(), lm.data_ptr(), pred.numel(), scratch.data_ptr(), None), "fd_l1_loss_masked")
    assert torch.isnan(lm).all() and float(dm.abs().max()) == 0.0
    # SGD vs torch.optim.SGD over two steps (first step initialises the momentum buffer with d)
    params = [torch.randn(1000, generator=g), torch.randn(7, 3, generator=g)]
    ref_params = [q.clone().requires_grad_(True) for q in params]
    opt = torch.optim.SGD(ref_params, lr=0.01, momentum=0.9, weight_decay=1e-4)
    bufs = [torch.zeros_like(q) for q in params]
    for step in range(2):
        grads = [torch.randn(q.shape, generator=g) for q in params]
        for q, gr in zip(ref_params, grads):
            q.grad = gr.clone()
        opt.step()
        table = (capi.SgdTensor * 2)(*[capi.SgdTensor(q.data_ptr(), gr.data_ptr(), b.data_ptr(), q.numel()) for q, gr, b in zip(params, grads, bufs)])
        capi.check(L, L.fd_sgd_step(ctypes.addressof(table), 2, sum(q.numel() for q in params), 0.01, 0.9, 1e-4, 1.0, int(step == 0), None), "fd_sgd_step")
    for q, r in zip(params, ref_params):
        assert torch.allclose(q, r.detach(), rtol=1e-6, atol=1e-7)


def _emu_metrics(L, out, tgt):
    sums = torch.zeros(10, dtype=torch.float64)
    scratch = torch.empty(L.fd_depth_metrics_scratch_bytes(), dtype=torch.uint8)
    capi.check(L, L.fd_depth_metrics(out.data_ptr(), tgt.data_ptr(), out.numel(), sums.data_ptr(), scratch.data_ptr(), None), "fd_depth_metrics")
    s = sums.numpy()
    n = s[0]
    r = {"mse": s[1] / n, "mae": s[2] / n, "lg10": s[3] / n, "absrel": s[4] / n, "delta1": s[5] / n, "delta2": s[6] / n, "delta3": s[7] / n,
         "irmse": np.sqrt(s[8] / n), "imae": s[9] / n}
    r["rmse"] = np.sqrt(r["mse"])
    return r, n


def test_emulated_depth_metrics_known_answer():
    """fd_depth_metrics (the kernels compiled for the CPU emulator) against the reference's own metrics on its own sample triple
    (golden.json `metrics_kat`, generated by importing /root/reference/metrics.py) and against the numpy restatement on a
    ragged case with invalid (0,0) pixels."""
    from oracle import inputs, metrics as ometrics
    L = harness.get_lib("emu")
    pred = torch.from_numpy(np.load(inputs.GOLD + "/sample_tvm_pred.npy")).float().contiguous()
    depth = inputs.load_sample()[1].float().contiguous()
    r, n = _emu_metrics(L, pred, depth)
    assert n == int(((depth > 0) | (pred.reshape(depth.shape) > 0)).sum())
    for k, v in inputs.golden_meta()["metrics_kat"].items():
        assert r[k] == pytest.approx(v, rel=5e-6), k
    g = torch.Generator().manual_seed(3)
    out, tgt = torch.rand(3, 1, 37, 21, generator=g) * 5 + 0.1, torch.rand(3, 1, 37, 21, generator=g) * 5 + 0.1
    out[0, 0, :5], tgt[0, 0, :5] = 0.0, 0.0                     # both zero -> excluded (metrics.py:32)
    r, n = _emu_metrics(L, out, tgt)
    assert n == out.numel() - 5 * 21
    want = ometrics.evaluate(out.numpy(), tgt.numpy())
    for k in ometrics.FIELDS:
        assert r[k] == pytest.approx(want[k], rel=2e-5), k


def test_emulated_depth_metrics_per_frame():
    """fd_depth_metrics_frames: the ten sums per image of a batch equal fd_depth_metrics run on each image alone (bit for bit: same
    blocks, same order), and their totals equal the pooled call up to the order of the last additions."""
    L = harness.get_lib("emu")
    g = torch.Generator().manual_seed(4)
    out, tgt = torch.rand(3, 1, 40, 56, generator=g) * 5 + 0.1, torch.rand(3, 1, 40, 56, generator=g) * 5 + 0.1
    out[2, 0, :9], tgt[2, 0, :9] = 0.0, 0.0
    sums = torch.zeros(3, 10, dtype=torch.float64)
    scratch = torch.empty(L.fd_depth_metrics_frames_scratch_bytes(3), dtype=torch.uint8)
    capi.check(L, L.fd_depth_metrics_frames(out.data_ptr(), tgt.data_ptr(), 3, out[0].numel(), sums.data_ptr(), scratch.data_ptr(), None), "fd_depth_metrics_frames")
    one = torch.zeros(1, 10, dtype=torch.float64)
    for i in range(3):
        o, t = out[i].contiguous(), tgt[i].contiguous()
        capi.check(L, L.fd_depth_metrics_frames(o.data_ptr(), t.data_ptr(), 1, o.numel(), one.data_ptr(), scratch.data_ptr(), None), "fd_depth_metrics_frames")
        assert torch.equal(one[0], sums[i])
    pooled = torch.zeros(10, dtype=torch.float64)
    sc2 = torch.empty(L.fd_depth_metrics_scratch_bytes(), dtype=torch.uint8)
    capi.check(L, L.fd_depth_metrics(out.data_ptr(), tgt.data_ptr(), out.numel(), pooled.data_ptr(), sc2.data_ptr(), None), "fd_depth_metrics")
    assert torch.allclose(sums.sum(0), pooled, rtol=1e-12)
    assert sums[2, 0] == out[2].numel() - 9 * 56
    with pytest.raises(capi.FastDepthError):
        capi.check(L, L.fd_depth_metrics_frames(out.data_ptr(), tgt.data_ptr(), 0, 10, sums.data_ptr(), scratch.data_ptr(), None), "fd_depth_metrics_frames")


LOCAL_TOL = {
    # fp32 plan: every category is fp32 arithmetic on identical inputs
    torch.float32: {"default": 2e-5},
    # bf16 plan: tensors STORED in bf16 carry one rounding (2^-8 relative to the tensor's max is the bound, 2^-9 typical);
    # everything kept in fp32 (tables, running statistics, parameter gradients, prediction) stays at fp32 accuracy
    # (conv_wgrad_lds16: depthwise units whose backward kernels keep bf16 LDS patches -- fp32 result of operands that were rounded from fp32
    # values, against a reference that rounds fp64 values: see harness.local_train_parity)
    torch.bfloat16: {"default": 2e-5, "z": 4e-3, "g_src": 4e-3, "dz": 4e-3, "skip_grad": 4e-3, "conv_wgrad_lds16": 2e-3},
}


def assert_local_parity(rep, dtype):
    tol = LOCAL_TOL[dtype]
    bad = {k: v for k, v in rep.items() if not v[0] <= tol.get(k, tol["default"])}
    assert not bad, "layer-local train parity out of tolerance: %s" % bad
    assert {"z", "bn_table", "running", "bn_grads", "conv_wgrad", "g_src", "skip_grad", "pred", "g_head"} <= set(rep)


@pytest.mark.parametrize("name,plan,dtype,flags", [("tiny", TINY, torch.float32, 0), ("tiny", TINY, torch.bfloat16, capi.FD_TUNE_WGRAD_TILE_ROWS),
                                                   ("ragged", RAGGED, torch.bfloat16, 0), ("tiny_wide", TINY, torch.float32, 0),
                                                   ("tiny_sat6", TINY, torch.float32, 0), ("ragged_sat6", RAGGED, torch.bfloat16, 0),
                                                   ("tiny", TINY, torch.bfloat16, capi.FD_PLAN_NO_BWD_PAIRING),
                                                   ("tiny", TINY, torch.bfloat16, capi.FD_TUNE_DW_BWD_PAIR),
                                                   ("tiny", TINY, torch.float32, capi.FD_TUNE_DW_BWD1),
                                                   ("tiny", TINY, torch.float32, capi.FD_TUNE_DW_PITCH4 | capi.FD_TUNE_DW_PITCH8 | capi.FD_TUNE_DW_WGRAD_TH4),
                                                   ("tiny", TINY, torch.float32, capi.FD_TUNE_DW_FORCE_ROWS), ("tiny_tall", TINY, torch.bfloat16, capi.FD_TUNE_DW_FORCE_ROWS),
                                                   # fd_lane<T, 8>: bf16 LDS patches, 8 channels per work-item -- paired launch / single-staging kernel, ragged channel counts
                                                   ("ragged", RAGGED, torch.bfloat16, capi.FD_TUNE_FORCE_DW_H8 | capi.FD_TUNE_DW_BWD_PAIR),
                                                   ("tiny", TINY, torch.bfloat16, capi.FD_TUNE_FORCE_DW_H8 | capi.FD_TUNE_DW_BWD1),
                                                   # every BatchNorm finalised by its own launch (default at this size: inside the consuming depthwise kernel /
                                                   # the unit's own first backward kernel, fd_bn_finalize_block / fd_bn_bwd_finalize_block)
                                                   # fp32 forward pointwise GEMMs on fd_pw_gemm16_f32<..., TRAIN> (TM = 13 / 7 / 4 as the maps shrink; statistics of whole-stride tiles)
                                                   ("g16", G16, torch.float32, capi.FD_TUNE_FORCE_GEMM16), ("g16_sat6", G16, torch.float32, capi.FD_TUNE_FORCE_GEMM16 | capi.FD_TUNE_NO_CONSUMER_FINALIZE),
                                                   ("tiny", TINY, torch.float32, capi.FD_TUNE_NO_CONSUMER_FINALIZE),
                                                   # ... and the depthwise backward launches finalising their own unit too (off by default: measured no faster)
                                                   ("tiny", TINY, torch.float32, capi.FD_TUNE_DW_BWD_FINALIZE), ("ragged", RAGGED, torch.bfloat16, capi.FD_TUNE_DW_BWD_FINALIZE | capi.FD_TUNE_DW_BWD1),
                                                   ("ragged", RAGGED, torch.bfloat16, capi.FD_TUNE_NO_CONSUMER_FINALIZE)])
def test_emulated_train_step_layer_local(name, plan, dtype, flags):
    """Every unit's forward and backward kernels on their own stored inputs against an fp64 single-unit autograd reference
    (harness.local_train_parity): the rigorous check of the bf16 train plan, whose end-to-end
    comparison is chaotic on a network this small."""
    m = small_model(plan[0], plan[1], seed=3)
    if name.endswith("sat6"):
        harness.saturate_encoder(m)
    g = torch.Generator().manual_seed(9)
    h, w = (160, 224) if name == "tiny_wide" else ((224, 32) if name == "tiny_tall" else (64, 64))   # tiny_tall: map heights 112 ... 7 (the 14-row backward-data tiles; H must be a multiple of 32)
    # tiny_wide: > 256 partial rows per reduction (280 for conv1.3 / decode_conv5.1) -> the sliced (last-arriver) path
    # (default plans: a stride-2 depthwise unit's backward is ONE single-staging kernel (fd_dw_bwd1), the other depthwise units' two kernels and a
    # pointwise unit's two GEMMs share a paired launch; FD_TUNE_DW_BWD1 / _PAIR: the single-staging kernel everywhere / nowhere;
    # FD_PLAN_NO_BWD_PAIRING: every kernel on its own)
    x = torch.rand(2, 3, h, w, generator=g)
    target = 2.0 + torch.rand(2, 1, h, w, generator=g)
    rep = harness.local_train_parity("emu", m, x, target, torch.device("cpu"), dtype=dtype, flags=flags)
    assert_local_parity(rep, dtype)
    info = harness.LAST_LOCAL_INFO
    assert (info["dw_units_with_16bit_lds_patches"] > 0) == bool(flags & capi.FD_TUNE_FORCE_DW_H8)
    # round 6: in a bf16 plan the three 5x5 units on up2 + skip run their backward on the row-walking pixel-pair kernel (fd_dw5_bwd_rows) unless a flag
    # asks for one of the LDS-tiled forms
    lds_forms = capi.FD_TUNE_DW_BWD1 | capi.FD_TUNE_DW_BWD_PAIR | capi.FD_TUNE_DW_BWD_FINALIZE | capi.FD_TUNE_NO_DW5_ROWS | capi.FD_PLAN_NO_BWD_PAIRING
    assert info["dw_units_on_dw5_bwd_rows"] == (3 if dtype == torch.bfloat16 and not flags & lds_forms else 0), info
    # ... and every 3x3 unit of the encoder on fd_dw3_bwd_rows (stride 1: 9 units) / fd_dw3s2_bwd_rows (stride 2: 4 units)
    # (FD_TUNE_DW_FORCE_ROWS keeps the stride-2 units on the older two register-window kernels)
    on_rows = 0 if dtype != torch.bfloat16 or flags & (lds_forms | capi.FD_TUNE_FORCE_DW_H8) else (12 if flags & capi.FD_TUNE_DW_FORCE_ROWS else 16)
    assert info["dw_units_backward_on_row_kernels"] == on_rows, info
    # ... and the forward of all 13 encoder units on fd_dw3_rows_fwd
    assert info["dw_units_on_dw3_rows_fwd"] == (13 if dtype == torch.bfloat16 and not flags & (capi.FD_TUNE_NO_DW5_ROWS | capi.FD_TUNE_FORCE_DW_H8 | capi.FD_TUNE_DW_FORCE_ROWS | capi.FD_TUNE_DW_NO_ROWS) else 0), info
    assert info["dw_units_on_dw5_rows_train"] == (3 if dtype == torch.bfloat16 and not flags & (capi.FD_TUNE_NO_DW5_ROWS | capi.FD_TUNE_FORCE_DW_H8) else 0), info
    # the forms the flags ask for did run: gemm16 train GEMMs (every pointwise unit but the head), in-kernel finalisations forward / backward
    assert info["pw_units_on_gemm16"] == (18 if name.startswith("g16") else 0)
    assert (info["units_finalised_by_consumer"] > 0) == (not flags & capi.FD_TUNE_NO_CONSUMER_FINALIZE)
    if flags & capi.FD_TUNE_NO_CONSUMER_FINALIZE:
        assert info["units_finalising_their_own_backward"] == 0
    elif flags & capi.FD_TUNE_DW_BWD_FINALIZE:
        assert info["units_finalising_their_own_backward"] >= (12 if dtype == torch.float32 else 24)      # depthwise units (+ the 16-bit pointwise ones)
    elif dtype == torch.bfloat16:
        # the apply pass of the 16-bit pointwise units; + the depthwise units whose backward is a row-walking kernel (prologue: fd_bstat_table_block)
        assert info["units_finalising_their_own_backward"] >= (20 if info["dw_units_backward_on_row_kernels"] else 10), info
    else:
        assert info["units_finalising_their_own_backward"] == 0
    if dtype == torch.bfloat16:
        assert "dz" in rep
    if name.endswith("sat6"):
        assert harness.LAST_SAT6_FRAC > 0.005, harness.LAST_SAT6_FRAC


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_statistics_rows_cover_large_and_small_magnitudes(dtype):
    """The BatchNorm statistics rows place every fp32 partial sum exactly into one of three integer accumulators chosen by its binary exponent
    (csrc/fd_device.h: fd_stat_add; forward bins below 2^-8 / below 2^16 / above, backward below 2^-32 / below 2^-8 / above).  Conv weights scaled by
    1e3 resp. 1e-3 (train-mode BatchNorm removes the scale from everything downstream, and divides that unit's weight gradient by it) push the sums of
    z, z^2 of alternating units into the highest and the lowest forward bin, and their gradients' sums across the backward bins; the layer-local fp64
    check must hold exactly as for the unscaled model."""
    m = small_model(TINY[0], TINY[1], seed=3)
    # (only units whose maps hold >= 128 values per channel at this test size: with the 8 values per channel of the 2 x 2 maps and no eps to hide
    # behind -- var >> eps once z is scaled by 1e3 -- the single-pass variance E[z^2] - mean^2 of ANY fp32 implementation loses digits in channels whose
    # |mean| >> std; that is a property of the small test geometry, not of the accumulation under test)
    scaled = [n for n, mod in m.named_modules() if isinstance(mod, torch.nn.Conv2d) and n.split(".")[0] in ("conv0", "conv1", "conv2", "conv3", "decode_conv4", "decode_conv5")]
    assert len(scaled) == 11
    for k, n in enumerate(scaled):
        dict(m.named_modules())[n].weight.data.mul_(1e3 if k % 2 == 0 else 1e-3)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 64, 64, generator=g)
    target = 2.0 + torch.rand(2, 1, 64, 64, generator=g)
    rep = harness.local_train_parity("emu", m, x, target, torch.device("cpu"), dtype=dtype)
    assert_local_parity(rep, dtype)



def test_train_plan_dtype_rules():
    m = small_model(TINY[0], TINY[1], seed=1)
    x = torch.rand(1, 3, 64, 64)
    with pytest.raises(capi.FastDepthError):
        harness.CTrainPlan("emu", m, x, dtype=torch.float16)          # fp16 gradients would need loss scaling: not offered
    odd = small_model((8, 12, 24, 24, 32, 32, 40, 40, 40, 40, 40, 40, 48, 48), (40, 32, 24, 12, 8, 1), seed=1)
    harness.CTrainPlan("emu", odd, x, dtype=torch.float32).close()    # multiples of 4 are fine in fp32
    with pytest.raises(capi.FastDepthError):
        harness.CTrainPlan("emu", odd, x, dtype=torch.bfloat16)       # 16-byte bf16 chunks need multiples of 8


def test_emulated_val_transform_gather():
    """Row f-1: fd_val_transform (kernel compiled for the emulator) gathers raw frames through the composed index tables; checked
    against the PIL restatement of the reference's val_transform (bit-exact, the /255 is done in double like the reference)."""
    import sys
    from oracle import inputs, val_transform as ovt
    sys.path.insert(0, inputs.PKG)
    from dataloaders.nyu import val_index_maps
    L = harness.get_lib("emu")
    g = np.random.default_rng(1)
    rgb = torch.from_numpy(g.integers(0, 256, (2, 480, 640, 3), dtype=np.uint8))
    depth = torch.from_numpy(g.random((2, 480, 640), dtype=np.float32) * 10)
    ymap, xmap = [torch.from_numpy(t) for t in val_index_maps((224, 224))]
    x = torch.empty(2, 3, 224, 224); d = torch.empty(2, 1, 224, 224)
    capi.check(L, L.fd_val_transform(rgb.data_ptr(), depth.data_ptr(), 2, 480, 640, 224, 224, ymap.data_ptr(), xmap.data_ptr(), x.data_ptr(), d.data_ptr(), None), "fd_val_transform")
    for f in range(2):
        want_rgb, want_d = ovt.val_transform(rgb[f].numpy(), depth[f].numpy())
        assert np.array_equal(x[f].permute(1, 2, 0).numpy(), want_rgb.astype(np.float32))
        assert np.array_equal(d[f, 0].numpy(), want_d)
    with pytest.raises(capi.FastDepthError):
        capi.check(L, L.fd_val_transform(rgb.data_ptr(), depth.data_ptr(), 2, 480, 640, 224, 224, ymap.data_ptr(), xmap.data_ptr(), x.data_ptr(), None, None), "fd_val_transform")


@pytest.mark.parametrize("dtype,flags", [(torch.float32, 0), (torch.bfloat16, 0), (torch.bfloat16, capi.FD_TUNE_DW_BWD1)])
def test_emulated_skip_concat_train_step_layer_local(dtype, flags):
    """Row f-3: train step of the concatenating sibling (depthwise MODE 3 in the train forward, backward-data and backward-weights
    kernels: two channel ranges read from / differentiated into two tensors), small widths, layer-local fp64 check."""
    from oracle import inputs
    models = inputs.product_models()
    torch.manual_seed(41)
    enc = (8, 32, 24, 32, 32, 32, 40, 40, 40, 40, 40, 40, 48, 48)          # skips: enc[1] = enc[3] = enc[5] = 32; producers dec[1..3] = 32 (multiples of 32)
    m = harness.randomize_bn(models.MobileNetSkipConcat((64, 64), pretrained=False, channels=(enc, (40, 32, 32, 32, 8, 1))), 42)
    g = torch.Generator().manual_seed(43)
    x = torch.rand(2, 3, 64, 64, generator=g)
    target = 2.0 + torch.rand(2, 1, 64, 64, generator=g)
    rep = harness.local_train_parity("emu", m, x, target, torch.device("cpu"), dtype=dtype, flags=flags)     # (DW_BWD1: MODE 3 through the single-staging backward kernel)
    assert_local_parity(rep, dtype)


def test_autograd_function_returns_gradients():
    """The drop-in autograd entry point (TrainFunction) hands its gradients to autograd: nothing the caller gets (`.grad`, the result of
    torch.autograd.grad, a hook's argument) aliases the plan's flat gradient buffer -- values held across a later backward stay put --, a
    second backward accumulates (old + new), torch.autograd.grad / tensor hooks see the gradients, frozen parameters get none, and a backward
    against a workspace that a later forward overwrote is refused."""
    from fastdepth_hip.train import TrainCore, autograd_forward
    L = harness.get_lib("emu")
    m = small_model(TINY[0], TINY[1], seed=3).train()
    m.conv3[3].weight.requires_grad_(False)                          # a frozen parameter
    core = TrainCore(m, torch.float32, _library=L)
    g = torch.Generator().manual_seed(2)
    x1, x2 = torch.rand(2, 3, 64, 64, generator=g), torch.rand(2, 3, 64, 64, generator=g)
    tgt = 2.0 + torch.rand(2, 1, 64, 64, generator=g)
    seen = []
    m.conv1[0].weight.register_hook(lambda gr: seen.append(gr.clone()))
    params = [p for p in m.parameters() if p.requires_grad]
    base = core.flat_grad.untyped_storage().data_ptr()
    # 1) plain backward: .grad holds the gradients but never the plan's buffer itself; the hook fired with the same values
    loss = (autograd_forward(core, x1) - tgt).abs().mean(); loss.backward()
    assert all(p.grad is not None and p.grad.untyped_storage().data_ptr() != base for p in params)
    held = params[0].grad                                            # a reference a caller keeps across later backwards
    assert m.conv3[3].weight.grad is None
    assert len(seen) == 1 and torch.equal(seen[0], m.conv1[0].weight.grad)
    g1 = [p.grad.clone() for p in params]
    # 2) autograd.grad: returns the gradients, leaves .grad alone
    loss = (autograd_forward(core, x2) - tgt).abs().mean()
    got = torch.autograd.grad(loss, params)
    assert all(t is not None for t in got)
    g2 = [t.clone() for t in got]
    assert any(not torch.equal(a, b) for a, b in zip(g1, g2))
    assert all(torch.equal(p.grad, a) for p, a in zip(params, g1))   # .grad kept step 1's values, untouched by autograd.grad
    assert held is params[0].grad and torch.equal(held, g1[0])
    # ... and the tensors autograd.grad returned survive the NEXT backward of the same model unchanged
    loss = (autograd_forward(core, x1) - tgt).abs().mean()
    again = torch.autograd.grad(loss, params)
    assert all(torch.equal(t, b) for t, b in zip(got, g2)) and all(torch.equal(t, a) for t, a in zip(again, g1))
    # 3) accumulation without zero_grad: old + new
    loss = (autograd_forward(core, x2) - tgt).abs().mean(); loss.backward()
    for p, a, b in zip(params, g1, g2):
        assert torch.allclose(p.grad, a + b, rtol=1e-6, atol=1e-9)
    # 4) accumulation into a foreign .grad tensor
    for p in params:
        p.grad = torch.ones_like(p)
    loss = (autograd_forward(core, x2) - tgt).abs().mean(); loss.backward()
    for p, b in zip(params, g2):
        assert torch.allclose(p.grad, 1.0 + b, rtol=1e-6, atol=1e-9) and p.grad.untyped_storage().data_ptr() != base
    # 5) stale forward
    y_old = autograd_forward(core, x1)
    autograd_forward(core, x2)
    with pytest.raises(capi.FastDepthError):
        (y_old - tgt).abs().mean().backward()


def test_library_exchange_control_flow_on_the_emulator():
    """fd_train_backward_allreduce's control flow -- bucket tiling check, backward ranges, the bf16 cast -> all-reduce -> cast back -- on the CPU
    emulator, whose communicator has exactly one rank (its all-reduce is the identity; RCCL itself needs the HIP build: the GPU tier's
    test_library_issued_rccl_exchange_one_rank).  fp32 exchange: bit-identical to the engine without a group; bf16 exchange: bit-identical to the
    torch.distributed route (both round every gradient to bfloat16 and back)."""
    import copy
    import socket
    import torch.distributed as dist
    from fastdepth_hip.train import TrainEngine
    L = harness.get_lib("emu")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        base = small_model(TINY[0], TINY[1], seed=4).train()
        g = torch.Generator().manual_seed(3)
        x, tgt = torch.rand(2, 3, 32, 32, generator=g), 2.0 + torch.rand(2, 1, 32, 32, generator=g)

        def run(**kw):
            eng = TrainEngine(copy.deepcopy(base), lr=0.01, momentum=0.9, weight_decay=1e-4, _library=L, **kw)
            return eng, [float(eng.step(x, tgt))]
        e0, l0 = run()
        e1, l1 = run(process_group=dist.group.WORLD, force_buckets=True, exchange="library")
        assert e1.comm is not None and len(e1.buckets) == 2 and l0 == l1
        assert all(torch.equal(p0, p1) for (_, _, p0), (_, _, p1) in zip(e0.param_list, e1.param_list))
        e2, l2 = run(process_group=dist.group.WORLD, force_buckets=True, exchange="library", grad_exchange_dtype=torch.bfloat16, dtype=torch.bfloat16)
        e3, l3 = run(process_group=dist.group.WORLD, force_buckets=True, exchange="torch", grad_exchange_dtype=torch.bfloat16, dtype=torch.bfloat16)
        assert e2.comm is not None and e3.comm is None and l2 == l3
        assert all(torch.equal(p2, p3) for (_, _, p2), (_, _, p3) in zip(e2.param_list, e3.param_list))
        # buckets that do not tile the layers n-1 .. 0 are refused
        bad = (capi.GradBucket * 2)(capi.GradBucket(e1.n - 1, 9, e1.flat_grad.data_ptr(), 16, None), capi.GradBucket(7, 0, e1.flat_grad.data_ptr(), 16, None))
        rc = L.fd_train_backward_allreduce(e1._plan.handle, e1._params, e1.c_grads, e1.n, e1._dpred.data_ptr(), e1.comm, bad, 2, None)
        assert rc == -1 and b"do not continue the backward order" in L.fd_last_error()
        # more than one rank is RCCL's business
        h = ctypes.c_void_p()
        uid = (ctypes.c_ubyte * 128)()
        assert L.fd_comm_create(uid, 0, 2, ctypes.byref(h)) == -2
        for e in (e1, e2, e3):
            e.close()
    finally:
        dist.destroy_process_group()
