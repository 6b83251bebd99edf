"""Parity of the HIP path (through the drop-in nn.Module -> Engine -> C ABI) on a real MI355X.

Checks, in order of strength:
  * against the REFERENCE's own outputs (tests/golden/*_out.npy, produced by /root/reference's models.py
    on the container's torch CPU) on the reference's NYU-v2 sample + deterministic variants;
  * against the CPU oracle (oracle/fd_oracle.c) layer by layer and at the output, incl. a pruned plan;
  * delta1 / RMSE reproduced to >= 4 significant digits;
  * at BASELINE.json's full size (B=32, 224x224): against the torch-functional oracle and through
    size-independent properties (frame independence, batch-permutation equivariance, determinism).
Tolerance: 1e-3 relative (max|a-b| / max|b|), fp32 -- the north star's figure.
"""
import numpy as np
import pytest
import torch

import harness
from oracle import inputs, metrics, oracle, torch_ref

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _loaded_libs():
    with open("/proc/self/maps") as f:
        return {l.split()[-1] for l in f if "libfastdepth" in l}


@pytest.mark.parametrize("name", ["base_s0", "sat6_s1", "affine_s2"])
def test_golden_reference_outputs(name):
    m, x, y_ref, meta = inputs.golden_case(name)
    m = m.cuda()
    with torch.no_grad():
        y = m(x.cuda())
    torch.cuda.synchronize()
    assert any(p.endswith("fastdepth_hip/libfastdepth_hip.so") for p in _loaded_libs()), "native library not loaded"
    assert y.shape == y_ref.shape and y.is_cuda and y.dtype == torch.float32
    err = harness.rel_err(y.cpu().numpy(), y_ref.numpy())
    assert err < TOL, err
    # metrics harness agreement (delta1 / RMSE reproduced), first frame vs the sample's ground-truth depth
    depth = inputs.load_sample()[1].numpy()
    got = metrics.evaluate(y[:1].cpu().numpy(), depth)
    want = meta["metrics_vs_sample_depth"]
    for k in ("rmse", "mae", "absrel", "delta1", "delta2", "delta3"):
        assert abs(got[k] - want[k]) <= 1e-4 * max(abs(want[k]), 1e-6) + 1e-7, (k, got[k], want[k])


def test_layerwise_vs_oracle_unpruned():
    m, x, _, _ = inputs.golden_case("base_s0")
    err, per_layer, info = harness.compare_with_oracle("hip", m, x[:2], torch.device("cuda"))
    bad = [(i, e, info[i]) for i, e in enumerate(per_layer) if not e < TOL]
    assert not bad, bad


def test_pruned_plan_vs_oracle():
    models = inputs.product_models()
    torch.manual_seed(11)
    m = harness.randomize_bn(models.MobileNetSkipAdd((224, 224), pretrained=False, channels=models.PRUNED_CHANNELS), 12)
    x = inputs.batch_variants(inputs.load_sample()[0], 3, seed=4)
    err, per_layer, info = harness.compare_with_oracle("hip", m, x, torch.device("cuda"))
    bad = [(i, e, info[i]) for i, e in enumerate(per_layer) if not e < TOL]
    assert not bad, bad


def test_full_batch32_vs_oracle_and_properties():
    m, _, _, _ = inputs.golden_case("base_s0")
    x = inputs.batch_variants(inputs.load_sample()[0], 32, seed=0)
    p = torch_ref.params_from_state(m.state_dict())
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        y_ref = torch.cat([torch_ref.forward(p, x[i:i + 8]) for i in range(0, 32, 8)])
    mg = m.cuda()
    with torch.no_grad():
        y = mg(x.cuda())
        y2 = mg(x.cuda())
        perm = torch.randperm(32, generator=torch.Generator().manual_seed(1))
        yp = mg(x[perm].cuda())
        y1 = torch.cat([mg(x[i:i + 1].cuda()) for i in (0, 7, 31)])
    assert harness.rel_err(y.cpu().numpy(), y_ref.numpy()) < TOL
    assert torch.equal(y, y2), "forward is not deterministic"
    assert torch.equal(yp, y[perm.cuda()]), "frames of a batch are not independent"
    assert harness.rel_err(y1.cpu().numpy(), y[[0, 7, 31]].cpu().numpy()) < 1e-5, "B=1 and B=32 plans disagree"
    # C oracle on a bounded sub-batch as an independent (non-torch) check
    yo = oracle.forward(m.state_dict(), x[:4].numpy())
    assert harness.rel_err(y[:4].cpu().numpy(), yo) < TOL


def test_error_behaviour():
    m, x, _, _ = inputs.golden_case("base_s0")
    with pytest.raises(RuntimeError):
        m(x)                                   # CPU tensor: no fallback
    mg = m.cuda()
    with pytest.raises(RuntimeError):
        mg(torch.rand(1, 3, 228, 304, device="cuda"))      # the reference fails on this size too (skip add)
    with pytest.raises(RuntimeError):
        mg(torch.rand(1, 3, 224, 224, device="cuda", dtype=torch.float64))


def test_repack_after_parameter_update():
    m, x, _, _ = inputs.golden_case("base_s0")
    mg = m.cuda()
    xg = x[:1].cuda()
    with torch.no_grad():
        y0 = mg(xg).clone()
        mg.decode_conv6[1].bias.add_(0.5)            # in-place edit bumps the version counter
        y1 = mg(xg)
    assert float((y1 - y0).mean()) == pytest.approx(0.5, abs=1e-4)


@pytest.mark.parametrize("dtype,tol,mtol", [(torch.float16, 4e-3, 2e-3), (torch.bfloat16, 3e-2, 1e-2)])
def test_16bit_storage_golden_case(dtype, tol, mtol):
    """fp16 / bf16 activation + pointwise-weight storage with fp32 accumulation (BASELINE.json configs 3-5 run in 16 bit).
    The 1e-3 criterion is an fp32 criterion: the REFERENCE itself drifts by 9e-4 (fp16) / 7.6e-3 (bf16) max-rel when its module is
    cast to 16 bit on CPU; bounds here: element-wise `tol`, depth metrics within `mtol` relative."""
    m, x, y_ref, meta = inputs.golden_case("base_s0")
    m = m.cuda().set_compute_dtype(dtype)
    with torch.no_grad():
        y = m(x.cuda())
    assert y.dtype == torch.float32
    assert harness.rel_err(y.cpu().numpy(), y_ref.numpy()) < tol
    got = metrics.evaluate(y[:1].cpu().numpy(), inputs.load_sample()[1].numpy())
    want = meta["metrics_vs_sample_depth"]
    for k in ("rmse", "mae", "absrel", "delta1"):
        assert abs(got[k] - want[k]) <= mtol * max(abs(want[k]), 1e-6) + 1e-6, (k, got[k], want[k])
    m.set_compute_dtype(torch.float32)
    with torch.no_grad():
        assert harness.rel_err(m(x.cuda()).cpu().numpy(), y_ref.numpy()) < TOL      # switching back re-plans in fp32


@pytest.mark.parametrize("dtype,tol,mtol_d1,mtol_rmse", [(torch.float16, 4e-3, 2e-3, 1e-2), (torch.bfloat16, 3e-2, 2e-3, 1e-2)])
def test_16bit_default_plan_batch32_vs_oracle(dtype, tol, mtol_d1, mtol_rmse):
    """The plan `bench.py` times as `other_configs` fp16 / bf16 B=32 (unpruned, DEFAULT flags) compared DIRECTLY with the oracle: the fp32 torch
    restatement in 8-frame chunks, as test_full_batch32_vs_oracle_and_properties does for the fp32 plan.  Asserts through plan.info() that the
    round-3 / round-4 / round-6 kernels are the ones that ran: fd_pw_gemm16_h16 with fused depthwise epilogues, the head on decode_conv5.1's GEMM tile,
    the 8-channel register-window 3x3 kernel and the row-walking pixel-pair 5x5 kernel (fd_dw5_rows) on the three up2 + skip units.  Element-wise bound `tol`; delta1 within
    `mtol_d1` and RMSE within `mtol_rmse` (relative) of the oracle's, per frame against the sample depth map."""
    m, _, _, _ = inputs.golden_case("base_s0")
    x = inputs.batch_variants(inputs.load_sample()[0], 32, seed=0)
    p = torch_ref.params_from_state(m.state_dict())
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        y_ref = torch.cat([torch_ref.forward(p, x[i:i + 8]) for i in range(0, 32, 8)])
    mg = m.cuda().set_compute_dtype(dtype)
    with torch.no_grad():
        y = mg(x.cuda()).cpu()
    info = [st[2] for st in mg._engine().layer_stats(x.cuda())]
    assert sum(s.startswith("pw_gemm16") and "fused dw" in s for s in info) >= 6, info
    assert any("head on its output tile" in s for s in info), info
    assert any(s.startswith("dw3_rows8") for s in info), info
    assert sum(s.startswith("dw5_rows<") for s in info) == 3, info     # decode_conv3 / 4 / 5: the row-walking pixel-pair kernel (round 6)
    assert harness.rel_err(y.numpy(), y_ref.numpy()) < tol
    depth = inputs.load_sample()[1].numpy()
    for i in (0, 5, 17, 31):
        got, want = metrics.evaluate(y[i:i + 1].numpy(), depth), metrics.evaluate(y_ref[i:i + 1].numpy(), depth)
        assert abs(got["delta1"] - want["delta1"]) <= mtol_d1 * max(abs(want["delta1"]), 1e-6) + 1e-6, (i, got["delta1"], want["delta1"])
        assert abs(got["rmse"] - want["rmse"]) <= mtol_rmse * max(abs(want["rmse"]), 1e-6) + 1e-6, (i, got["rmse"], want["rmse"])
    mg.set_compute_dtype(torch.float32)


def test_pruned_fp16_batch64_config5():
    """BASELINE.json configs[4]: pruned plan (irregular channel counts, multiples of 8), batch 64, fp16."""
    models = inputs.product_models()
    torch.manual_seed(11)
    m = harness.randomize_bn(models.MobileNetSkipAdd((224, 224), pretrained=False, channels=models.PRUNED_CHANNELS), 12).eval()
    x = inputs.batch_variants(inputs.load_sample()[0], 64, seed=4)
    yo = oracle.forward(m.state_dict(), x[:6].numpy())
    mg = m.cuda().set_compute_dtype(torch.float16)
    with torch.no_grad():
        y = mg(x.cuda())
    assert harness.rel_err(y[:6].cpu().numpy(), yo) < 1e-2
    err, per_layer, info = harness.compare_with_oracle("hip", m.cpu(), x[:2], torch.device("cuda"), dtype=torch.float16)
    assert max(per_layer) < 2e-2, [(i, e, info[i]) for i, e in enumerate(per_layer) if e >= 2e-2]
    assert sum(s.startswith("dw5_rows<") for s in info) == 3, info     # the pruned widths (200 / 120 / 56 channels) run the pixel-pair kernel with ragged channel blocks


def test_on_device_metrics_and_eval_harness(tmp_path):
    """Rows f-2 / f-1: the fused metrics kernel reproduces the reference's known answer on its own sample triple, and the
    main.py-compatible harness runs a reference-format checkpoint end to end."""
    import sys
    sys.path.insert(0, inputs.PKG)
    import metrics as fd_metrics
    import evaluate as fd_eval
    pred = torch.from_numpy(np.load(inputs.GOLD + "/sample_tvm_pred.npy")).cuda()
    depth = inputs.load_sample()[1].cuda()
    r = fd_metrics.Result()
    r.evaluate(pred, depth)
    kat = inputs.golden_meta()["metrics_kat"]
    for k, v in kat.items():
        assert getattr(r, k) == pytest.approx(v, rel=5e-6), k
    with pytest.raises(RuntimeError):
        r.evaluate(pred.cpu(), depth.cpu())
    # reference checkpoint format: {'epoch', 'best_result', 'model'} with the module pickled whole (main.py:49-57)
    m, x, y_ref, meta = inputs.golden_case("base_s0")
    ck = str(tmp_path / "model_best.pth.tar")
    torch.save({"epoch": 7, "best_result": None, "model": m}, ck)
    sd = tmp_path / "samples"; sd.mkdir()
    for i in range(8):                    # the reference's own NYU sample (tests/golden), eight times
        np.savez(str(sd / ("s%d.npz" % i)), rgb=np.load(inputs.GOLD + "/sample_rgb_u8.npy"), depth=np.load(inputs.GOLD + "/sample_depth.npy"))
    avg = fd_eval.main(["--evaluate", ck, "--samples", str(sd), "--batch-size", "4", "-p", "1"])
    want = meta["metrics_vs_sample_depth"]
    assert avg.rmse == pytest.approx(want["rmse"], rel=1e-4) and avg.delta1 == pytest.approx(want["delta1"], rel=1e-4)


def test_no_skip_sibling_matches_reference_output():
    """Row f-3: `MobileNet('nnconv5dw')` against the reference's own output on the same seeded weights / inputs
    (tests/golden/nnconv5dw_s4_*, generated by oracle/make_golden_siblings.py from /root/reference)."""
    m, x, y_ref, meta = inputs.golden_sibling_case("nnconv5dw_s4")
    m = m.cuda()
    with torch.no_grad():
        y = m(x.cuda()).cpu()
    assert harness.rel_err(y.numpy(), y_ref.numpy()) < 1e-3
    m.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        yb = m(x.cuda()).cpu()
    assert harness.rel_err(yb.numpy(), y_ref.numpy()) < 3e-2


def test_skip_concat_sibling_matches_reference_output():
    """Row f-3: `MobileNetSkipConcat` against the reference's own output (tests/golden/skipconcat_s6_*), fp32 and fp16 storage."""
    m, x, y_ref, meta = inputs.golden_sibling_case("skipconcat_s6")
    m = m.cuda()
    with torch.no_grad():
        y = m(x.cuda()).cpu()
    assert harness.rel_err(y.numpy(), y_ref.numpy()) < 1e-3
    m.set_compute_dtype(torch.float16)
    with torch.no_grad():
        yh = m(x.cuda()).cpu()
    assert harness.rel_err(yh.numpy(), y_ref.numpy()) < 2e-2


def test_gpu_val_transform_and_raw_frame_evaluation(tmp_path):
    """Row f-1: raw 480x640 uint8 frames -> GpuValTransform -> network, against the PIL restatement of the reference's
    val_transform; and the evaluation harness on raw-size .npz samples."""
    import sys
    sys.path.insert(0, inputs.PKG)
    from dataloaders.nyu import GpuValTransform
    from oracle import val_transform as ovt
    import evaluate as fd_eval
    g = np.random.default_rng(2)
    rgb = g.integers(0, 256, (3, 480, 640, 3), dtype=np.uint8)
    depth = (g.random((3, 480, 640), dtype=np.float32) * 9 + 0.7).astype(np.float32)
    x, d = GpuValTransform((224, 224))(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda())
    for f in range(3):
        want_rgb, want_d = ovt.val_transform(rgb[f], depth[f])
        assert np.array_equal(x[f].permute(1, 2, 0).cpu().numpy(), want_rgb.astype(np.float32))
        assert np.array_equal(d[f, 0].cpu().numpy(), want_d)
    for f in range(3):
        np.savez(str(tmp_path / ("%05d.npz" % f)), rgb=rgb[f], depth=depth[f])
    avg = fd_eval.main(["--samples", str(tmp_path), "--batch-size", "2", "-p", "1"])
    assert np.isfinite(avg.rmse) and avg.rmse > 0


def test_large_batch_matches_small_batches_bitwise():
    """Maximum-size edge: B = 160 frames in one plan (5x the headline batch; > 2^31 bytes of intermediate activations are never
    indexed with 32-bit offsets) equals the same frames run as B = 32 batches (frames are independent): bit for bit where both plans run
    the same kernels (fp16 storage with the batch-dependent epilogue fusion switched off -- the plain kernels keep a frame's arithmetic order
    independent of its position), to rounding where the plan picks different kernels for M = 160*HW and M = 32*HW (fp32: fd_pw_gemm16_f32
    sums the two halves of each K tile separately; fp16 default plans: fd_pw_gemm16_h16 on the 14x14 maps of the B = 32 plan only)."""
    import sys
    sys.path.insert(0, inputs.PKG)
    from fastdepth_hip import capi
    from fastdepth_hip.engine import Engine
    m, x, _, _ = inputs.golden_case("base_s0")
    x = inputs.batch_variants(inputs.load_sample()[0], 160, 9).cuda()
    m = m.cuda()
    for dt, flags, tol in ((torch.float32, 0, 2e-6), (torch.float16, capi.FD_PLAN_NO_EPILOGUE_FUSION, 0.0), (torch.float16, 0, 5e-3)):
        eng = Engine(m, dtype=dt, plan_flags=flags)
        with torch.no_grad():
            big = eng.forward(x)
            small = torch.cat([eng.forward(x[i:i + 32]) for i in range(0, 160, 32)])
        if tol == 0.0:
            assert torch.equal(big, small), (dt, flags)
        else:
            assert harness.rel_err(big.cpu().numpy(), small.cpu().numpy()) < tol, (dt, flags)
        assert bool(torch.isfinite(big).all())


@pytest.mark.parametrize("b,h,w,pruned", [(1, 96, 160, False), (3, 192, 256, True), (5, 256, 224, False)])
def test_other_input_shapes_layerwise(b, h, w, pruned):
    """Ragged-shape edge cases (tile edges of the depthwise kernels, partial GEMM tiles, odd batch): non-square inputs, H and W any
    multiples of 32, layer by layer against the C oracle."""
    models = inputs.product_models()
    torch.manual_seed(50 + b)
    m = harness.randomize_bn(models.MobileNetSkipAdd((h, w), pretrained=False, channels=models.PRUNED_CHANNELS if pruned else None), 51 + b)
    x = torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(52 + b))
    err, per_layer, info = harness.compare_with_oracle("hip", m, x, torch.device("cuda"))
    bad = [(i, e, info[i]) for i, e in enumerate(per_layer) if not e < TOL]
    assert not bad and err < TOL, bad


@pytest.mark.parametrize("pruned,b", [(False, 3), (True, 2)])
def test_forced_gemm16_layerwise(pruned, b):
    """fd_pw_gemm16_f32 forced onto every fp32 pointwise layer at 224x224 (row-tile counts 13 / 7, strides below the full tile, ragged
    N and K of the pruned plan), layer by layer against the C oracle; the default plan uses it only where one round of workgroups
    covers the layer (test_full_batch32_vs_oracle_and_properties runs that selection)."""
    models = inputs.product_models()
    torch.manual_seed(70 + b)
    m = harness.randomize_bn(models.MobileNetSkipAdd((224, 224), pretrained=False, channels=models.PRUNED_CHANNELS if pruned else None), 71 + b)
    x = inputs.batch_variants(inputs.load_sample()[0], b, seed=6)
    err, per_layer, info = harness.compare_with_oracle("hip", m, x, torch.device("cuda"), flags=harness.capi.FD_TUNE_FORCE_GEMM16)
    assert sum(s.startswith("pw_gemm16") for s in info) == 18, info
    bad = [(i, e, info[i]) for i, e in enumerate(per_layer) if not e < TOL]
    assert not bad and err < TOL, bad


def test_batch32_plan_selects_gemm16():
    m, x, _, _ = inputs.golden_case("base_s0")
    plan = harness.CPlan("hip", m, inputs.batch_variants(inputs.load_sample()[0], 32, seed=0).cuda(), keep=False)
    info = plan.info()
    plan.close()
    assert sum(s.startswith("pw_gemm16") for s in info) >= 10 and sum("evaluated in the epilogue" in s for s in info) >= 8, info


@pytest.mark.parametrize("b,h,w", [(3, 224, 224), (2, 160, 288)])
def test_dwpw_units_layerwise(b, h, w):
    """fd_dwpw_f32 (depthwise + pointwise unit of a large map as one persistent, wave-specialised kernel) on every eligible pair --
    conv1..conv3 and decode_conv4 / 5 at full width -- with the other layers' outputs kept, layer by layer against the C oracle;
    160 x 288 gives ragged tiles (80 x 144, 40 x 72 maps); batch 3 / 2 use the round-robin tile deal, the batch-32 test below the
    per-XCD image deal."""
    models = inputs.product_models()
    torch.manual_seed(90 + b)
    m = harness.randomize_bn(models.MobileNetSkipAdd((h, w), pretrained=False), 91 + b).eval()
    x = torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(92 + b))
    y_ref, taps_ref = oracle.forward(m.state_dict(), x.numpy(), taps=True)
    cp = harness.CPlan("hip", m, x.cuda(), keep=True, flags=harness.capi.FD_TUNE_FORCE_UNIT_FUSION)
    info = cp.info()
    y = cp.forward(x.cuda()).cpu().numpy()
    assert sum(s.startswith("dwpw<") for s in info) == 5, info
    for i in range(len(taps_ref) - 1):
        if info[i].startswith("(fused into"):
            continue
        e = harness.rel_err(cp.tap(i).cpu().numpy(), taps_ref[i])
        assert e < TOL, (i, e, info[i])
    assert harness.rel_err(y, y_ref) < TOL
    cp.close()


def test_dwpw_units_ragged_batch_dealt_to_xcds():
    """Batch 9 (>= 8: images are dealt to the XCDs, image n on XCD n % 8, so XCD 0 owns two images and the others one) through the default
    plan at 160 x 192 (80 x 96 / 40 x 48 maps: ragged 16- and 8-pixel tiles): output against the C oracle and against the unfused plan."""
    models = inputs.product_models()
    torch.manual_seed(97)
    m = harness.randomize_bn(models.MobileNetSkipAdd((160, 192), pretrained=False), 98).eval()
    x = torch.rand(9, 3, 160, 192, generator=torch.Generator().manual_seed(99))
    y_ref = oracle.forward(m.state_dict(), x.numpy())
    cp = harness.CPlan("hip", m, x.cuda(), keep=False)
    info = cp.info()
    y = cp.forward(x.cuda()).cpu().numpy()
    cp.close()
    assert sum(s.startswith("dwpw<") for s in info) == 3 and sum("head on the accumulators" in s for s in info) == 1, info
    plain = harness.CPlan("hip", m, x.cuda(), keep=False, flags=harness.capi.FD_PLAN_NO_UNIT_FUSION)
    y0 = plain.forward(x.cuda()).cpu().numpy()
    plain.close()
    assert harness.rel_err(y, y_ref) < TOL and harness.rel_err(y, y0) < 1e-5


def test_batch32_plan_selects_dwpw_units():
    m, x, _, _ = inputs.golden_case("base_s0")
    xb = inputs.batch_variants(inputs.load_sample()[0], 32, seed=0).cuda()
    plan = harness.CPlan("hip", m, xb, keep=False)
    info = plan.info()
    y = plan.forward(xb)
    plan.close()
    assert sum(s.startswith("dwpw<") for s in info) == 3, info          # conv1, conv2, decode_conv5 (where the fused unit was measured to pay)
    assert sum("head on the accumulators" in s for s in info) == 1, info   # ... and decode_conv6 (the 32 -> 1 head) rides on decode_conv5's unit
    plain = harness.CPlan("hip", m, xb, keep=False, flags=harness.capi.FD_PLAN_NO_UNIT_FUSION)
    assert not any(s.startswith("dwpw<") for s in plain.info())
    y0 = plain.forward(xb)
    plain.close()
    assert harness.rel_err(y.cpu().numpy(), y0.cpu().numpy()) < 1e-5      # same arithmetic, different summation order in the GEMMs


def test_batched_evaluation_equals_per_image_protocol(tmp_path):
    """The reference evaluates one image at a time and averages the per-image metrics (main.py:40-41 batch size 1, :80-82); RMSE / iRMSE
    of pooled pixels are not that average.  fd_depth_metrics_frames gives the ten sums per image, so the harness at --batch-size 4
    reports exactly what it reports at --batch-size 1, and the per-image values equal the numpy restatement of metrics.py:31-55."""
    import sys
    sys.path.insert(0, inputs.PKG)
    import metrics as fd_metrics
    import evaluate as fd_eval
    from oracle import metrics as ometrics
    g = np.random.default_rng(5)
    for i in range(6):    # raw-style 224 x 224 frames with different depth statistics per image and some invalid pixels
        rgb = g.random((224, 224, 3), dtype=np.float32)
        depth = (0.7 + (2 + 1.5 * i) * g.random((224, 224), dtype=np.float32)).astype(np.float32)
        depth[:5 * i] = 0.0
        np.savez(str(tmp_path / ("f%d.npz" % i)), rgb=rgb, depth=depth)
    a1 = fd_eval.main(["--samples", str(tmp_path), "--batch-size", "1", "-p", "100"])
    a4 = fd_eval.main(["--samples", str(tmp_path), "--batch-size", "4", "-p", "100"])
    for f in ("rmse", "mse", "mae", "absrel", "lg10", "irmse", "imae", "delta1", "delta2", "delta3"):
        assert getattr(a4, f) == pytest.approx(getattr(a1, f), rel=1e-6), f
    out = (torch.rand(3, 1, 64, 48, generator=torch.Generator().manual_seed(2)) * 4 + 0.2).cuda()
    tgt = (torch.rand(3, 1, 64, 48, generator=torch.Generator().manual_seed(3)) * 4 + 0.2).cuda()
    tgt[1, 0, :7] = 0.0; out[1, 0, :7] = 0.0
    per = fd_metrics.Result.evaluate_frames(out, tgt)
    for i, r in enumerate(per):
        want = ometrics.evaluate(out[i].cpu().numpy(), tgt[i].cpu().numpy())
        for k in ometrics.FIELDS:
            assert getattr(r, k) == pytest.approx(want[k], rel=2e-5), (i, k)
    pooled = fd_metrics.Result(); pooled.evaluate(out, tgt)
    assert abs(pooled.rmse - np.mean([r.rmse for r in per])) > 1e-6        # the two protocols really differ


@pytest.mark.parametrize("b,streams", [(1, 1), (4, 1), (4, 2)])
def test_forward_graph_replay_equals_eager_forward(b, streams):
    """Engine.forward_graph (hipGraph replay; bench.py's B=1 latency line) returns bit for bit what the eager forward returns, on first
    capture, on replay with new input CONTENTS in the same buffer, and after a parameter edit (the graph is re-captured with repacked
    weights).  streams = 2 splits the batch over two captured streams with one plan per half: compared with the eager forward of the two
    halves (a plan's kernel selection, and with it the last bits, depends on its batch size -- see test_full_batch32_vs_oracle_and_properties)."""
    m, x, _, _ = inputs.golden_case("base_s0")
    m = m.cuda()
    xs = inputs.batch_variants(inputs.load_sample()[0], b, seed=3).cuda()
    eng = m._engine()
    sub = b // streams

    def eager():
        return torch.cat([m(xs[i:i + sub]) for i in range(0, b, sub)]).clone()
    with torch.no_grad():
        want = eager()
        assert torch.equal(eager(), want), "the eager forward is not reproducible run to run"
        got = eng.forward_graph(xs, streams=streams).clone()
        assert torch.equal(got, want), (float((got - want).abs().max()), int((got != want).sum()))
        xs.copy_(torch.flip(xs, dims=(3,)) * 0.9 + 0.05)                                    # same address, new frames
        want2 = eager()
        got2 = eng.forward_graph(xs, streams=streams).clone()
        assert torch.equal(got2, want2) and not torch.equal(got2, got)
        m.conv5[3].weight.mul_(1.25)                                                        # version counter moves -> re-capture
        want3 = eager()
        got3 = eng.forward_graph(xs, streams=streams).clone()
        assert torch.equal(got3, want3) and not torch.equal(got3, got2)
    with pytest.raises(RuntimeError):
        eng.forward_graph(xs.cpu())


@pytest.mark.parametrize("dtype,ulp", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("pruned,b,flags", [(False, 32, 0), (False, 32, harness.capi.FD_TUNE_FORCE_EPILOGUE_FUSION), (True, 64, harness.capi.FD_TUNE_FORCE_EPILOGUE_FUSION),
                                            (True, 5, harness.capi.FD_TUNE_FORCE_EPILOGUE_FUSION), (False, 3, harness.capi.FD_TUNE_FORCE_GEMM16), (True, 2, harness.capi.FD_TUNE_FORCE_GEMM16)])
def test_16bit_gemm16_fused_epilogues_vs_first_generation_kernels(pruned, b, flags, dtype, ulp):
    """fd_pw_gemm16_h16 (whole frames per workgroup, depthwise consumers in the GEMM epilogue: 9 launches and their round trips fewer per
    forward) at BASELINE's sizes -- unpruned batch 32 and the pruned plan at batch 64 (configs[4]: irregular channel counts) -- against the
    first-generation 16-bit kernels on the same plan, layer by layer: identical rounding points, so every stored tensor agrees to the last
    bit or two of the storage type.  FORCE_GEMM16: the kernel on every pointwise layer (ragged M, strides that are not whole frames)."""
    models = inputs.product_models()
    torch.manual_seed(11)
    m = harness.randomize_bn(models.MobileNetSkipAdd((224, 224), pretrained=False, channels=models.PRUNED_CHANNELS if pruned else None), 12).eval()
    x = inputs.batch_variants(inputs.load_sample()[0], b, seed=4).cuda()
    cap = harness.capi
    new = harness.CPlan("hip", m, x, dtype=dtype, flags=flags)
    old = harness.CPlan("hip", m, x, dtype=dtype, flags=cap.FD_PLAN_NO_GEMM16 | cap.FD_PLAN_NO_EPILOGUE_FUSION)
    y_new, y_new2, y_old = new.forward(x), new.forward(x), old.forward(x)
    assert torch.equal(y_new, y_new2)
    info = new.info()
    used = [s for s in info if s.startswith("pw_gemm16")]
    fused = [s for s in info if "evaluated in the epilogue" in s]
    if flags == cap.FD_TUNE_FORCE_GEMM16:
        assert len(used) == 18, info
    elif flags:
        assert len(used) == 9 and len(fused) == 9, info               # conv6.3 ... conv13.3, decode_conv1.1, each with the next depthwise layer
    else:
        assert len(used) == 6 and len(fused) == 6, info               # the product's pick at batch 32: conv6.3 ... conv11.3 (14x14 maps, one round of workgroups)
    for i in range(len(new.layers) - 1):
        a, r = new.tap(i).double(), old.tap(i).double()
        assert float((a - r).abs().max()) <= 2.5 * ulp * max(float(r.abs().max()), 1e-30), (i, info[i])
    assert harness.rel_err(y_new.cpu().numpy(), y_old.cpu().numpy()) < 4 * ulp
    new.close(); old.close()


@pytest.mark.parametrize("pruned", [False, True])
def test_16bit_head_fusion_and_rows8_match_the_separate_kernels(pruned):
    """Round-3 kernels of the 16-bit plans at full size: the network head on decode_conv5.1's GEMM tile (fd_pw_gemm_head_h16) equals the separate
    head kernel to fp32 rounding (same T-rounded operands, different summation order), and the 8-channel 3x3 depthwise kernel (fd_dw3_rows8) is
    bit-identical to the 4-channel one (same arithmetic per channel)."""
    import sys
    sys.path.insert(0, inputs.PKG)
    from fastdepth_hip import capi
    from fastdepth_hip.engine import Engine
    models = inputs.product_models()
    torch.manual_seed(90)
    m = harness.randomize_bn(models.MobileNetSkipAdd((224, 224), pretrained=False, channels=models.PRUNED_CHANNELS if pruned else None), 91).eval().cuda()
    x = torch.rand(5, 3, 224, 224, generator=torch.Generator().manual_seed(92)).cuda()
    for dt in (torch.float16, torch.bfloat16):
        with torch.no_grad():
            y = Engine(m, dtype=dt).forward(x)
            y_sep_head = Engine(m, dtype=dt, plan_flags=capi.FD_PLAN_NO_EPILOGUE_FUSION).forward(x)
            y_rows4 = Engine(m, dtype=dt, plan_flags=capi.FD_PLAN_NO_ROWS8).forward(x)
        assert torch.equal(y, y_rows4), dt
        assert harness.rel_err(y.cpu().numpy(), y_sep_head.cpu().numpy()) < (5e-3 if dt == torch.float16 else 4e-2), dt   # (NO_EPILOGUE_FUSION also un-fuses the 14x14 pairs: storage-type rounding)
        assert bool(torch.isfinite(y).all())
