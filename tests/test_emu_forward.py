"""CPU-emulated runs of the product's HIP kernels + plan code through the C ABI (tests/hipemu), checked against
the C oracle.  These catch indexing / tiling / barrier / MFMA-layout bugs without a GPU; the real parity tests
are the `gpu`-marked ones in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import harness
from oracle import inputs

TOL = 1e-3    # north-star tolerance: 1e-3 relative, fp32


def small_model(enc, dec, seed):
    models = inputs.product_models()
    torch.manual_seed(seed)
    m = models.MobileNetSkipAdd((64, 64), pretrained=False, channels=(enc, dec))
    return harness.randomize_bn(m, seed + 1)


TINY = ((8, 16, 24, 24, 32, 32, 40, 40, 40, 40, 40, 40, 48, 48), (40, 32, 24, 16, 8, 1))
RAGGED = ((16, 56, 88, 120, 144, 72, 104, 40, 72, 88, 96, 128, 80, 112), (200, 72, 120, 56, 16, 1))   # multiples of 8, like the pruned plan
G16 = ((32, 32, 64, 64, 96, 96, 128, 128, 128, 128, 128, 128, 160, 160), (128, 96, 64, 32, 32, 1))    # every pointwise reduction a multiple of 32 (fd_pw_gemm16_f32 train mode); 96 / 160 outputs: a ragged last 64-column tile


@pytest.mark.parametrize("name,plan,b,hw", [("tiny", TINY, 2, 64), ("ragged", RAGGED, 1, 64), ("tiny_rect", TINY, 1, (32, 96))])
def test_emulated_forward_matches_oracle(name, plan, b, hw):
    h, w = (hw, hw) if isinstance(hw, int) else hw
    m = small_model(plan[0], plan[1], seed=hash(name) % 1000)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(b, 3, h, w, generator=g)
    err, per_layer, info = harness.compare_with_oracle("emu", m, x, torch.device("cpu"))
    bad = [(i, e, info[i]) for i, e in enumerate(per_layer) if not e < TOL]
    assert not bad, "layers out of tolerance: %s" % bad
    assert err < TOL


@pytest.mark.parametrize("name,plan,b,hw", [("ragged", RAGGED, 2, (32, 96)), ("tiny", TINY, 2, 64)])
def test_emulated_gemm16_matches_oracle(name, plan, b, hw):
    """fd_pw_gemm16_f32 (16x16x4 MFMA, k-split wave pairs, leader/follower LDS-DMA, LDS-transposed epilogue) forced onto every
    pointwise layer: the batch / image sizes make M = 1536, 384, 96, 24, 6 (ragged) resp. 2048 ... 8 (tiny), i.e. all three row-tile
    counts (13, 7, 4), strides below the full tile, ragged M, ragged N (not a multiple of 64) and ragged K (not a multiple of 32)."""
    h, w = (hw, hw) if isinstance(hw, int) else hw
    m = small_model(plan[0], plan[1], seed=21)
    x = torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(8))
    err, per_layer, info = harness.compare_with_oracle("emu", m, x, torch.device("cpu"), flags=harness.capi.FD_TUNE_FORCE_GEMM16)
    used = [s for s in info if s.startswith("pw_gemm16")]
    assert len(used) == 18, info
    if name == "ragged":
        assert {s.split("TM=")[1].split(":")[0] for s in used} == {"13", "7", "4"}, used
    if name == "tiny":
        # 64 x 64 frames: from 4 x 4 down a workgroup holds whole frames, so the depthwise consumers run in the GEMM epilogues -- all four
        # variants: 3x3 stride 1, 3x3 stride 2, 5x5, 5x5 on the nearest-x2 upsampling (each checked layer-wise against the oracle above)
        fused = [s for s in info if "evaluated in the epilogue" in s]
        assert {s.split("(dw ")[1].split(" evaluated")[0] for s in fused} >= {"k3 s1", "k3 s2", "k5 s1", "k5 s1 on up2"}, fused
    bad = [(i, e, info[i]) for i, e in enumerate(per_layer) if not e < TOL]
    assert not bad and err < TOL, bad


UNITS = ((32, 64, 128, 128, 256, 256, 40, 40, 40, 40, 40, 40, 48, 48), (40, 256, 128, 64, 32, 1))   # the large-map units at full width


@pytest.mark.parametrize("b,hw", [(2, (64, 64)), (1, (96, 160)), (9, (32, 32))])
def test_emulated_dwpw_units_match_oracle(b, hw):
    """fd_dwpw_f32 (depthwise + pointwise unit of a large map as ONE persistent, wave-specialised kernel: producer waves stage patch
    chunks and run the depthwise taps into the GEMM's A tile, consumer waves run the 32x32x2 MFMAs over all output channels) forced
    onto every eligible pair: conv1 / conv3 (3x3 stride 1), conv2 (3x3 stride 2, 64-pixel tiles), decode_conv4 / 5 (5x5 on up2(low) +
    skip); 1..4 channel chunks, several tiles per workgroup at batch 2, ragged tiles (96 x 160 input: 48 x 80, 24 x 40 maps), checked
    layer by layer against the oracle."""
    from oracle import oracle
    m = small_model(UNITS[0], UNITS[1], seed=31).eval()
    x = torch.rand(b, 3, *hw, generator=torch.Generator().manual_seed(9))
    y_ref, taps_ref = oracle.forward(m.state_dict(), x.numpy(), taps=True)
    cp = harness.CPlan("emu", m, x, keep=True, flags=harness.capi.FD_TUNE_FORCE_UNIT_FUSION)
    info = cp.info()
    y = cp.forward(x).numpy()
    units = [i for i, s in enumerate(info) if s.startswith("dwpw<")]
    assert len(units) == 5, info
    assert {info[i].split("<")[1].split(" +")[0] for i in units} == {"dw k3 s1 mode0", "dw k3 s2 mode0", "dw k5 s1 mode2"}, info
    for i in range(len(taps_ref) - 1):
        if info[i].startswith("(fused into"):
            continue
        e = harness.rel_err(cp.tap(i).numpy(), taps_ref[i])
        assert e < TOL, (i, e, info[i])
    assert harness.rel_err(y, y_ref) < TOL
    cp.close()
    # the default plan (no force flag, no kept activations) selects the kernel only where it was measured to pay: units with <= 64 depthwise
    # channels on maps of >= 28 x 28 pixels (conv1, conv2, decode_conv5)
    cp = harness.CPlan("emu", m, x, keep=False)
    sel = [s for s in cp.info() if s.startswith("dwpw<")]
    if hw != (32, 32):
        # decode_conv5's unit also evaluates the network head (32 -> 1 pointwise, written 2x2) on its accumulators
        assert sum("head on the accumulators" in s for s in sel) == 1 and any(s.startswith("(pointwise head evaluated") for s in cp.info()), cp.info()
    y2 = cp.forward(x).numpy()
    cp.close()
    assert len(sel) == {(64, 64): 2, (96, 160): 3, (32, 32): 0}[hw], sel     # (batch 9: images dealt to XCDs, a ragged last group)
    assert harness.rel_err(y2, y_ref) < TOL


def test_plan_rejects_bad_shapes():
    m = small_model(*TINY, seed=1)
    with pytest.raises(harness.capi.FastDepthError):
        harness.CPlan("emu", m, torch.rand(1, 3, 48, 64))       # not a multiple of 32 (reference fails at the skip add)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 5e-3), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("name,plan", [("tiny", TINY), ("ragged", RAGGED)])
def test_emulated_16bit_forward_matches_oracle(name, plan, dtype, tol):
    """16-bit activation / pointwise-weight storage (fp32 accumulate): bounded drift against the fp32 oracle.  The reference's
    own drift when run in fp16 / bf16 is 9e-4 / 7.6e-3 max-rel on the NYU sample; the tiny random nets used
    here are less forgiving, hence the looser bounds."""
    m = small_model(plan[0], plan[1], seed=21)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(6))
    err, per_layer, info = harness.compare_with_oracle("emu", m, x, torch.device("cpu"), dtype=dtype)
    assert err < tol, (err, max(per_layer))
    assert max(per_layer) < 4 * tol, [(i, e, info[i]) for i, e in enumerate(per_layer) if e >= 4 * tol]


def test_emulated_no_skip_sibling_forward():
    """Row f-3: the no-skip `MobileNet('nnconv5dw')` runs on the same kernels (plan walk `mobilenet.*` / `decoder.*`, nearest x2
    folded into the next unit's read, skip = -1).  Full widths, 32x32 input, against a torch-functional restatement of the
    reference's forward (models.py:244-270, 455-458) built from the product module's own tensors."""
    import torch.nn.functional as F
    models = inputs.product_models()
    torch.manual_seed(21)
    m = harness.randomize_bn(models.MobileNet("nnconv5dw", (32, 32), pretrained=False), 22).eval()
    x = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(23))

    def unit(t, seq):
        mods = list(seq)
        for i in range(0, len(mods), 3):
            conv, bn, act = mods[i:i + 3]
            t = F.conv2d(t.double(), conv.weight.double(), None, conv.stride, conv.padding, 1, conv.groups)
            t = F.batch_norm(t, bn.running_mean.double(), bn.running_var.double(), bn.weight.double(), bn.bias.double(), False, 0.1, bn.eps)
            t = t.clamp(0, 6) if isinstance(act, torch.nn.ReLU6) else t.clamp(min=0)
        return t

    with torch.no_grad():
        t = x
        for blk in m.mobilenet:
            t = unit(t, blk)
        for j in range(1, 6):
            blk = getattr(m.decoder, "conv%d" % j)
            t = unit(unit(t, blk[0]), blk[1])
            t = F.interpolate(t, scale_factor=2, mode="nearest")
        ref = unit(t, m.decoder.conv6)
    plan = harness.CPlan("emu", m, x, keep=False)
    y = plan.forward(x)
    plan.close()
    assert harness.rel_err(y.numpy(), ref.numpy()) < TOL


def test_emulated_skip_concat_sibling_forward():
    """Row f-3: `MobileNetSkipConcat` -- the depthwise kernel reads cat(up2(x), skip) as two channel ranges of two tensors
    (fd_layer_desc.concat, MODE 3).  Full widths, 32x32 input, against a torch-functional restatement of the reference's
    forward (models.py:786-813) built from the product module's own tensors."""
    import torch.nn.functional as F
    models = inputs.product_models()
    torch.manual_seed(31)
    m = harness.randomize_bn(models.MobileNetSkipConcat((32, 32), pretrained=False), 32).eval()
    x = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(33))

    def unit(t, seq):
        mods = []
        for c in seq:
            mods += list(c) if isinstance(c, torch.nn.Sequential) else [c]
        for i in range(0, len(mods), 3):
            conv, bn, act = mods[i:i + 3]
            t = F.conv2d(t.double(), conv.weight.double(), None, conv.stride, conv.padding, 1, conv.groups)
            t = F.batch_norm(t, bn.running_mean.double(), bn.running_var.double(), bn.weight.double(), bn.bias.double(), False, 0.1, bn.eps)
            t = t.clamp(0, 6) if isinstance(act, torch.nn.ReLU6) else t.clamp(min=0)
        return t

    with torch.no_grad():
        t, skips = x, {}
        for i in range(14):
            t = unit(t, getattr(m, "conv%d" % i))
            if i in (1, 3, 5):
                skips[i] = t
        for j in range(1, 6):
            t = unit(t, getattr(m, "decode_conv%d" % j))
            t = F.interpolate(t, scale_factor=2, mode="nearest")
            if j in (2, 3, 4):
                t = torch.cat((t, skips[{2: 5, 3: 3, 4: 1}[j]]), 1)
        ref = unit(t, m.decode_conv6)
    plan = harness.CPlan("emu", m, x, keep=False)
    y = plan.forward(x)
    plan.close()
    assert harness.rel_err(y.numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("dtype,ulp", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("name,plan,b,hw,flags", [("tiny", TINY, 2, (64, 64), harness.capi.FD_TUNE_FORCE_EPILOGUE_FUSION), ("tiny5", TINY, 5, (64, 64), harness.capi.FD_TUNE_FORCE_EPILOGUE_FUSION),
                                                  ("ragged", RAGGED, 2, (64, 64), harness.capi.FD_TUNE_FORCE_EPILOGUE_FUSION),
                                                  ("ragged_forced", RAGGED, 2, (32, 96), harness.capi.FD_TUNE_FORCE_GEMM16),
                                                  ("tiny_forced", TINY, 3, (64, 64), harness.capi.FD_TUNE_FORCE_GEMM16)])
def test_emulated_16bit_gemm16_and_fused_epilogues(name, plan, b, hw, flags, dtype, ulp):
    """fd_pw_gemm16_h16 (16x16x32 MFMA, whole frames per workgroup, depthwise consumer in the epilogue) against the first-generation 16-bit
    kernels (fd_pw_gemm_h16 + separate depthwise launches) on the same plan: both round the pointwise output to the storage type before the
    depthwise layer reads it, so every stored tensor agrees to the last bit or two of the storage type (the k-halves are summed in a different
    order).  FORCE_EPILOGUE_FUSION picks the kernel wherever a depthwise consumer fuses behind it (maps of <= 208 pixels; product plans: only where
    it was measured to pay); FORCE_GEMM16 puts it on every pointwise layer (ragged M / N / K, strides that are not whole frames: no fusion there)."""
    m = small_model(plan[0], plan[1], seed=21).eval()
    x = torch.rand(b, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(8))
    cap = harness.capi
    new = harness.CPlan("emu", m, x, dtype=dtype, flags=flags)
    old = harness.CPlan("emu", m, x, dtype=dtype, flags=cap.FD_PLAN_NO_GEMM16 | cap.FD_PLAN_NO_EPILOGUE_FUSION | cap.FD_PLAN_NO_ROWS8)   # (also: 4- instead of 8-channel 3x3 depthwise kernel)
    y_new, y_old = new.forward(x), old.forward(x)
    info = new.info()
    used = [s for s in info if s.startswith("pw_gemm16")]
    fused = [s for s in info if "evaluated in the epilogue" in s]
    assert not any(s.startswith("pw_gemm16") for s in old.info())
    if flags == cap.FD_TUNE_FORCE_GEMM16:
        assert len(used) == 18, info
    else:
        assert len(used) >= 6 and len(fused) == len(used), info           # picked exactly where a consumer fuses
        if name.startswith("tiny"):
            assert {s.split("(dw ")[1].split(" evaluated")[0] for s in fused} >= {"k3 s1", "k3 s2", "k5 s1", "k5 s1 on up2"}, fused
    n = len(new.layers)
    for i in range(n - 1):
        a, r = new.tap(i).double(), old.tap(i).double()
        assert float((a - r).abs().max()) <= 2.5 * ulp * max(float(r.abs().max()), 1e-30), (i, info[i])
    assert harness.rel_err(y_new.numpy(), y_old.numpy()) < 4 * ulp
    new.close(); old.close()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,plan", [("tiny", TINY), ("ragged", RAGGED)])
def test_emulated_16bit_head_on_the_last_gemm(name, plan, dtype):
    """16-bit product plans evaluate the network head (decode_conv6: Cout -> 1 pointwise + ReLU, nearest x2) on the output tile of decode_conv5.1's
    GEMM (fd_pw_gemm_head_h16: that layer's tensor is neither written nor re-read).  Same arithmetic as the separate head kernel on the stored
    tensor (the tile is rounded to the storage type first), different summation order: the two plans agree to fp32 rounding."""
    m = small_model(plan[0], plan[1], seed=33).eval()
    x = torch.rand(3, 3, 64, 96, generator=torch.Generator().manual_seed(12))
    cap = harness.capi
    fused = harness.CPlan("emu", m, x, keep=False, dtype=dtype)
    plain = harness.CPlan("emu", m, x, keep=False, dtype=dtype, flags=cap.FD_PLAN_NO_EPILOGUE_FUSION)
    info = fused.info()
    assert any("head on its output tile" in s for s in info) and any("pointwise head evaluated" in s for s in info), info
    assert not any("head on its output tile" in s for s in plain.info())
    ya, yb = fused.forward(x), plain.forward(x)
    assert ya.shape == (3, 1, 64, 96) and harness.rel_err(ya.numpy(), yb.numpy()) < 2e-6
    fused.close(); plain.close()


WIDE = ((16, 32, 64, 64, 128, 128, 40, 40, 40, 40, 40, 40, 48, 16), (200, 128, 64, 32, 16, 1))   # 64-channel depthwise blocks (cb = 64), a padded pruned width, a 16-channel block


@pytest.mark.parametrize("dtype,ulp", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("b,hw", [(2, (64, 64)), (1, (64, 96))])
def test_emulated_16bit_depthwise_8_channels_per_work_item(b, hw, dtype, ulp):
    """16-bit plans run the LDS-tiled depthwise layers (the decoder's 5x5 units: plain, on up2, on up2 + skip) with storage-typed LDS patches and
    8 channels (16 bytes) per work-item (fd_dwconv<T, ..., 8>) where that was measured to pay (the large maps) or, as here, under FD_TUNE_FORCE_DW_H8 wherever
    eligible; FD_TUNE_NO_DW_H8 keeps the fp32-patch / 4-channel form everywhere.  Plain and upsampled inputs
    are copied into LDS bit for bit and the taps accumulate in fp32 in the same order, so those layers agree exactly; the up2(low) + skip sum is
    rounded to the storage type on its way into LDS (the 4-channel form keeps it in fp32): one extra rounding of the conv input."""
    m = small_model(WIDE[0], WIDE[1], seed=44).eval()
    x = torch.rand(b, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(13))
    cap = harness.capi
    # (FD_TUNE_NO_DW5_ROWS: since round 6 the up2 + skip units of a product plan run on fd_dw5_rows -- test below; this test keeps the LDS-tiled form on them)
    new = harness.CPlan("emu", m, x, dtype=dtype, flags=cap.FD_PLAN_NO_EPILOGUE_FUSION | cap.FD_TUNE_FORCE_DW_H8 | cap.FD_TUNE_NO_DW5_ROWS)
    old = harness.CPlan("emu", m, x, dtype=dtype, flags=cap.FD_PLAN_NO_EPILOGUE_FUSION | cap.FD_TUNE_NO_DW_H8 | cap.FD_TUNE_NO_DW5_ROWS)
    info = new.info()
    h8 = [i for i, s in enumerate(info) if s.startswith("dwconv<") and "8 channels per work-item" in s]
    assert len(h8) == 5 and {info[i].split("tile ")[1].split(" ")[0].split("x")[2] for i in h8} >= {"64", "32", "16"}, info
    assert not any("8 channels per work-item" in s for s in old.info())
    y_new, y_old = new.forward(x), old.forward(x)
    prev_exact = True
    for i in range(len(new.layers) - 1):
        a, r = new.tap(i).double(), old.tap(i).double()
        d = float((a - r).abs().max()) / max(float(r.abs().max()), 1e-30)
        if i in h8 and prev_exact and "mode2" not in info[i]:
            assert d == 0.0, (i, info[i], d)                  # same inputs, bit-for-bit staging, same accumulation order
        assert d <= 3.0 * ulp, (i, info[i], d)
        prev_exact = prev_exact and d == 0.0
    assert harness.rel_err(y_new.numpy(), y_old.numpy()) < 6 * ulp
    new.close(); old.close()


@pytest.mark.parametrize("dtype,ulp", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("b,hw", [(2, (64, 64)), (1, (64, 96)), (1, (32, 32))])
def test_emulated_16bit_dw5_rows_pixel_pair_kernel(b, hw, dtype, ulp):
    """Round 6: the 5x5 units on up2(low) + skip (decode_conv3 / 4 / 5) of a 16-bit plan run on fd_dw5_rows (fd_kernels_dw5p.h): independent waves walk
    down bands of rows with the input window as PIXEL PAIRS and the taps as 16-bit pairs in registers, v_dot2 accumulation in fp32, raw-buffer access
    with the horizontal padding done by the range check.  Against the LDS-tiled fp32-patch form (FD_TUNE_NO_DW5_ROWS | FD_TUNE_NO_DW_H8) on the SAME stored
    inputs it differs by the rounding of the up2 + skip sum and of the 25 folded taps to the storage type: a few units in the last place of the layer's
    range.  Shapes: 64-channel blocks, a ragged pruned width (200 = 4 x 56 - 24), 16- and 32-channel units (half-empty waves), bands with a ragged last
    band (H = 8 ... 32), the 128-channel wave form (3 strips per row at 64 x 96: odd)."""
    m = small_model(WIDE[0], WIDE[1], seed=45).eval()
    x = torch.rand(b, 3, hw[0], hw[1], generator=torch.Generator().manual_seed(14))
    cap = harness.capi
    new = harness.CPlan("emu", m, x, dtype=dtype, flags=cap.FD_PLAN_NO_EPILOGUE_FUSION)
    old = harness.CPlan("emu", m, x, dtype=dtype, flags=cap.FD_PLAN_NO_EPILOGUE_FUSION | cap.FD_TUNE_NO_DW5_ROWS | cap.FD_TUNE_NO_DW_H8)
    info = new.info()
    rows = [i for i, s in enumerate(info) if s.startswith("dw5_rows<")]
    assert len(rows) == 3 and all("mode2" in info[i] for i in rows), info
    assert not any(s.startswith("dw5_rows<") for s in old.info())
    if hw == (64, 96):
        assert any("64 channel lanes per strip" in info[i] for i in rows), info
    y_new, y_old = new.forward(x), old.forward(x)
    for i in rows:
        # the unit on ITS OWN stored inputs: re-run the reference form's layer i on the new plan's inputs is not possible through the C ABI, so compare
        # the taps of both plans layer by layer -- the layers before the first dw5_rows unit are bit-identical, later ones inherit the earlier difference
        a, r = new.tap(i).double(), old.tap(i).double()
        d = float((a - r).abs().max()) / max(float(r.abs().max()), 1e-30)
        assert d <= 6.0 * ulp, (i, info[i], d)
    first = rows[0]
    for i in range(first):
        assert float((new.tap(i).double() - old.tap(i).double()).abs().max()) == 0.0, (i, info[i])
    assert harness.rel_err(y_new.numpy(), y_old.numpy()) < 8 * ulp
    new.close(); old.close()
