"""Diagnostic: is the inference forward bit-reproducible at small batches, eager and under hipGraph replay, and which plan feature matters?
usage: python tools/diag_b1.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd")); sys.path.insert(0, REPO)
import torch
import models
from fastdepth_hip.engine import Engine
from fastdepth_hip import capi
torch.manual_seed(0)
base = models.MobileNetSkipAdd((224, 224), pretrained=False).eval()
for flags, tag in ((0, "default"), (capi.FD_PLAN_NO_UNIT_FUSION, "no dwpw units"), (capi.FD_PLAN_NO_EPILOGUE_FUSION, "no epilogue fusion"),
                   (capi.FD_PLAN_NO_GEMM16, "no gemm16"), (capi.FD_PLAN_NO_UNIT_FUSION | capi.FD_PLAN_NO_EPILOGUE_FUSION | capi.FD_PLAN_NO_GEMM16, "round-1 plan")):
    Engine.default_plan_flags = flags
    import copy
    m = copy.deepcopy(base).cuda()
    for b in (1, 2, 4):
        x = torch.rand(b, 3, 224, 224, generator=torch.Generator().manual_seed(b)).cuda()
        with torch.no_grad():
            ys = [m(x).clone() for _ in range(8)]
            eager_same = all(torch.equal(ys[0], y) for y in ys[1:])
            eng = m._engine()
            gs = [eng.forward_graph(x).clone() for _ in range(8)]
            graph_same = all(torch.equal(gs[0], g) for g in gs[1:])
            eq = torch.equal(ys[0], gs[0])
            d = float((ys[0] - gs[0]).abs().max())
            nbad = int((ys[0] != gs[0]).sum())
        print("%-20s B=%d eager reproducible %s | graph reproducible %s | eager == graph %s (max diff %.3e, %d elements differ)" % (tag, b, eager_same, graph_same, eq, d, nbad))
