"""Experiment: the batch split into S sub-batches, each on its own HIP stream with its own plan/workspace."""
import argparse, copy, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd")); sys.path.insert(0, REPO)
import torch, models
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--iters", type=int, default=50); a = ap.parse_args()
torch.manual_seed(0)
base = models.MobileNetSkipAdd((224, 224), pretrained=False).eval().cuda()
x = torch.rand(a.batch, 3, 224, 224, device="cuda")
for S in (1, 2, 4, 8):
    ms_ = [copy.deepcopy(base) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    xs = list(x.chunk(S))
    def step():
        for m, s, xi in zip(ms_, streams, xs):
            with torch.cuda.stream(s):
                m(xi)
    with torch.no_grad():
        for _ in range(5): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time; t0 = time.perf_counter()
        for _ in range(a.iters): step()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("streams=%d sub-batch=%d: %.4f ms/step -> %.0f frames/s" % (S, a.batch // S, dt / a.iters * 1e3, a.batch * a.iters / dt))
