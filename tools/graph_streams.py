"""Experiment: eager vs hipGraph replay, 1/2/4 sub-batch streams."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd")); sys.path.insert(0, REPO)
import torch, models
torch.manual_seed(0)
m = models.MobileNetSkipAdd((224, 224), pretrained=False).eval().cuda()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.rand(B, 3, 224, 224, device="cuda"); eng = m._engine()
def bench(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): y = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3, y
with torch.no_grad():
    t, y0 = bench(lambda: m(x)); print("eager            : %.4f ms/step -> %.0f frames/s" % (t, B / t * 1e3))
    for S in (1, 2, 4, 8):
        t, y = bench(lambda: eng.forward_graph(x, S)); print("graph, %d stream(s): %.4f ms/step -> %.0f frames/s   max|diff| vs eager %.2e" % (S, t, B / t * 1e3, float((y - y0).abs().max())))
