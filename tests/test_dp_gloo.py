"""Data-parallel logic of the train step on CPU with the gloo backend, world_size 2 (the N>1 path without GPUs).

What can run without a GPU is everything around the kernels: the bucket partition of the flat gradient buffer (reverse
layer order), the bucket-by-bucket summing all-reduce, the gradient mean folded into the SGD step (grad_scale = 1/world),
and that every rank ends with identical parameters equal to a single-process step on the averaged gradients.  The HIP
kernels themselves are covered by the emulator and GPU tests; here the C-ABI SGD kernel runs in the emulator build."""
import ctypes
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "fast-depth_amd"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import harness
    from fastdepth_hip import capi
    from fastdepth_hip.train import make_buckets
    L = harness.get_lib("emu")
    g = torch.Generator().manual_seed(0)
    sizes = [30, 7, 7, 1200, 40, 40, 64, 8, 8, 5000, 100, 100]             # 4 "layers" x (conv, gamma, beta)
    n_layers = len(sizes) // 3
    params0 = [torch.randn(s, generator=g) for s in sizes]
    params = [q.clone() for q in params0]
    # flat gradient buffer in reverse layer order, like TrainCore
    order = [3 * i + j for i in reversed(range(n_layers)) for j in range(3)]
    total = sum(sizes)
    flat_grad, flat_mom = torch.zeros(total), torch.zeros(total)
    span, off, views = {}, 0, {}
    for idx in order:
        views[idx] = flat_grad[off:off + sizes[idx]]
        lo, hi = span.get(idx // 3, (off, off)); span[idx // 3] = (min(lo, off), off + sizes[idx]); off += sizes[idx]
    buckets = make_buckets([4 * (span[i][1] - span[i][0]) for i in range(n_layers)], 3)
    assert buckets[0][0] == n_layers - 1 and buckets[-1][1] == 0 and all(a >= b for a, b in buckets)
    assert all(buckets[i][1] == buckets[i + 1][0] + 1 for i in range(len(buckets) - 1))      # contiguous, no gaps
    table_rows, off = [], 0
    for idx in order:
        table_rows.append((params[idx].data_ptr(), flat_grad.data_ptr() + 4 * off, flat_mom.data_ptr() + 4 * off, sizes[idx])); off += sizes[idx]
    table = (capi.SgdTensor * len(order))(*[capi.SgdTensor(*r) for r in table_rows])
    all_local = []
    for step in range(2):
        gl = torch.Generator().manual_seed(100 * step + rank)              # rank-dependent "local gradients"
        local = {idx: torch.randn(sizes[idx], generator=gl) for idx in range(len(sizes))}
        all_local.append(local)
        for idx in range(len(sizes)):
            views[idx].copy_(local[idx])
        works = [dist.all_reduce(flat_grad[span[a][0]:span[b][1]], async_op=True) for a, b in buckets]   # bucket slices, as TrainEngine.step
        for w in works:
            w.wait()
        capi.check(L, L.fd_sgd_step(ctypes.addressof(table), len(order), total, 0.01, 0.9, 1e-4, 1.0 / world, int(step == 0), None), "fd_sgd_step")
    gathered = [None] * world
    dist.all_gather_object(gathered, [q.clone() for q in params])
    if rank == 0:
        torch.save({"params": gathered, "params0": params0, "sizes": sizes}, out)
    dist.destroy_process_group()


def test_bucketed_allreduce_and_mean_sgd_world2(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    p_rank0, p_rank1 = r["params"]
    for a, b in zip(p_rank0, p_rank1):
        assert torch.equal(a, b)                                           # replicas stay bit-identical
    # single-process reference: torch.optim.SGD on the MEAN of the two ranks' gradients
    ref = [q.clone().requires_grad_(True) for q in r["params0"]]
    opt = torch.optim.SGD(ref, lr=0.01, momentum=0.9, weight_decay=1e-4)
    for step in range(2):
        for idx, q in enumerate(ref):
            gs = [torch.randn(r["sizes"][idx], generator=torch.Generator().manual_seed(100 * step + rk)) for rk in range(2)]
            # per-rank generators are consumed tensor by tensor in index order: rebuild that stream
        gens = [torch.Generator().manual_seed(100 * step + rk) for rk in range(2)]
        for idx, q in enumerate(ref):
            q.grad = sum(torch.randn(r["sizes"][idx], generator=gens[rk]) for rk in range(2)) / 2
        opt.step()
    for a, q in zip(p_rank0, ref):
        assert torch.allclose(a, q.detach(), rtol=1e-5, atol=1e-6)


def test_make_buckets_covers_real_network():
    from fastdepth_hip.train import make_buckets
    from oracle import inputs
    models = inputs.product_models()
    from fastdepth_hip.plan import layers_of
    ls = layers_of(models.MobileNetSkipAdd((224, 224), pretrained=False))
    nbytes = [4 * (l.conv.weight.numel() + 2 * l.bn.weight.numel()) for l in ls]
    b = make_buckets(nbytes, 4)
    assert 1 <= len(b) <= 4 and b[0][0] == 37 and b[-1][1] == 0
    assert sum(sum(nbytes[i] for i in range(lo, hi + 1)) for hi, lo in b) == sum(nbytes) == 4 * 3960930
