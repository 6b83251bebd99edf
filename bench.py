#!/usr/bin/env python3
"""bench.py -- frames/s of the FastDepth MobileNetSkipAdd hot path on MI355X (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one inference forward of a batch of 32 synthetic 224x224 RGB frames already resident in HBM
(BASELINE.json configs[1]: unpruned model, batch 32, fp32, 1xMI355X, HIP kernels).  With N > 1 every rank
runs its own batch of 32 (weak scaling; inference shards over frames with no collective);
the timed region is bracketed by barrier + torch.cuda.synchronize() and the max over ranks is taken.
Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      for the kernel symbol with the largest share of device time, timed live with HIP events on the
                launch stream (fd_forward_timed), priced against the fp32 MFMA peak or HBM bandwidth
  cpu_baseline  the oracle's torch-functional restatement (the same ATen CPU kernels the reference dispatches
                to) timed on this box's host cores on a bounded sample (N=1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd"))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak (== fp32 vector peak)


def build_model(device):
    import models
    torch.manual_seed(0)
    m = models.MobileNetSkipAdd((224, 224), pretrained=False)
    gold = os.path.join(REPO, "tests", "golden", "base_s0_bn.npz")
    if os.path.exists(gold):        # calibrated-synthetic BN statistics; random-init conv weights
        bn = np.load(gold)
        m.load_state_dict({k: torch.from_numpy(bn[k]) for k in bn.files}, strict=False)
    return m.eval().to(device)


def cpu_baseline(batch, budget_s):
    """Times oracle/torch_ref.py (ATen CPU kernels, fp32, eval) on the host cores: small sweep over
    (batch, threads, memory format), bounded by `budget_s` seconds in total.  Returns the best frames/s."""
    from oracle import torch_ref
    import models
    torch.manual_seed(0)
    p = torch_ref.params_from_state(models.MobileNetSkipAdd((224, 224), pretrained=False).state_dict())
    ncpu = os.cpu_count() or 1
    thread_opts = sorted({min(ncpu, t) for t in (16, 32, 64, 128)})
    best, t_start, tried = None, time.time(), []
    for threads in thread_opts:
        for b in (8, batch):
            for cl in (False, True):
                if time.time() - t_start > budget_s:
                    break
                torch.set_num_threads(threads)
                x = torch.rand(b, 3, 224, 224)
                if cl:
                    x = x.contiguous(memory_format=torch.channels_last)
                with torch.no_grad():
                    torch_ref.forward(p, x)                       # warm-up
                    t0 = time.time(); n = 0
                    while n < 3 and (n == 0 or time.time() - t0 < budget_s / 8):
                        torch_ref.forward(p, x); n += 1
                    dt = (time.time() - t0) / n
                fps = b / dt
                tried.append({"batch": b, "threads": threads, "channels_last": cl, "fps": round(fps, 1)})
                if best is None or fps > best["fps"]:
                    best = {"fps": fps, "threads": threads, "batch": b, "channels_last": cl}
    return {"value": round(best["fps"], 2), "unit": "frames/s", "cores": best["threads"], "kind": "port",
            "sample": "oracle/torch_ref.py (ATen CPU conv/batch_norm/hardtanh/upsample/add, fp32 eval), best of a "
                      "%.0f s sweep: batch %d, %d threads, channels_last=%s; %d configs tried"
                      % (budget_s, best["batch"], best["threads"], best["channels_last"], len(tried)),
            "sweep": tried}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step (BASELINE.json configs[1]: 32)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="time budget of the host-CPU baseline sweep")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=10, help="instrumented steps for the per-kernel roofline")
    ap.add_argument("--extra-steps", type=int, default=30, help="timed steps for each entry of `other_configs` (pruned fp16 B=64, "
                    "bf16 / fp16 B=32, fp32 B=1 latency); 0 disables")
    ap.add_argument("--train-steps", type=int, default=20, help="timed fp32 train steps (fwd + L1 + bwd [+ RCCL all-reduce] + SGD) reported as "
                    "`train_step`; 0 disables")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("FD_BENCH_FORCE_DIST") == "1":      # the env switch exercises the RCCL code path on one rank
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    model = build_model(dev)
    eng = model._engine()
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(args.batch, 3, 224, 224, generator=g).to(dev)      # synthetic NYU-shaped frames in [0,1), resident in HBM

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            y = model(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = model(x)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert torch.isfinite(y).all()

    # per-kernel device time, HIP events on the launch stream
    stats = eng.layer_stats(x)
    acc = np.zeros(len(stats))
    for _ in range(max(args.profile_steps, 1)):
        _, ms = eng.forward_timed(x)
        acc += np.array(ms)
    acc /= max(args.profile_steps, 1)
    by_sym = {}
    for (name, sym, info, nbytes, flops), ms in zip(stats, acc):
        e = by_sym.setdefault(sym, {"launches": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0})
        e["launches"] += 1; e["ms"] += float(ms); e["bytes"] += nbytes; e["flops"] += flops
    dom_sym, dom = max(by_sym.items(), key=lambda kv: kv[1]["ms"])
    t_s = dom["ms"] / 1e3
    hbm_time, mfma_time = dom["bytes"] / (HBM_PEAK_GBS * 1e9), dom["flops"] / (MFMA_F32_PEAK_TFLOPS * 1e12)
    is_gemm = dom_sym.startswith("fd_pw_gemm")
    if is_gemm and mfma_time >= hbm_time:
        roof = {"bound": "mfma", "achieved": round(dom["flops"] / t_s / 1e12, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": round(dom["bytes"] / t_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    roof["traffic"] = None
    traffic_file = os.path.join(REPO, "profiles", "pmc_traffic.json")   # per-launch HBM bytes from rocprofv3 --pmc passes, if collected
    if os.path.exists(traffic_file):
        try:
            roof["traffic"] = json.load(open(traffic_file)).get(dom_sym)
        except Exception:
            pass
    roof.update({"kernel": dom_sym, "launches_per_step": dom["launches"], "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
                 "share_of_device_time": round(dom["ms"] / float(acc.sum()), 4),
                 "algorithmic_per_launch": {"bytes": dom["bytes"] / dom["launches"], "flops": dom["flops"] / dom["launches"]}})
    kernels = [{"kernel": s, "launches": e["launches"], "ms_per_step": round(e["ms"], 4),
                "GBps": round(e["bytes"] / (e["ms"] / 1e3) / 1e9, 1), "TFLOPs": round(e["flops"] / (e["ms"] / 1e3) / 1e12, 2)}
               for s, e in sorted(by_sym.items(), key=lambda kv: -kv[1]["ms"])]
    total_bytes = sum(s[3] for s in stats); total_flops = sum(s[4] for s in stats)

    # ---- secondary measurement: the train step (BASELINE.json metric: "fwd + train-step"), same batch per GPU: fp32 plan
    # (`train_step`) and bf16 plan (`train_step_bf16`, BASELINE.json configs[2]/[3]: bf16 storage + bf16 MFMA, fp32 masters)
    train, train_bf16 = None, None
    if args.train_steps > 0:
        from fastdepth_hip.train import TrainEngine
        import models

        def time_train(dtype, tag):
            torch.manual_seed(0)
            tm = models.MobileNetSkipAdd((224, 224), pretrained=False)
            tm.decode_conv6[1].bias.data.fill_(2.8)
            tm = tm.to(dev).train()
            teng = TrainEngine(tm, lr=0.01, momentum=0.9, weight_decay=1e-4, process_group=(dist.group.WORLD if dist is not None else None),
                               force_buckets=os.environ.get("FD_BENCH_FORCE_DIST") == "1", dtype=dtype)
            gt = torch.Generator().manual_seed(1)
            tgt = (0.7 + 9.3 * torch.rand(args.batch, 1, 224, 224, generator=gt)).to(dev)     # synthetic depth, U[0.7, 10) m
            for _ in range(3):
                loss = teng.step(x, tgt)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.train_steps):
                loss = teng.step(x, tgt)
            torch.cuda.synchronize()
            t_el = time.perf_counter() - t1
            barrier()
            tt = torch.tensor([t_el], dtype=torch.float64, device=dev)
            if dist is not None:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_el = float(tt.item())
            assert torch.isfinite(loss).all()
            return {"metric": "frames/sec (224x224) train step: fwd + L1 loss + bwd + gradient all-reduce + SGD(momentum, wd)",
                    "value": round(world * args.batch * args.train_steps / t_el, 1), "unit": "frames/s", "steps": args.train_steps, "warmup": 3,
                    "ms_per_step": round(t_el / args.train_steps * 1e3, 4), "dtype": tag, "batch_per_gpu": args.batch,
                    "parallelism": "dp%d (RCCL all-reduce of the 15.84 MB fp32 gradient vector, %d buckets overlapped with backward)" % (world, len(teng.buckets))
                                   if world > 1 else "single GPU", "final_loss": round(float(loss), 5)}

        try:
            train = time_train(torch.float32, "f32")
            train_bf16 = time_train(torch.bfloat16, "bf16 storage + bf16 MFMA, fp32 accumulate / master weights / statistics")
        except Exception as e:       # the headline line must survive a failure of the secondary measurement
            train = train or {"error": repr(e)}
            train_bf16 = train_bf16 or {"error": repr(e)}

    # ---- other BASELINE.json configurations, measured briefly on rank 0 only (N=1): parity for them is in tests/test_gpu_parity.py
    extras = []
    if args.extra_steps > 0 and world == 1:
        import models

        def timed(mod, xin, steps, fn=None):
            fn = fn or (lambda: mod(xin))
            with torch.no_grad():
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for _ in range(steps):
                    fn()
                torch.cuda.synchronize()
            return (time.perf_counter() - t0_) / steps

        torch.manual_seed(0)
        pm = models.MobileNetSkipAdd((224, 224), pretrained=False, channels=models.PRUNED_CHANNELS).eval().to(dev)
        x64 = torch.rand(64, 3, 224, 224, generator=g).to(dev)
        pm.set_compute_dtype(torch.float16)
        dt = timed(pm, x64, args.extra_steps)
        extras.append({"config": "configs[4]: pruned plan (mobilenet-nnconv5dw-skipadd-pruned), batch=64, fp16 storage / fp32 accumulate, inference",
                       "value": round(64 / dt, 1), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 4), "dtype": "f16"})
        for dtype, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
            model.set_compute_dtype(dtype)
            dt = timed(model, x, args.extra_steps)
            extras.append({"config": "unpruned, batch=32, %s storage / fp32 accumulate, inference" % tag, "value": round(args.batch / dt, 1),
                           "unit": "frames/s", "ms_per_step": round(dt * 1e3, 4), "dtype": tag})
        model.set_compute_dtype(torch.float32)
        x1 = x[:1].contiguous()
        dt = timed(model, x1, args.extra_steps, fn=lambda: eng.forward_graph(x1))
        extras.append({"config": "unpruned, batch=1, fp32, hipGraph replay (latency; the reference publishes 5.6 ms for the PRUNED model on a Jetson TX2)",
                       "value": round(1 / dt, 1), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 4), "dtype": "f32"})
        del pm

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "frames/sec (224x224) MobileNet-NNConv5dw-skipadd inference forward",
            "value": round(world * args.batch * args.steps / elapsed, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (U[0,1) NYU-v2-shaped frames, random-init weights with calibrated BN statistics)",
            "config": {"workload": "configs[1]: MobileNet-NNConv5(dw)+skipadd unpruned, batch=32 per GPU, 224x224 fp32 "
                                   "inference forward, inputs resident in HBM", "batch_per_gpu": args.batch,
                       "global_batch": world * args.batch, "parallelism": "frames sharded over %d GPU(s), no collective" % world,
                       "kernels_per_step": len(stats)},
            "roofline": roof,
            "whole_step": {"algorithmic_GB": round(total_bytes / 1e9, 4), "algorithmic_GFLOP": round(total_flops / 1e9, 3),
                           "device_ms_sum_of_kernels": round(float(acc.sum()), 4),
                           "roofline_bound_ms": None},
            "kernels": kernels,
            "train_step": train,
            "train_step_bf16": train_bf16,
            "other_configs": extras,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.batch, args.cpu_seconds)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
