#!/usr/bin/env python3
"""bench.py -- frames/s of the FastDepth MobileNetSkipAdd hot path on MI355X (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: re-launches itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one inference forward of a batch of 32 synthetic 224x224 RGB frames already resident in HBM
(BASELINE.json configs[1]: unpruned model, batch 32, fp32, 1xMI355X, HIP kernels).  With N > 1 every rank
runs its own batch of 32 (weak scaling; inference shards over frames with no collective);
the timed region is bracketed by barrier + torch.cuda.synchronize() and the max over ranks is taken.  Setup (plan creation,
weight packing and 0.3 s of untimed forwards that take a freshly leased GPU past its power-management transient: `config.untimed_device_wakeup_steps_before_warmup`)
precedes the W warm-up steps; exactly K full steps are timed.
Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  windows_ms_per_step, host_enqueue_ms_per_step
                    six consecutive windows of K steps (the first IS the headline) and the host's enqueue time per step of the headline window:
                    shows whether a run sat on a clock transient (a later window faster), on the host, or on neither
  roofline          the kernel symbol with the largest share of device time, timed live with HIP events on the launch stream,
                    priced against the fp32 MFMA peak or HBM bandwidth (MI355X_MICROARCH.md); `traffic` from the committed PMC summary
  whole_step        algorithmic bytes / flops of the step and its layer-wise roofline bound sum_l max(bytes/HBM, flops/MFMA)
  train_step, train_step_bf16
                    the other half of BASELINE's metric: fwd + L1 + bwd + [RCCL all-reduce] + SGD at 32 frames per GPU (configs[2], and
                    configs[3] at N = 8), each with its own `roofline` (dominant kernel family, fd_trace) and, for N > 1, the all-reduce time
                    and how much of it backward hides
  other_configs     configs[4] (pruned fp16 B=64), 16-bit storage B=32, B=1 latency -- N = 1 only, each with a `roofline`
  cpu_baseline      N = 1, rank 0: the reference's UNMODIFIED module (kind "reference") when /root/reference exists, otherwise the
                    oracle's torch-functional restatement (kind "port": the same ATen CPU kernels), on the host cores: inference at
                    B in {1, 8, 32} and the train step (L1Loss + SGD) at B in {8, 32}; warm-up + repeated timed runs, median
                    (protocol of the reference's deploy/tx2_run_tvm.py:44-53,77-80); bounded to ~25 s
  train_check       N = 1, rank 0: three SGD steps on 4 frames -- losses of the fp32 and bf16 HIP plans next to the fp64 oracle's
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd"))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak (== fp32 vector peak)
MFMA_H16_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak
REFERENCE = "/root/reference"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step (BASELINE.json configs[1]: 32)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="time budget of the host-CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=10, help="instrumented steps for the per-kernel roofline")
    ap.add_argument("--extra-steps", type=int, default=30, help="timed steps for each entry of `other_configs`; 0 disables")
    ap.add_argument("--train-steps", type=int, default=20, help="timed train steps per plan (`train_step`, `train_step_bf16`); 0 disables")
    ap.add_argument("--grad-exchange", default="f32", choices=["f32", "bf16"], help="dtype of the data-parallel gradient all-reduce (bf16: 7.92 MB instead of 15.84 MB)")
    ap.add_argument("--only", default="", choices=["", "infer", "train_f32", "train_bf16", "f16", "bf16", "pruned_f16"],
                    help="run ONE configuration's timed loop and nothing else (one rocprofv3 invocation per configuration: tools/gpu_round.sh)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (RCCL over xGMI)."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.exit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------------------------------------------------------------------------
def build_model(device, pruned=False):
    import numpy as np
    import torch
    import models
    torch.manual_seed(0)
    m = models.MobileNetSkipAdd((224, 224), pretrained=False, channels=models.PRUNED_CHANNELS if pruned else None)
    gold = os.path.join(REPO, "tests", "golden", "base_s0_bn.npz")
    if os.path.exists(gold) and not pruned:   # calibrated-synthetic BN statistics; random-init conv weights
        bn = np.load(gold)
        m.load_state_dict({k: torch.from_numpy(bn[k]) for k in bn.files}, strict=False)
    return m.eval().to(device)


def layerwise_bound_ms(stats, mfma_peak_tflops):
    """sum over layers of max(bytes / HBM, flops / MFMA): the roofline that bounds the step.  The pointwise GEMMs are
    priced against `mfma_peak_tflops`, everything else (VALU work) against the fp32 vector peak, which equals the fp32 MFMA peak."""
    t = 0.0
    for st in stats:
        name, sym, info, nbytes, flops = st[:5]
        peak = gemm_peak_of(sym, mfma_peak_tflops) if "pw_gemm" in sym else MFMA_F32_PEAK_TFLOPS
        t += max(nbytes / (HBM_PEAK_GBS * 1e9), flops / (peak * 1e12))
    return t * 1e3


def fused_plan_bound_ms(stats, mfma_peak_tflops):
    """The same sum over the plan's LAUNCHES, each priced with the bytes it actually has to move (fd_plan_layer_traffic: a fused launch
    keeps its intermediate tensors on chip) and all the flops it performs: the bound of the plan as built, without credit for bytes that
    fusion removed.  Launches with a pointwise GEMM inside are priced against the MFMA peak (their depthwise part is a few percent of the flops)."""
    t = 0.0
    for name, sym, info, nbytes, flops, needed in stats:
        if not sym:
            continue
        peak = gemm_peak_of(sym, mfma_peak_tflops) if ("pw_gemm" in sym or "dwpw" in sym) else MFMA_F32_PEAK_TFLOPS
        t += max(needed / (HBM_PEAK_GBS * 1e9), flops / (peak * 1e12))
    return t * 1e3


_PMC = None
# kernel families that contain a matrix product (MFMA): classified MFMA- or HBM-bound by which of the two times is larger, not by name
_GEMM_FAMILIES = ("gemm", "fd_pw_bwd", "fd_pw_dgrad", "fd_pw_wgrad", "fd_dwpw", "fd_stem")


def gemm_peak_of(sym, plan_peak_tflops):
    """MFMA peak that prices a kernel's flops: the stem kernels and every *_f32 kernel multiply in fp32 (v_mfma_f32_32x32x2_f32 /
    16x16x4_f32) whatever the plan's storage type; the 16-bit GEMMs run on the 16-bit matrix instructions."""
    return MFMA_F32_PEAK_TFLOPS if ("stem" in sym or "_f32" in sym) else plan_peak_tflops


def pmc_source():
    pmc_traffic("infer", "")
    meta = (_PMC or {}).get("_meta") or {}
    return "profiles/pmc_traffic.json (%s): committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x 2 + WRITE_SIZE per launch; NOT re-measured in this run" % (
        meta.get("source", "round and run not recorded in the file"))


def pmc_traffic(cfg, name):
    """Per-launch HBM bytes of a kernel symbol (inference configurations) or kernel family (train steps) from the committed rocprofv3 --pmc
    summary profiles/pmc_traffic.json ({configuration: {kernel: {"bytes_per_launch", "launches_per_step"}}}, written by tools/archive_round.py)."""
    global _PMC
    if _PMC is None:
        try:
            _PMC = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        except Exception:
            _PMC = {}
    e = (_PMC.get(cfg) or {}).get(name)
    return e


def roofline_of(entry, sym, mfma_peak_tflops, total_ms, cfg="infer"):
    """entry: {"launches", "ms", "bytes", "flops"} summed over the launches of one kernel symbol / family in ONE step."""
    t_s = entry["ms"] / 1e3
    mfma_peak_tflops = gemm_peak_of(sym, mfma_peak_tflops)
    hbm_time, mfma_time = entry["bytes"] / (HBM_PEAK_GBS * 1e9), entry["flops"] / (mfma_peak_tflops * 1e12)
    if any(t in sym for t in _GEMM_FAMILIES) and mfma_time >= hbm_time:
        roof = {"bound": "mfma", "achieved": round(entry["flops"] / t_s / 1e12, 3), "peak": mfma_peak_tflops, "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": round(entry["bytes"] / t_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    e = pmc_traffic(cfg, sym)                                # per-launch HBM bytes from the rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE), if collected
    roof["traffic"] = round(e["bytes_per_launch"], 1) if e else None
    roof["traffic_source"] = pmc_source() if e else None
    roof["time_at_peak_us"] = {"hbm": round(hbm_time / entry["launches"] * 1e6, 2), "mfma": round(mfma_time / entry["launches"] * 1e6, 2)}
    roof.update({"kernel": sym, "launches_per_step": entry["launches"], "avg_launch_us": round(entry["ms"] * 1e3 / entry["launches"], 2),
                 "share_of_device_time": round(entry["ms"] / total_ms, 4),
                 "algorithmic_per_launch": {"bytes": entry["bytes"] / entry["launches"], "flops": entry["flops"] / entry["launches"]}})
    return roof


def inference_profile(eng, x, steps, mfma_peak_tflops, cfg="infer"):
    """Per-kernel device time of the inference forward (HIP events on the launch stream, fd_forward_timed), aggregated by kernel symbol.
    Per kernel: `GBps` = the bytes its launches have to move (fd_plan_layer_traffic) / time; `GBps_unfused_units` = the SURVEY 8(d) bytes of
    the reference units a launch replaces / time (what `roofline.achieved` uses for HBM-bound kernels: the per-unit convention)."""
    import numpy as np
    stats = eng.layer_stats(x, traffic=True)
    acc = np.zeros(len(stats))
    for _ in range(max(steps, 1)):
        _, ms = eng.forward_timed(x)
        acc += np.array(ms)
    acc /= max(steps, 1)
    by_sym = {}
    for (name, sym, info, nbytes, flops, needed), ms in zip(stats, acc):
        if not sym:                  # a depthwise layer evaluated in its producer's epilogue: no launch of its own (its algorithmic work is
            continue                 # credited to the producer's launch by fd_plan_layer_stats)
        e = by_sym.setdefault(sym, {"launches": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0, "needed": 0.0})
        e["launches"] += 1; e["ms"] += float(ms); e["bytes"] += nbytes; e["flops"] += flops; e["needed"] += needed
    total_ms = float(acc.sum())
    dom_sym, dom = max(by_sym.items(), key=lambda kv: kv[1]["ms"])
    kernels = []
    for s, e in sorted(by_sym.items(), key=lambda kv: -kv[1]["ms"]):
        k = {"kernel": s, "launches": e["launches"], "ms_per_step": round(e["ms"], 4), "GBps": round(e["needed"] / (e["ms"] / 1e3) / 1e9, 1),
             "GBps_unfused_units": round(e["bytes"] / (e["ms"] / 1e3) / 1e9, 1), "TFLOPs": round(e["flops"] / (e["ms"] / 1e3) / 1e12, 2)}
        t = pmc_traffic(cfg, s)
        if t:
            k["GBps_pmc"] = round(t["bytes_per_launch"] * e["launches"] / (e["ms"] / 1e3) / 1e9, 1)
        kernels.append(k)
    whole = {"algorithmic_GB": round(sum(s[3] for s in stats) / 1e9, 4), "algorithmic_GFLOP": round(sum(s[4] for s in stats) / 1e9, 3),
             "needed_GB_fused_plan": round(sum(s[5] for s in stats) / 1e9, 4),
             "device_ms_sum_of_kernels": round(total_ms, 4), "roofline_bound_ms": round(layerwise_bound_ms(stats, mfma_peak_tflops), 4),
             "fused_plan_bound_ms": round(fused_plan_bound_ms(stats, mfma_peak_tflops), 4)}
    pm = [pmc_traffic(cfg, s) for s in by_sym]
    if all(pm):
        whole["pmc_traffic_GB"] = round(sum(t["bytes_per_launch"] * e["launches"] for t, e in zip(pm, by_sym.values())) / 1e9, 4)
    roof = roofline_of(dom, dom_sym, mfma_peak_tflops, total_ms, cfg)
    roof["needed_bytes_per_launch"] = dom["needed"] / dom["launches"]
    return roof, whole, kernels, sum(1 for st in stats if st[1])


# kernel families of the train step whose launches move a unit's activations once (fwd 1x + bwd 2x the inference bytes);
# everything else (BatchNorm finalisation, partial reductions, operand packing) is overhead with no algorithmic traffic of its own
_TRAIN_MAJOR = ("gemm_train", "fd_pw_gemm16_f32", "dwconv_train", "dw3_rows_train", "dw5_rows_train", "stem_train", "head_train", "dgrad", "wgrad", "head_bwd<", "fd_dw_bwd<", "fd_dw_bwd1<",
                "fd_dw5_bwd_rows<", "fd_dw3_bwd_rows<", "fd_dw3s2_bwd_rows<", "dw3_rows_fwd", "fd_pw_bwd_")       # ("wgrad" also matches fd_stem_wgrad_rows)
_TRAIN_PAIRED = ("head_bwd<", "fd_dw_bwd<", "fd_dw_bwd1<", "fd_dw5_bwd_rows<", "fd_dw3_bwd_rows<", "fd_dw3s2_bwd_rows<", "fd_pw_bwd_")       # one launch = a unit's backward-data AND backward-weights pass


def train_profile(teng, xs, tgts, stats, steps, mfma_peak_tflops, param_bytes, cfg="train_bf16"):
    """Per-kernel-family device time of the fused train step (fd_trace: HIP events around every launch), with the algorithmic bytes /
    flops of the unit each launch belongs to."""
    import ctypes
    from fastdepth_hip import capi
    L = teng.L
    import torch
    fam = {}
    x = xs[0]
    for i in range(steps):
        capi.check(L, L.fd_trace_begin(), "fd_trace_begin")
        teng.step(xs[i % len(xs)], tgts[i % len(tgts)])
        n = ctypes.c_int32()
        recs = (capi.TraceRecord * 4096)()
        capi.check(L, L.fd_trace_end(torch.cuda.current_stream(x.device).cuda_stream, recs, 4096, ctypes.byref(n)), "fd_trace_end")
        for r in recs[:min(n.value, 4096)]:
            name = r.kernel.decode().strip("()").split("<")[0].strip()
            e = fam.setdefault(name, {"launches": 0, "ms": 0.0, "bytes": 0.0, "flops": 0.0})
            e["launches"] += 1; e["ms"] += r.ms
            raw = r.kernel.decode()
            if any(t in raw for t in _TRAIN_MAJOR) and r.layer >= 0:
                mult = 2.0 if any(t in raw for t in _TRAIN_PAIRED) else 1.0
                e["bytes"] += mult * stats[r.layer][3]; e["flops"] += mult * stats[r.layer][4]
            elif "sgd" in raw:
                e["bytes"] += 5.0 * param_bytes                     # grad read, param read+write, momentum read+write
            elif "l1_loss" in raw:
                e["bytes"] += 3.0 * x.shape[0] * x.shape[2] * x.shape[3] * 4
    for e in fam.values():
        for k in e:
            e[k] /= steps
    total_ms = sum(e["ms"] for e in fam.values())
    dom_sym, dom = max(fam.items(), key=lambda kv: kv[1]["ms"])
    kernels = [{"kernel": s, "launches": round(e["launches"], 1), "ms_per_step": round(e["ms"], 4),
                "GBps": round(e["bytes"] / (e["ms"] / 1e3) / 1e9, 1), "TFLOPs": round(e["flops"] / (e["ms"] / 1e3) / 1e12, 2)}
               for s, e in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])]
    alg_bytes = 3.0 * sum(s[3] for s in stats) + 5.0 * param_bytes
    alg_flops = 3.0 * sum(s[4] for s in stats)
    whole = {"algorithmic_GB": round(alg_bytes / 1e9, 4), "algorithmic_GFLOP": round(alg_flops / 1e9, 3), "device_ms_sum_of_kernels": round(total_ms, 4),
             "launches_per_step": round(sum(e["launches"] for e in fam.values()), 1),
             "roofline_bound_ms": round(3.0 * layerwise_bound_ms(stats, mfma_peak_tflops) + 5.0 * param_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, 4)}
    pm = {f: pmc_traffic(cfg, f) for f in fam}
    if any(pm.values()):                                    # counter traffic of the whole step (families without a PMC record are small: listed)
        whole["pmc_traffic_GB"] = round(sum(t["bytes_per_launch"] * fam[f]["launches"] for f, t in pm.items() if t) / 1e9, 4)
        whole["pmc_traffic_over_algorithmic"] = round(whole["pmc_traffic_GB"] / whole["algorithmic_GB"], 3)
        missing = [f for f, t in pm.items() if not t]
        if missing:
            whole["families_without_pmc_record"] = missing
        for k in kernels:
            t = pm.get(k["kernel"])
            if t:
                k["GBps_pmc"] = round(t["bytes_per_launch"] * fam[k["kernel"]]["launches"] / (fam[k["kernel"]]["ms"] / 1e3) / 1e9, 1)
    return roofline_of(dom, dom_sym, mfma_peak_tflops, total_ms, cfg), whole, kernels


# ------------------------------------------------------------------------------------------------------------------------------------
def cpu_baseline(budget_s):
    """The host-CPU yard-stick of BASELINE.md section 3: the reference's unmodified module (torchvision stub only) when /root/reference is
    present, else the oracle's torch-functional restatement of it (the same ATen CPU kernels), fp32, eval + no_grad for inference,
    L1Loss + SGD(0.01, 0.9, 1e-4) for the train step; warm-up then repeated timed runs, median (deploy/tx2_run_tvm.py:44-53,77-80)."""
    import torch
    import types
    t_start = time.time()
    kind, module, fwd = "port", None, None
    if os.path.isdir(REFERENCE):
        sys.dont_write_bytecode = True                 # importing the reference must not leave __pycache__ in its (read-only) tree
        # the reference's own packages are called `models` / `imagenet` like the product's drop-ins: swap them in sys.modules for the
        # duration of the import only (the reference's module objects stay alive through `module`)
        names = ("torchvision", "torchvision.models", "models", "imagenet", "imagenet.mobilenet")
        saved = {k: sys.modules.pop(k, None) for k in names}
        try:
            import importlib
            stub = types.ModuleType("torchvision"); stub.models = types.ModuleType("torchvision.models")
            sys.modules["torchvision"], sys.modules["torchvision.models"] = stub, stub.models     # only the ResNet classes use it
            sys.path.insert(0, REFERENCE)
            ref_models = importlib.import_module("models")
            assert os.path.abspath(ref_models.__file__).startswith(REFERENCE)
            torch.manual_seed(0)
            module = ref_models.MobileNetSkipAdd((224, 224), pretrained=False)
            kind = "reference"
        except Exception:
            module = None
        finally:
            if REFERENCE in sys.path:
                sys.path.remove(REFERENCE)
            for k in names:
                sys.modules.pop(k, None)
                if saved[k] is not None:
                    sys.modules[k] = saved[k]
    traced = None
    if module is None:
        # the GPU boxes have no /root/reference: replay the TorchScript traces of the reference's own module that oracle/make_ref_trace.py
        # made where the reference is present (oracle/_ref/, git-ignored, travels with the snapshot) -- the reference's operator graph on
        # the same ATen CPU kernels, eager == traced bit for bit at trace time
        ev_p, tr_p = (os.path.join(REPO, "oracle", "_ref", "reference_module_%s.pt" % k) for k in ("eval", "train"))
        if os.path.exists(ev_p) and os.path.exists(tr_p):
            try:
                traced = (torch.jit.load(ev_p), torch.jit.load(tr_p))
                kind = "reference"
            except Exception:
                traced = None
    if traced is not None:
        class Traced:                                      # the two traces behind the small part of nn.Module's surface the timing loops use
            def __init__(self):
                self.cur = traced[0]

            def eval(self):
                self.cur = traced[0]

            def train(self):
                self.cur = traced[1]

            def parameters(self):
                return traced[1].parameters()

            def __call__(self, x):
                return self.cur(x)
        module = Traced()
    if module is None:
        from oracle import torch_ref
        import models
        torch.manual_seed(0)
        proto = models.MobileNetSkipAdd((224, 224), pretrained=False)

        class Port(torch.nn.Module):                      # parameters as a module so that the train leg can use torch.optim.SGD
            def __init__(self):
                super().__init__()
                sd = proto.state_dict()
                self.keys = [k for k in sd if not k.endswith("num_batches_tracked")]
                self.p = torch.nn.ParameterList([torch.nn.Parameter(sd[k].clone(), requires_grad=not ("running" in k)) for k in self.keys])

            def forward(self, x):
                return torch_ref.forward(dict(zip(self.keys, self.p)), x, train=self.training)
        module = Port()
    ncpu = os.cpu_count() or 1

    def run_infer(b, cl, iters, threads):
        torch.set_num_threads(threads)
        module.eval()
        x = torch.rand(b, 3, 224, 224)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        ts = []
        with torch.no_grad():
            for _ in range(2):
                module(x)
            for _ in range(iters):
                t0 = time.perf_counter(); module(x); ts.append(time.perf_counter() - t0)
        return b / statistics.median(ts)

    def run_train(b, iters, threads):
        torch.set_num_threads(threads)
        module.train()
        opt = torch.optim.SGD([p for p in module.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=1e-4)
        x, tgt = torch.rand(b, 3, 224, 224), 0.7 + 9.3 * torch.rand(b, 1, 224, 224)
        crit = torch.nn.L1Loss()
        ts = []
        for i in range(iters + 1):
            t0 = time.perf_counter()
            opt.zero_grad(); loss = crit(module(x), tgt); loss.backward(); opt.step()
            if i:
                ts.append(time.perf_counter() - t0)
        return b / statistics.median(ts)

    # thread count: quick probe at B = 8 (the 256-thread hosts of the GPU boxes run this network fastest on 16-64 threads)
    cand = sorted({min(ncpu, t) for t in (16, 32, 64)})
    probe = {t: run_infer(8, False, 2, t) for t in cand}
    threads = max(probe, key=probe.get)
    rows, train_rows = [], []
    for b, iters in ((1, 10), (8, 10), (32, 3)):
        for cl in (False, True):
            if time.time() - t_start > 0.6 * budget_s and rows:
                break
            rows.append({"batch": b, "channels_last": cl, "threads": threads, "iters": iters, "fps": round(run_infer(b, cl, iters, threads), 1)})
    for b, iters in ((8, 3), (32, 2)):
        if time.time() - t_start > budget_s and train_rows:
            break
        train_rows.append({"batch": b, "threads": threads, "iters": iters, "fps": round(run_train(b, iters, threads), 1)})
    best = max(rows, key=lambda r: r["fps"])
    b32 = max([r for r in rows if r["batch"] == 32] or [best], key=lambda r: r["fps"])
    b1 = max([r for r in rows if r["batch"] == 1] or [best], key=lambda r: r["fps"])
    what = ("TorchScript trace of /root/reference's models.MobileNetSkipAdd (oracle/make_ref_trace.py: the reference's own operator graph, eager == traced bit for bit)" if traced is not None
            else "/root/reference models.MobileNetSkipAdd, unmodified (torchvision stub only)" if kind == "reference"
            else "oracle/torch_ref.py (the ATen CPU conv / batch_norm / hardtanh / upsample / add kernels the reference dispatches to)")
    return {"value": best["fps"], "unit": "frames/s", "cores": threads, "kind": kind,
            "sample": "%s, fp32 eval + no_grad on %d of %d host threads: best of B in {1, 8, 32} x {contiguous, channels_last}, 2 warm-ups + median of "
                      "10 timed runs (3 at B=32); best = B %d, channels_last=%s; %.0f s in total" % (what, threads, ncpu, best["batch"], best["channels_last"], time.time() - t_start),
            "batch32_fps": b32["fps"], "batch1_fps": b1["fps"], "batch1_ms": round(1e3 / b1["fps"], 2), "sweep": rows,
            "train_step": {"value": max(r["fps"] for r in train_rows), "unit": "frames/s", "sweep": train_rows,
                           "sample": "torch.nn.L1Loss + torch.optim.SGD(0.01, 0.9, 1e-4), module in .train(), fp32, 1 warm-up + median"}}


def train_check(dev):
    """Three SGD steps on 4 frames: the losses of the fp32 and bf16 HIP plans next to the fp64 oracle's on the same seed (replaces the
    non-discriminating `final_loss`)."""
    import copy
    import torch
    import models
    from fastdepth_hip.train import TrainEngine
    from oracle import torch_ref
    torch.manual_seed(0)
    base = models.MobileNetSkipAdd((224, 224), pretrained=False)
    base.decode_conv6[1].bias.data.fill_(2.8)
    g = torch.Generator().manual_seed(77)
    x, tgt = torch.rand(4, 3, 224, 224, generator=g), 0.7 + 9.3 * torch.rand(4, 1, 224, 224, generator=g)
    out = {"steps": 3, "batch": 4}
    for dtype, tag in ((torch.float32, "loss_hip_f32"), (torch.bfloat16, "loss_hip_bf16")):
        eng = TrainEngine(copy.deepcopy(base).to(dev).train(), lr=0.01, momentum=0.9, weight_decay=1e-4, dtype=dtype)
        out[tag] = [round(float(eng.step(x.to(dev), tgt.to(dev))), 6) for _ in range(3)]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    p = torch_ref.params_from_state(base.state_dict(), torch.float64, requires_grad=True)
    bufs, losses = {}, []
    for _ in range(3):
        loss, grads = torch_ref.l1_train_grads(p, x.double(), tgt.double())
        torch_ref.sgd_step(p, grads, bufs, 0.01, 0.9, 1e-4)
        losses.append(round(float(loss), 6))
    out["loss_oracle_f64"] = losses
    out["max_rel_dev_f32"] = round(max(abs(a - b) / b for a, b in zip(out["loss_hip_f32"], losses)), 6)
    out["max_rel_dev_bf16"] = round(max(abs(a - b) / b for a, b in zip(out["loss_hip_bf16"], losses)), 6)
    return out


# ------------------------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        self_launch(args)
    import numpy as np  # noqa: F401
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    if args.gpus != world:
        sys.exit("bench.py --gpus %d launched with WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = os.environ.get("FD_BENCH_FORCE_DIST") in ("1", "2")    # exercises the RCCL code path on one rank ("2": with the ncclAllReduce calls themselves elided)
    if world > 1 or force_dist:
        import torch.distributed as dist
        if not dist.is_initialized():
            if "MASTER_ADDR" not in os.environ:
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
            # the compute stream's hardware queue exists BEFORE RCCL creates its own (measured on one rank, scratch probe / DESIGN.md section 12: with the
            # process group initialised before the first device allocation the bucket hand-over of the train step costs +150 us per step, after it +50)
            torch.zeros(1, device=dev)
            torch.cuda.synchronize(dev)
            dist.init_process_group("nccl", device_id=dev)

    import models
    from fastdepth_hip.train import TrainEngine
    model = build_model(dev)
    eng = model._engine()
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(args.batch, 3, 224, 224, generator=g).to(dev)      # synthetic NYU-shaped frames in [0,1), resident in HBM
    # the timed inference loop walks a ring of NRING different input batches (HBM-resident, 19 MB each), so that the network input is not
    # served from the 256 MB Infinity Cache step after step
    NRING = 8
    x_ring = [x] + [torch.rand(args.batch, 3, 224, 224, generator=g).to(dev) for _ in range(NRING - 1)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    last_host_enqueue = [0.0]

    def time_forward(mod, xin, steps, warmup, fn=None):
        ring = xin if isinstance(xin, list) else None
        it = [0]

        def step_default():
            it[0] += 1
            return mod(ring[it[0] % len(ring)] if ring else xin)
        fn = fn or step_default
        with torch.no_grad():
            for _ in range(warmup):
                y = fn()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                y = fn()
            t_host = time.perf_counter() - t0                     # the host has enqueued all K steps; the device is still running them
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            barrier()
        assert torch.isfinite(y).all()
        last_host_enqueue[0] = t_host
        return max_over_ranks(elapsed)

    def make_train_engine(dtype):
        torch.manual_seed(0)
        tm = models.MobileNetSkipAdd((224, 224), pretrained=False)
        tm.decode_conv6[1].bias.data.fill_(2.8)
        tm = tm.to(dev).train()
        kw = dict(lr=0.01, momentum=0.9, weight_decay=1e-4, process_group=(dist.group.WORLD if dist is not None else None), force_buckets=force_dist, dtype=dtype,
                  grad_exchange_dtype=torch.bfloat16 if args.grad_exchange == "bf16" else torch.float32)
        # exchange="auto": the library-issued RCCL exchange where every rank's communicator comes up, otherwise torch.distributed.all_reduce per bucket --
        # decided JOINTLY by the ranks inside TrainEngine (a rank-local fallback here would pair one rank's all-reduce with its peers' rendezvous broadcast)
        return TrainEngine(tm, _elide_collectives=os.environ.get("FD_BENCH_FORCE_DIST") == "2", **kw)   # (measurement hook, one rank: the machinery without the ncclAllReduce calls)

    gt = torch.Generator().manual_seed(1)
    # synthetic depth, U[0.7, 10) m: one target per input batch of the ring (the train loops rotate (input, target) pairs like the inference loop)
    tgt_ring = [(0.7 + 9.3 * torch.rand(args.batch, 1, 224, 224, generator=gt)).to(dev) for _ in range(NRING)]
    tgt = tgt_ring[0]

    def time_train(dtype, tag, steps):
        teng = make_train_engine(dtype)
        for i in range(3):
            loss = teng.step(x_ring[i % NRING], tgt_ring[i % NRING])
        barrier()
        t1 = time.perf_counter()
        for i in range(steps):
            loss = teng.step(x_ring[(i + 3) % NRING], tgt_ring[(i + 3) % NRING])
        torch.cuda.synchronize()
        t_el = time.perf_counter() - t1
        barrier()
        t_el = max_over_ranks(t_el)
        assert torch.isfinite(loss).all()
        res = {"metric": "frames/sec (224x224) train step: fwd + L1 loss + bwd + gradient all-reduce + SGD(momentum, wd)",
               "value": round(world * args.batch * steps / t_el, 1), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": 3,
               "ms_per_step": round(t_el / steps * 1e3, 4), "dtype": tag, "batch_per_gpu": args.batch, "global_batch": world * args.batch,
               "input_batches_in_rotation": NRING,
               "parallelism": ("dp%d: RCCL all-reduce of the %s gradient vector in %d buckets (cut by finish time) on a side stream, overlapped with backward; collectives issued by %s" % (
                   world, "7.92 MB bf16" if args.grad_exchange == "bf16" else "15.84 MB fp32", len(teng.buckets),
                   "libfastdepth_hip.so itself (fd_train_backward_allreduce: one call per step, RCCL bound at run time)" if teng.comm is not None else "torch.distributed, one call per bucket"))
                              if teng.use_comm else "single GPU"}
        if teng.use_comm:        # how long the collectives take and how much of them backward hides (3 instrumented steps, synchronising)
            cs = []
            for _ in range(3):
                teng.step(x, tgt, time_comm=True)
                cs.append(teng.last_comm_us)
            comm = statistics.median(c[0] for c in cs); exposed = statistics.median(c[2] for c in cs)
            res["allreduce"] = {"ranks": world, "buckets": len(teng.buckets), "us_per_step_first_issue_to_last_done": round(comm, 1),
                                "us_exposed_after_backward": round(exposed, 1), "overlap_fraction": round(max(0.0, 1.0 - exposed / comm), 3) if comm > 0 else None}
        if args.profile_steps > 0 and args.only == "":
            eng.set_dtype(dtype)
            stats = eng.layer_stats(x)                      # algorithmic bytes / flops per unit at this storage type
            eng.set_dtype(torch.float32)
            peak = MFMA_F32_PEAK_TFLOPS if dtype == torch.float32 else MFMA_H16_PEAK_TFLOPS
            roof, whole, kernels = train_profile(teng, x_ring, tgt_ring, stats, 3, peak, 4.0 * teng.total, "train_f32" if dtype == torch.float32 else "train_bf16")
            res["roofline"], res["whole_step"], res["kernels"] = roof, whole, kernels[:12]
            res["whole_step"]["frac_of_roofline"] = round(whole["roofline_bound_ms"] / res["ms_per_step"], 4)
        teng.close()                                            # the library's RCCL communicator, before the process group goes away
        return res

    # ---- one configuration only (profiling runs) ------------------------------------------------------------------------------------
    if args.only:
        if args.only == "infer":
            t = time_forward(model, x_ring, args.steps, args.warmup)
        elif args.only in ("f16", "bf16"):
            model.set_compute_dtype(torch.float16 if args.only == "f16" else torch.bfloat16)
            t = time_forward(model, x_ring, args.steps, args.warmup)
        elif args.only == "pruned_f16":
            pm = build_model(dev, pruned=True); pm.set_compute_dtype(torch.float16)
            x64_ring = [torch.cat([x_ring[2 * i], x_ring[2 * i + 1]]) for i in range(NRING // 2)] + [torch.rand(64, 3, 224, 224, generator=g).to(dev) for _ in range(NRING - NRING // 2)]
            t = time_forward(pm, x64_ring, args.steps, args.warmup) / 2.0   # reported per 32 frames for comparability
        else:
            r = time_train(torch.float32 if args.only == "train_f32" else torch.bfloat16, args.only, args.steps)
            t = r["ms_per_step"] * args.steps / 1e3
        finish(dist, {"only": args.only, "steps": args.steps, "ms_per_step": round(t / args.steps * 1e3, 4), "n_gpus": world} if rank == 0 else None)
        return

    # ---- headline: configs[1] ---------------------------------------------------------------------------------------------------------
    # Device wake-up, part of setup like plan creation and weight packing.  A freshly leased GPU does not run at its sustained clocks at once: measured on
    # the round-5 boxes with six back-to-back windows of K = 20 steps (`windows_ms_per_step`, profiles/r05/bench_driver_protocol_*.json), the step
    # time is 0.75 ms, then 0.79 - 0.80 ms for a 15 - 30 ms stretch that begins 40 - 70 ms after the first forward (power management settling), then
    # 0.75 ms for good.  With the round-4 wake-up of 40 forwards (30 ms) the K = 20, W = 5 window (16 ms) started 34 ms in and fell into that stretch
    # in one run out of three (40.1 k instead of 42.4 k frames/s; the round-4 driver run: 39.7 k).  The wake-up therefore runs forwards for
    # WAKEUP_S seconds (>= 40 of them), untimed, BEFORE the W warm-up steps; the timed region is still exactly K full steps.
    WAKEUP_S = 0.3
    PREROLL = 0
    t_wake = time.perf_counter()
    with torch.no_grad():
        while PREROLL < 40 or time.perf_counter() - t_wake < WAKEUP_S:
            for _ in range(20):
                model(x)
            torch.cuda.synchronize()
            PREROLL += 20
    elapsed = time_forward(model, x_ring, args.steps, args.warmup)
    host_enqueue_ms = last_host_enqueue[0] / args.steps * 1e3
    # The headline is the window above (the contract's K steps after W warm-up steps).  NWIN - 1 further windows of K steps each follow it
    # back to back, untimed by the contract: they show whether the first window sat on a clock ramp (later windows faster), on the host
    # (host_enqueue_ms_per_step close to ms_per_step) or on neither.
    NWIN = 6
    windows = [round(elapsed / args.steps * 1e3, 4)] + [round(time_forward(model, x_ring, args.steps, 0) / args.steps * 1e3, 4) for _ in range(NWIN - 1)]
    roof, whole, kernels, n_kernels = inference_profile(eng, x, args.profile_steps, MFMA_F32_PEAK_TFLOPS)
    ms_per_step = elapsed / args.steps * 1e3
    whole["frac_of_roofline"] = round(whole["roofline_bound_ms"] / ms_per_step, 4)
    whole["frac_of_fused_plan_bound"] = round(whole["fused_plan_bound_ms"] / ms_per_step, 4)

    # ---- the train step (BASELINE.json metric: "fwd + train-step"): fp32 plan and bf16 plan (configs[2]; configs[3] at N = 8) ---------
    train, train_bf16 = None, None
    if args.train_steps > 0:
        try:
            train = time_train(torch.float32, "f32", args.train_steps)
            train_bf16 = time_train(torch.bfloat16, "bf16 storage + bf16 MFMA, fp32 accumulate / master weights / statistics", args.train_steps)
        except Exception as e:       # the headline line must survive a failure of the secondary measurement
            train = train or {"error": repr(e)}
            train_bf16 = train_bf16 or {"error": repr(e)}

    # ---- other BASELINE.json configurations, N = 1 only: parity for them is in tests/test_gpu_parity.py ------------------------------
    extras = []
    if args.extra_steps > 0 and world == 1:
        pm = build_model(dev, pruned=True)
        x64_ring = [torch.cat([x_ring[2 * i], x_ring[2 * i + 1]]) for i in range(NRING // 2)] + [torch.rand(64, 3, 224, 224, generator=g).to(dev) for _ in range(NRING - NRING // 2)]
        x64 = x64_ring[0]
        pm.set_compute_dtype(torch.float16)
        dt = time_forward(pm, x64_ring, args.extra_steps, 5) / args.extra_steps
        r64, w64, _, _ = inference_profile(pm._engine(), x64, 3, MFMA_H16_PEAK_TFLOPS, "pruned_f16")
        w64["frac_of_roofline"] = round(w64["roofline_bound_ms"] / (dt * 1e3), 4)
        w64["frac_of_fused_plan_bound"] = round(w64["fused_plan_bound_ms"] / (dt * 1e3), 4)
        extras.append({"config": "configs[4]: pruned plan (mobilenet-nnconv5dw-skipadd-pruned), batch=64, fp16 storage / fp32 accumulate, inference",
                       "value": round(64 / dt, 1), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 4), "dtype": "f16", "input_batches_in_rotation": NRING, "roofline": r64, "whole_step": w64})
        del pm
        for dtype, tag in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
            model.set_compute_dtype(dtype)
            dt = time_forward(model, x_ring, args.extra_steps, 5) / args.extra_steps
            r16, w16, _, _ = inference_profile(eng, x, 3, MFMA_H16_PEAK_TFLOPS, tag)
            w16["frac_of_roofline"] = round(w16["roofline_bound_ms"] / (dt * 1e3), 4)
            w16["frac_of_fused_plan_bound"] = round(w16["fused_plan_bound_ms"] / (dt * 1e3), 4)
            extras.append({"config": "unpruned, batch=32, %s storage / fp32 accumulate, inference" % tag, "value": round(args.batch / dt, 1),
                           "unit": "frames/s", "ms_per_step": round(dt * 1e3, 4), "dtype": tag, "input_batches_in_rotation": NRING, "roofline": r16, "whole_step": w16})
        model.set_compute_dtype(torch.float32)
        x1 = x[:1].contiguous()
        dt = time_forward(model, x1, args.extra_steps, 5, fn=lambda: eng.forward_graph(x1)) / args.extra_steps
        extras.append({"config": "configs[0] on the GPU: unpruned, batch=1, fp32, hipGraph replay (latency; the reference publishes 5.6 ms for the PRUNED model on a Jetson TX2)",
                       "value": round(1 / dt, 1), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 4), "dtype": "f32"})

    line = None
    if rank == 0:
        line = {
            "metric": "frames/sec (224x224) MobileNet-NNConv5dw-skipadd inference forward",
            "value": round(world * args.batch * args.steps / elapsed, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "windows_ms_per_step": windows, "host_enqueue_ms_per_step": round(host_enqueue_ms, 4),
            "data": "synthetic (U[0,1) NYU-v2-shaped frames, random-init weights with calibrated BN statistics)",
            "config": {"workload": "configs[1]: MobileNet-NNConv5(dw)+skipadd unpruned, batch=32 per GPU, 224x224 fp32 "
                                   "inference forward, inputs resident in HBM", "batch_per_gpu": args.batch,
                       "global_batch": world * args.batch, "parallelism": "frames sharded over %d GPU(s), no collective" % world,
                       "kernels_per_step": n_kernels, "rccl_ranks": world if dist is not None else 0,
                       "untimed_device_wakeup_steps_before_warmup": PREROLL, "input_batches_in_rotation": NRING},
            "roofline": roof,
            "whole_step": whole,
            "kernels": kernels,
            "train_step": train,
            "train_step_bf16": train_bf16,
            "other_configs": extras,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
            try:
                line["train_check"] = train_check(dev)
            except Exception as e:
                line["train_check"] = {"error": repr(e)}
    finish(dist, line)


def finish(dist, line):
    """Rank 0's JSON line must be the LAST thing on stdout: RCCL writes a version banner through C stdio, which would otherwise be flushed
    at exit, after Python's print."""
    import ctypes
    if dist is not None:
        dist.destroy_process_group()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if line is not None:
        print(json.dumps(line))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
