"""Copies the judged summaries of a tools/gpu_round.sh visit from gpurun_out/<tag>/ into profiles/<round>/ (tracked): one
kernel_stats.csv per configuration (its own rocprofv3 run), the PMC per-kernel means, the bench line, test and smoke logs."""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fd_demangle import demangle
tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join("gpurun_out", tag), os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
for name, out in (("profile_summary.txt", "profile_summary.txt"), ("profile_summary.json", "profile_summary.json"),
                  ("bench.json", "bench_n1.json"), ("pytest_gpu.log", "pytest_gpu.log"), ("smoke.log", "smoke.log"), ("bf16_grad_probe.txt", "bf16_grad_probe.txt"),
                  ("bench_driver_protocol_1.json", "bench_driver_protocol_1.json"), ("bench_driver_protocol_2.json", "bench_driver_protocol_2.json"),
                  ("bench_driver_protocol_3.json", "bench_driver_protocol_3.json")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, out))
if os.path.exists(os.path.join(src, "dist_overhead.txt")):
    shutil.copy(os.path.join(src, "dist_overhead.txt"), os.path.join(dst, "dist_overhead_one_rank.txt"))
for d in glob.glob(os.path.join(src, "pmc_valu_*")):
    if os.path.exists(os.path.join(d, "summary.txt")):
        cfg = os.path.basename(d)[len("pmc_valu_"):]
        os.makedirs(os.path.join(dst, cfg), exist_ok=True)
        shutil.copy(os.path.join(d, "summary.txt"), os.path.join(dst, cfg, "pmc_valu_lds_wait_per_kernel.txt"))
for cfg in ("infer", "train_f32", "train_bf16", "f16", "bf16", "pruned_f16"):
    m = glob.glob(os.path.join(src, "prof_" + cfg, "**", "*kernel_stats.csv"), recursive=True)
    if m:
        os.makedirs(os.path.join(dst, cfg), exist_ok=True)
        shutil.copy(m[0], os.path.join(dst, cfg, "kernel_stats.csv"))
        if os.path.exists(os.path.join(src, "prof_%s.json" % cfg)):
            shutil.copy(os.path.join(src, "prof_%s.json" % cfg), os.path.join(dst, cfg, "bench_only.json"))
for d in glob.glob(os.path.join(src, "pmc_*")):
    if not os.path.isdir(d) or os.path.basename(d).startswith("pmc_valu_"):        # (the SQ accounting passes are archived as their summaries, above)
        continue
    m = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not m:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(m[0])):
        acc[(demangle(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    rest = os.path.basename(d)[4:]
    cfg = next(c for c in ("train_bf16", "train_f32", "pruned_f16", "infer", "f16", "bf16") if rest.startswith(c + "_"))
    ctr = rest[len(cfg) + 1:]
    os.makedirs(os.path.join(dst, cfg), exist_ok=True)
    with open(os.path.join(dst, cfg, "pmc_%s_per_kernel.csv" % ctr), "w") as fh:        # (closed before the traffic table below reads it back)
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "launches", "mean_value_per_launch"])
        for (k, c), v in sorted(acc.items()):
            w.writerow([k, c, len(v), sum(v) / len(v)])
# profiles/pmc_traffic.json: per-launch HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE
# reports half the bytes of wide coalesced reads) per configuration -- inference configurations keyed by kernel symbol, train steps keyed by
# kernel FAMILY (the name before the template arguments: what bench.py's fd_trace aggregates by), launch-weighted.
traffic = {}
for cfg in ("infer", "train_f32", "train_bf16", "f16", "bf16", "pruned_f16"):
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(dst, cfg, "pmc_%s_per_kernel.csv" % ctr)
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            if not r["kernel"].startswith("fd_") or r["counter"] != ctr:
                continue
            name = r["kernel"].split("<")[0] if cfg.startswith("train") else r["kernel"]
            e = per.setdefault(name, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
            e[ctr][0] += float(r["mean_value_per_launch"]) * int(r["launches"]); e[ctr][1] += int(r["launches"])
    if per:
        traffic[cfg] = {k: {"bytes_per_launch": (2.0 * e["FETCH_SIZE"][0] / max(e["FETCH_SIZE"][1], 1) + e["WRITE_SIZE"][0] / max(e["WRITE_SIZE"][1], 1)) * 1024.0,
                            "launches_in_pmc_run": e["FETCH_SIZE"][1]} for k, e in sorted(per.items())}
if traffic:
    traffic["_meta"] = {"source": "round %s, gpurun_out/%s -> profiles/%s/<configuration>/pmc_{FETCH,WRITE}_SIZE_per_kernel.csv" % (rnd, tag, rnd)}
    json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1)
