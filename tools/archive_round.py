"""Copies the judged summaries of a tools/gpu_round.sh visit from gpurun_out/<tag>/ into profiles/<round>/ (tracked)."""
import collections, csv, glob, json, os, shutil, sys
tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join("gpurun_out", tag), os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
for name, out in (("prof/trace_kernel_stats.csv", "rocprofv3_kernel_stats.csv"), ("profile_summary.txt", "profile_summary.txt"), ("profile_summary.json", "profile_summary.json"),
                  ("bench.json", "bench_n1.json"), ("pytest_gpu.log", "pytest_gpu.log"), ("smoke.log", "smoke.log")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, out))
for t in ("pmc_fetch", "pmc_write"):
    m = glob.glob(os.path.join(src, t, "**", "*counter_collection.csv"), recursive=True)
    if not m:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(m[0])):
        acc[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    w = csv.writer(open(os.path.join(dst, t + "_per_kernel.csv"), "w"))
    w.writerow(["kernel", "counter", "launches", "mean_value_per_launch"])
    for (k, c), v in sorted(acc.items()):
        w.writerow([k, c, len(v), sum(v) / len(v)])
s = json.load(open(os.path.join(src, "profile_summary.json")))
json.dump({k: v for k, v in s["traffic_per_launch_bytes"].items() if k.startswith("fd_")}, open("profiles/pmc_traffic.json", "w"), indent=1)
d = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d.get("train_step", {}).get("value"), d.get("cpu_baseline", {}).get("value"))
