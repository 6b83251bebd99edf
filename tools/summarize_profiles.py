"""Condenses a tools/gpu_round.sh output directory into the per-configuration, per-kernel summaries that get committed under profiles/:
kernel-trace stats (calls, avg / total duration) of each configuration's own rocprofv3 run, and per-launch HBM traffic from the
FETCH_SIZE / WRITE_SIZE PMC passes.  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their
bytes (MI355X_MICROARCH.md, HBM section): traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fd_demangle import demangle
import collections, csv, glob, json, os, sys

out = sys.argv[1]
CONFIGS = ("infer", "train_f32", "train_bf16", "f16", "bf16", "pruned_f16")


def find(pattern):
    m = glob.glob(os.path.join(out, pattern), recursive=True)
    return m[0] if m else None


def short(name):
    return demangle(name)                                 # (rocprofv3 leaves _Float16 instantiations mangled)


summary = {"kernel_stats": {}, "traffic_per_launch_bytes": {}, "mfma": {}}
for cfg in CONFIGS:
    stats = find("prof_%s/**/*kernel_stats.csv" % cfg)
    if not stats:
        continue
    rows = [{"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
             "pct": float(r["Percentage"])} for r in csv.DictReader(open(stats))]
    summary["kernel_stats"][cfg] = rows
    print("== %s (bench.py --only %s --steps 20 --warmup 3: 23 steps)" % (cfg, cfg))
    print("%-52s %6s %10s %10s %6s" % ("kernel", "calls", "avg us", "total ms", "%"))
    for k in rows[:14]:
        print("%-52s %6d %10.2f %10.3f %6.2f" % (k["kernel"][:52], k["calls"], k["avg_us"], k["total_ms"], k["pct"]))
    print()
for cfg in ("infer", "train_bf16", "train_f32", "f16", "pruned_f16"):
    pmc = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        f = find("pmc_%s_%s/**/*counter_collection.csv" % (cfg, counter))
        if not f:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            pmc.setdefault(k, {})[counter] = sum(v) / len(v)
    if not pmc:
        continue
    print("== HBM traffic per launch, %s" % cfg)
    print("%-52s %14s %14s %16s" % ("kernel", "FETCH_SIZE KiB", "WRITE_SIZE KiB", "HBM bytes/launch"))
    summary["traffic_per_launch_bytes"][cfg] = {}
    for k, d in sorted(pmc.items()):
        fetch, write = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
        total = (2.0 * fetch + write) * 1024.0
        summary["traffic_per_launch_bytes"][cfg][k] = total
        if k.startswith("fd_"):
            print("%-52s %14.0f %14.0f %16.0f" % (k[:52], fetch, write, total))
    print()
f = find("pmc_infer_SQ/**/*counter_collection.csv")
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== matrix-pipe utilisation of the pointwise GEMMs, infer (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs))")
    for k, d in sorted(acc.items()):
        if "gemm" not in k:
            continue
        m = {c: sum(v) / len(v) for c, v in d.items()}
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(m.get("GRBM_GUI_ACTIVE", 1.0) / 8.0 * 1024.0, 1.0)
        summary["mfma"][k] = {"mfma_busy_fraction": busy, **m}
        print("%-52s mfma busy %.3f  launches %d  %s" % (k[:52], busy, len(d.get("GRBM_GUI_ACTIVE", [])), {c: round(v) for c, v in m.items()}))
json.dump(summary, open(os.path.join(out, "profile_summary.json"), "w"), indent=1)
