"""Per-kernel VALU / LDS / wait accounting of a tools/pmc_valu.sh run.  usage: python tools/summarize_valu_pmc.py <counter_collection.csv> <kernel_trace.csv>
Columns: mean duration; VALU instructions per wave64 ... ; share of the kernel's SIMD time (1024 SIMDs x duration x 2.4 GHz) in which a VALU / LDS
instruction was being issued (SQ_ACTIVE_INST_* tick in quad-cycles); share of the resident waves' cycles spent parked (SQ_WAIT_ANY) or issue-stalled."""
import collections, csv, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fd_demangle import demangle
cc, kt = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    k = demangle(r["Kernel_Name"])
    dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(cc)):
    k = demangle(r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
print("%-52s %5s %8s %12s %9s %9s %9s %9s" % ("kernel", "calls", "us", "VALU inst", "VALU busy", "LDS busy", "parked", "stalled"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    if not k.startswith("fd_"): continue
    n = cnt[k] or 1
    d = sum(dur[k]) / max(len(dur[k]), 1)                      # ns (under the counter run: inflated by the PMC set-up, ratios are what matters)
    simd_cycles = d * 2.4 * 1024
    print("%-52s %5d %8.1f %12.0f %9.3f %9.3f %9.3f %9.3f" % (k[:52], n, d / 1e3, v["SQ_INSTS_VALU"] / n, 4 * v["SQ_ACTIVE_INST_VALU"] / n / max(simd_cycles, 1),
          4 * v["SQ_ACTIVE_INST_LDS"] / n / max(simd_cycles, 1), v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), v["SQ_WAIT_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1)))
