"""Per-kernel LDS counters of a tools/pmc_lds.sh run: conflict share of the LDS-array cycles and LDS activity.  usage: python tools/summarize_lds_pmc.py <counter_collection.csv>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
print("%-48s %5s %14s %16s %18s %12s" % ("kernel", "calls", "LDS insts/call", "conflict/idx_act", "LDS active/SQ busy", "VALU insts"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"]):
    if not k.startswith("fd_"): continue
    n = cnt[k] or 1
    print("%-48s %5d %14.0f %16.3f %18.3f %12.0f" % (k[:48], n, v["SQ_INSTS_LDS"] / n, v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1),
                                                      v["SQ_ACTIVE_INST_LDS"] / max(v["SQ_BUSY_CYCLES"], 1), v["SQ_INSTS_VALU"] / n))
