"""Prints the interesting parts of a bench.py JSON line (tools/gpu_round.sh)."""
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, "roofline", {k: d["roofline"][k] for k in ("kernel", "bound", "achieved", "frac", "avg_launch_us")})
    print("whole_step", d["whole_step"])
    for k in d["kernels"]:
        print("  ", k)
    for t in ("train_step", "train_step_bf16"):
        r = d.get(t) or {}
        print(t, {k: r.get(k) for k in ("value", "ms_per_step", "error")}, (r.get("roofline") or {}).get("kernel"), (r.get("roofline") or {}).get("frac"), r.get("whole_step"))
        for k in (r.get("kernels") or []):
            print("  ", k)
    for e in d.get("other_configs", []):
        print(e["config"][:60], e["value"], e["ms_per_step"], (e.get("roofline") or {}).get("kernel"), (e.get("roofline") or {}).get("frac"), (e.get("whole_step") or {}).get("frac_of_roofline"))
    c = d.get("cpu_baseline", {})
    print("cpu", c.get("kind"), c.get("value"), c.get("batch1_ms"), c.get("batch32_fps"), (c.get("train_step") or {}).get("value"), "|", c.get("sample"))
    print("train_check", d.get("train_check"))
except Exception as e:
    print("bench parse failed", repr(e))
