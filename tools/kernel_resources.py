"""Compact per-kernel register / LDS / scratch / occupancy table from hipcc -Rpass-analysis=kernel-resource-usage."""
import re
import subprocess
import sys

srcs = sys.argv[1:] if len(sys.argv) > 1 else ["fast-depth_amd/csrc/" + f for f in ("fd_api.hip", "fd_train_fwd.hip", "fd_train_bwd.hip")]   # (the library's translation units)
procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Wno-unused-value",
                           "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"], stderr=subprocess.PIPE, text=True) for src in srcs]
out = "".join(p.communicate()[1] for p in procs)
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (?:\S+ )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
print("%-58s %5s %5s %5s %7s %4s %6s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r in rows:
    print("%-58s %5s %5s %5s %7s %4s %6s" % (r["name"][-58:], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"),
                                            r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
