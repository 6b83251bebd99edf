"""Builds a variant of the HIP library with extra compiler switches into scratch/variants/<name>.so (tuning A/B aid; the product library is
fast-depth_amd/build.py's).  usage: python tools/build_variant.py <name> -DFD_H16_STAGES=2 ...
Measurement tools take it with --lib scratch/variants/<name>.so."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "fast-depth_amd"))
import build as fd_build
out = os.path.join(REPO, "scratch", "variants", sys.argv[1] + ".so")
os.makedirs(os.path.dirname(out), exist_ok=True)
if sys.argv[1].startswith("-"):
    raise SystemExit(__doc__)
print(fd_build.compile_and_link(out, sys.argv[2:], tag="_" + sys.argv[1]))
# the variant's object files are not needed once it is linked (they travel to the GPU box with every gpurun snapshot: 12 MB per variant)
import glob
for o in glob.glob(os.path.join(REPO, "fast-depth_amd", "csrc", "_obj", "*_" + sys.argv[1] + ".o")):
    os.remove(o)
