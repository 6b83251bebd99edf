"""Builds fast-depth_amd/fastdepth_hip/libfastdepth_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

    python fast-depth_amd/build.py [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "fastdepth_hip", "libfastdepth_hip.so")
SOURCES = [os.path.join(CSRC, "fd_api.hip")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-value",
         "-Wno-unused-function", "-DNDEBUG"]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "fastdepth_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, extra=()):
    if not force and not _stale():
        return OUT
    cmd = [HIPCC] + FLAGS + list(extra) + SOURCES + ["-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
