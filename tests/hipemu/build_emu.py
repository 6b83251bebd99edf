"""Builds tests/hipemu/_build/libfastdepth_emu.so: the product's kernel + plan sources compiled for the
CPU emulator in hipemu.h (clang++ -DFD_EMU).  TEST INFRASTRUCTURE ONLY."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "fast-depth_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libfastdepth_emu.so")
CLANG = os.environ.get("FD_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "hipemu.h"),
                                                                os.path.join(REPO, "include", "fastdepth_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-DFD_EMU", "-I", HERE, "-Wall",
           "-Wno-unused-function", "-Wno-unused-variable", "-Wno-psabi", "-Wno-comment", "-mavx2", os.path.join(CSRC, "fd_api.hip"), "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
