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
    import fcntl
    with open(os.path.join(OUT_DIR, ".lock"), "w") as lk:      # parallel test workers (pytest -n): one builds, the others wait and reuse
        fcntl.flock(lk, fcntl.LOCK_EX)
        return _build(force)


def _build(force):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f != "_obj"] + [os.path.join(HERE, "hipemu.h"),
                                                                os.path.join(REPO, "include", "fastdepth_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    # the library's three translation units, compiled in parallel and linked (as fast-depth_amd/build.py does with hipcc)
    flags = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-DFD_EMU", "-I", HERE, "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-psabi",
             "-Wno-comment", "-mavx2"]
    objs, procs = [], []
    for name in ("fd_api", "fd_train_fwd", "fd_train_bwd"):
        obj = os.path.join(OUT_DIR, name + ".o")
        objs.append(obj)
        procs.append(subprocess.Popen([CLANG] + flags + ["-c", os.path.join(CSRC, name + ".hip"), "-o", obj]))
    if any(p.wait() != 0 for p in procs):
        raise subprocess.CalledProcessError(1, "clang++ -DFD_EMU -c")
    subprocess.check_call([CLANG, "-shared", "-fPIC"] + objs + ["-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force=True))
