"""Builds oracle/_build/libfd_oracle.so from fd_oracle.c with gcc (test infrastructure only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fd_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libfd_oracle.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    import fcntl
    with open(os.path.join(OUT_DIR, ".lock"), "w") as lk:      # parallel test workers: one builds, the others wait and reuse
        fcntl.flock(lk, fcntl.LOCK_EX)
        return _build(force)


def _build(force):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=c99", "-Wall", SRC, "-o", LIB, "-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
