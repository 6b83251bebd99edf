"""Builds tests/rccl_stub/_build/librccl_stub.so (TEST INFRASTRUCTURE ONLY: the nccl* entry points over POSIX shared memory, for the emulator build)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "librccl_stub.so")


def build():
    import fcntl
    src = os.path.join(HERE, "rccl_stub.c")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(os.path.join(os.path.dirname(OUT), ".lock"), "w") as lk:      # parallel test workers: one builds, the others wait and reuse
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-Wall", "-o", OUT, src, "-lrt", "-pthread"])
    return OUT


if __name__ == "__main__":
    print(build())
