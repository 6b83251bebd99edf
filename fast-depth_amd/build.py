"""Builds fast-depth_amd/fastdepth_hip/libfastdepth_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

    python fast-depth_amd/build.py [--force]

The library is three translation units (inference; train plan + forward; train backward / loss / SGD / exchange) compiled IN PARALLEL into
fast-depth_amd/csrc/_obj/ and linked.  hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
OUT = os.path.join(HERE, "fastdepth_hip", "libfastdepth_hip.so")
SOURCES = [os.path.join(CSRC, f) for f in ("fd_api.hip", "fd_train_fwd.hip", "fd_train_bwd.hip")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-value", "-Wno-unused-function", "-DNDEBUG"]
FLAGS = CFLAGS + ["-shared"]          # (one-command form, kept for callers that build a single source)


def _deps():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f != "_obj") + [os.path.join(HERE, "..", "include", "fastdepth_hip.h")]


def source_hash(extra=()):
    """First 16 hex digits of the SHA-256 over the library's sources (file names + contents, sorted) AND the compiler flags (CFLAGS + `extra`:
    a build with other -D switches is another library -- ADVICE r05): compiled into the binary as FD_SOURCE_HASH and reported by fd_version(),
    so that whoever loads the .so can tell whether it was built from the sources next to it."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(CFLAGS + list(extra)).encode() + b"\0")
    for d in _deps():
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def built_hash():
    """The FD_SOURCE_HASH stamped into the library that is on disk now ("" if there is none / it is unstamped)."""
    import re
    try:
        with open(OUT, "rb") as f:
            m = re.search(rb"train step f32/bf16; sources ([0-9a-f]{16})\)", f.read())
        return m.group(1).decode() if m else ""
    except OSError:
        return ""


def _stale(extra=()):
    if not os.path.exists(OUT):
        return True
    return built_hash() != source_hash(extra)       # content, not mtimes: a checkout or a copy must neither force nor hide a rebuild


def compile_and_link(out, extra=(), tag=""):
    """Every source -> its object (all at once), then one link.  `tag` keeps the objects of library variants (tools/build_variant.py) apart."""
    os.makedirs(OBJ, exist_ok=True)
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + tag + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([HIPCC] + CFLAGS + ['-DFD_SOURCE_HASH="%s"' % source_hash(extra)] + list(extra) + ["-c", src, "-o", obj])))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    return out


LAST_BUILD_MODE = ""


def build(force=False, extra=()):
    """Returns the library path; LAST_BUILD_MODE says what happened ("compiled" / "reused: stamp == source hash")."""
    global LAST_BUILD_MODE
    import fcntl
    with open(os.path.join(os.path.dirname(OUT), ".build.lock"), "w") as lk:      # concurrent callers (parallel test workers): one compiles, the others wait and reuse
        fcntl.flock(lk, fcntl.LOCK_EX)
        return _build_locked(force, extra)


def _build_locked(force, extra):
    global LAST_BUILD_MODE
    if not force and not _stale(extra):
        LAST_BUILD_MODE = "reused (stamp %s == hash of the sources + flags in the tree)" % built_hash()
        return OUT
    LAST_BUILD_MODE = "compiled (sources + flags %s)" % source_hash(extra)
    return compile_and_link(OUT, extra)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
