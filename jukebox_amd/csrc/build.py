"""Build libjukebox_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m jukebox_amd.csrc.build        # or: from jukebox_amd.csrc.build import build; build()

The shared object is written next to the sources (in-tree, git-ignored) so that it travels to the GPU
box with the repository snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "gemm.hip", "attention.hip", "elementwise.hip", "engine.hip"]
HEADERS = ["common.h", "gemm_glds_index.h", "gemm_8phase.h", os.path.join("..", "..", "include", "jukebox_hip.h")]
LIB = os.path.join(HERE, "libjukebox_hip.so")
LIB_SEGMENTS = os.path.join(HERE, "libjukebox_hip_segments.so")       # measurement build: -DJB_PIPE_SEGMENTS (common.h)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS + ["build.py"])


def build(force=False, verbose=False, segments=False):
    """segments: the measurement build (per-segment stamps of the pipelined launches) next to the product library."""
    lib = LIB_SEGMENTS if segments else LIB
    if not force and not _stale(lib):
        return lib
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", ".seg.o" if segments else ".o"))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall",
               "-Wno-unused-function"] + (["-DJB_PIPE_SEGMENTS"] if segments else []) + ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src} ---\n{out}\n")
        elif out.strip() and verbose:
            print(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, segments="--segments" in sys.argv))
