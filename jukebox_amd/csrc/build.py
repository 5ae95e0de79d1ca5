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


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS + ["build.py"])


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall",
               "-Wno-unused-function", "-c", os.path.join(HERE, src), "-o", obj]
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
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
