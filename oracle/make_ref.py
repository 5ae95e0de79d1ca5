"""Recipe for `oracle/_ref/`: a read-only snapshot of the reference's Python package, so that the UNMODIFIED reference
can be timed on the GPU box's host cores (bench.py's `cpu_baseline`, kind "reference") where `/root/reference` does not
exist.  Test infrastructure only -- nothing under `jukebox_amd/` imports it.

    python oracle/make_ref.py            # build container only (needs /root/reference); idempotent

`oracle/_ref/` is git-ignored (reference sources never enter this repository's history) but not gpurun-ignored, so it
travels to the GPU box with the working tree exactly like the built `.so` files.  Only `jukebox/**/*.py` and the
`jukebox/data/ids/*.txt` id tables are snapshotted (no apex / tensorboardX / lyric sheets); the import shim that lets it run on a CPU is
`tests/golden/refshim.py`.
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("JUKEBOX_REFERENCE", "/root/reference")
DST = os.path.join(HERE, "_ref")


def make(verbose=True):
    src_pkg = os.path.join(SRC, "jukebox")
    if not os.path.isdir(src_pkg):
        if verbose:
            print(f"oracle/make_ref: no reference tree at {SRC}; keeping {DST} as it is")
        return os.path.isdir(os.path.join(DST, "jukebox"))
    n = 0
    for root, dirs, files in os.walk(src_pkg):
        dirs[:] = [d for d in dirs if d not in ("__pycache__", "tests")]
        rel = os.path.relpath(root, SRC)
        for f in files:
            if not (f.endswith(".py") or (f.endswith(".txt") and os.path.basename(root) == "ids")):
                continue                                   # sources + the artist / genre id tables the Labeller opens
            out_dir = os.path.join(DST, rel)
            os.makedirs(out_dir, exist_ok=True)
            s, d = os.path.join(root, f), os.path.join(out_dir, f)
            if not os.path.exists(d) or os.path.getmtime(d) < os.path.getmtime(s) or os.path.getsize(d) != os.path.getsize(s):
                if os.path.exists(d):
                    os.remove(d)
                shutil.copyfile(s, d)
                os.chmod(d, 0o444)
            n += 1
    with open(os.path.join(DST, "README"), "w") as fh:
        fh.write("Snapshot of /root/reference/jukebox/**/*.py made by oracle/make_ref.py (git-ignored; CPU timing baseline only).\n")
    if verbose:
        print(f"oracle/make_ref: {n} files under {DST}")
    return True


if __name__ == "__main__":
    sys.exit(0 if make() else 1)
