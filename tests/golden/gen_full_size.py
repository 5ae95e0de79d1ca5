"""Digests of the numpy oracle's prefill for the full-size parity cases (tests/full_size_cases.py) -- run HERE, on the CPU:

    python tests/golden/gen_full_size.py [tag ...]        # all cases: ~5 minutes on 8 cores

For each case the oracle (oracle.transformer.Transformer, itself pinned to the unmodified reference's outputs by
tests/test_oracle_golden.py) prefills sample 0 over the case's whole prefix -- 384 ... 8064 positions, every layer -- on the
seeded inputs the GPU test rebuilds bit for bit, and tests/golden/full_size_<tag>.npz keeps per layer the k / v rows at a dozen
positions in full and |k|^2 + |v|^2 of every row.  The GPU test (tests/test_hip_baseline_configs.py::_full_size_case) holds the
HIP engine's caches to that; what used to be minutes of CPU oracle inside the GPU lease is a file read."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import full_size_cases as FS  # noqa: E402


def main():
    tags = sys.argv[1:] or list(FS.CASES)
    for tag in tags:
        case = FS.CASES[tag]
        t = time.time()
        kv = FS.oracle_prefill(case, progress=lambda n: print(f"  {tag}: {n} / {case['t0']} positions, {time.time() - t:.0f} s", flush=True))
        dg = FS.digest(case, kv)
        np.savez(FS.golden_path(tag), **dg)
        size = os.path.getsize(FS.golden_path(tag)) / 1e6
        print(f"{tag}: {len(dg['pos'])} positions {dg['pos'].tolist()}, rows per layer {sorted(set(dg['n_rows'].tolist()))}, "
              f"{size:.1f} MB, {time.time() - t:.0f} s", flush=True)


if __name__ == "__main__":
    main()
