"""Per-slot kernel durations of the decode step from a rocprofv3 kernel trace.
Usage: python tools/slot_stats.py <dir with *_kernel_trace.csv> [n_layers] [launches_per_layer]
A step is n_layers x (c_attn, attention, c_proj, c_fc, c_proj2), logits, sample -- 5 L + 2 dispatches ending with
sample_kernel (launches_per_layer = 4 for wide-value engines: c_attn, attention, c_fc, c_proj2); durations and the gap to the previous kernel's end are averaged per slot over the complete steps of the
trace (second half only: graph replay in tools/bench_engine.py)."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 72
PL = int(sys.argv[3]) if len(sys.argv) > 3 else 5
per_step = PL * L + 2
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, r in enumerate(rows) if "sample_kernel" in r[2]]
steps = [e - per_step + 1 for p, e in zip(ends, ends[1:]) if e - p == per_step]
steps = steps[len(steps) // 2:]
print(f"{len(rows)} dispatches, {len(ends)} sampler launches, {len(steps)} complete steps used")
dur, gap, name = defaultdict(list), defaultdict(list), {}
for s in steps:
    for k in range(per_step):
        st, en, nm = rows[s + k]
        slot = ("layer", k % PL) if k < PL * L else ("tail", k - PL * L)
        dur[slot].append(en - st)
        gap[slot].append(st - rows[s + k - 1][1])
        name[slot] = nm[:80]
tot = 0.0
for slot in sorted(dur):
    n = len(dur[slot]) / len(steps)
    a, g = sum(dur[slot]) / len(dur[slot]), sum(gap[slot]) / len(gap[slot])
    tot += n * (a + g)
    print(f"{slot}: x{n:.0f}/step dur {a / 1e3:.2f} us  gap-before {g / 1e3:.2f} us  min {min(dur[slot]) / 1e3:.2f}  {name[slot]}")
print(f"sum over a step: {tot / 1e6:.3f} ms")
# per pattern (attn_order 2: layer l uses block / transpose / prev for l % 3 = 0 / 1 / 2) for the attention and c_proj slots
for which in ((1, 2) if PL == 5 else (1,)):
    for pat in range(3):
        v = [rows[s + PL * l + which][1] - rows[s + PL * l + which][0] for s in steps for l in range(pat, L, 3)]
        print(f"slot {which} layers l%3=={pat}: {sum(v) / len(v) / 1e3:.2f} us")
