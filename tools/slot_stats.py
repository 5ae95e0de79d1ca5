"""Per-slot kernel durations of the decode step from a rocprofv3 kernel trace.
Usage: python tools/slot_stats.py <dir with *_kernel_trace.csv> [launches_per_step]
A step is embed, 72 x (c_attn, attention, c_proj, c_fc, c_proj2), final_add, logits, sample, inc -- found by the embed
kernel's name; durations and the gap to the previous kernel's end are averaged per slot over the steps of the trace
(second half only: graph replay in tools/bench_engine.py)."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 365
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "embed_kernel" in r[2]]
steps = [i for i in starts if i + per_step <= len(rows) and "inc_int" in rows[i + per_step - 1][2]]
steps = steps[len(steps) // 2:]
print(f"{len(rows)} dispatches, {len(starts)} embed launches, {len(steps)} complete steps used")
dur, gap, name = defaultdict(list), defaultdict(list), {}
for s in steps:
    for k in range(per_step):
        st, en, nm = rows[s + k]
        slot = ("layer", (k - 1) % 5) if 1 <= k <= per_step - 5 else ("edge", k if k == 0 else k - per_step)
        dur[slot].append(en - st)
        gap[slot].append(st - rows[s + k - 1][1])
        name[slot] = nm[:70]
tot = 0.0
for slot in sorted(dur):
    n = len(dur[slot]) / len(steps)
    a, g = sum(dur[slot]) / len(dur[slot]), sum(gap[slot]) / len(gap[slot])
    tot += n * (a + g)
    print(f"{slot}: x{n:.0f}/step dur {a / 1e3:.2f} us  gap-before {g / 1e3:.2f} us  min {min(dur[slot]) / 1e3:.2f}  {name[slot]}")
print(f"sum over a step: {tot / 1e6:.3f} ms")
# early vs late layers for the two LayerNorm projections
for which in (0, 3):
    for lo, hi in ((0, 8), (32, 40), (64, 72)):
        v = [rows[s + 1 + 5 * l + which][1] - rows[s + 1 + 5 * l + which][0] for s in steps for l in range(lo, hi)]
        print(f"slot {which} layers {lo}-{hi}: {sum(v) / len(v) / 1e3:.2f} us")
