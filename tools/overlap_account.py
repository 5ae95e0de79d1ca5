"""Per-kernel account of the phase in which several levels decode side by side (missing 5): reads the kernel trace
of a short real job -- rocprofv3 --kernel-trace --output-format csv -- python bench.py --seconds 2 --steps 1 --no-cpu-baseline
with JB_PIPELINE_LAUNCHES=0 (plain chains throughout: what the overlap phase of the real job runs, and the profiler serialises
nothing that was not serial) -- and prints, per kernel, calls and average duration while ONE hardware queue was busy and while
SEVERAL were (10-ms bins of the job's time line), plus how long each state lasted.

Usage: python tools/overlap_account.py <kernel_trace.csv> [out.csv]"""
import collections
import csv
import re
import sys

path = sys.argv[1]
rows = []
with open(path, newline="") as f:
    r = csv.DictReader(f)
    cols = {c.lower(): c for c in r.fieldnames}
    name_c = cols.get("kernel_name") or cols.get("name")
    q_c = cols.get("queue_id") or cols.get("queue")
    s_c, e_c = cols.get("start_timestamp") or cols.get("start"), cols.get("end_timestamp") or cols.get("end")
    for row in r:
        rows.append((int(row[s_c]), int(row[e_c]), row[q_c], row[name_c]))
rows.sort()
t0, t1 = rows[0][0], max(x[1] for x in rows)
BIN = 10_000_000                                      # 10 ms
busy = collections.defaultdict(lambda: collections.Counter())
for s, e, q, _ in rows:
    busy[(s - t0) // BIN][q] += e - s
# a queue counts as busy in a bin when its kernels fill >= 20 % of it (a decode chain fills ~90 %)
state = {b: sum(1 for q, ns in c.items() if ns >= 0.2 * BIN) for b, c in busy.items()}
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:70]
acc = collections.defaultdict(lambda: [[0, 0], [0, 0]])        # name -> [alone (calls, ns), side by side (calls, ns)]
for s, e, q, n in rows:
    k = 1 if state.get((s - t0) // BIN, 0) >= 2 else 0
    a = acc[short(n)][k]
    a[0] += 1
    a[1] += e - s
dur = collections.Counter(min(v, 3) for v in state.values())
print(f"job {1e-9 * (t1 - t0):.1f} s, {len(rows)} dispatches; seconds with 1 / 2 / >= 3 queues busy: "
      f"{dur[1] * BIN * 1e-9:.1f} / {dur[2] * BIN * 1e-9:.1f} / {dur[3] * BIN * 1e-9:.1f}")
out = [("kernel", "calls_alone", "avg_ns_alone", "calls_side_by_side", "avg_ns_side_by_side", "ratio")]
for n, (a, b) in sorted(acc.items(), key=lambda kv: -(kv[1][0][1] + kv[1][1][1]))[:24]:
    av_a = a[1] / a[0] if a[0] else 0.0
    av_b = b[1] / b[0] if b[0] else 0.0
    out.append((n, a[0], round(av_a), b[0], round(av_b), round(av_b / av_a, 3) if av_a and av_b else ""))
for line in out:
    print(",".join(str(x) for x in line))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w", newline="") as f:
        csv.writer(f).writerows(out)
