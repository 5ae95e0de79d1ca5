#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && rm -rf proft && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/proft -- python $GRAFT_REPO_ROOT/tools/bench_engine.py up --steps 12 --pipelined 1 > $GRAFT_REPO_ROOT/gpurun_out/r03_pipe_trace.log 2>&1
f=$(find /tmp/proft -name "*kernel_trace.csv" | head -1); wc -l $f
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last 4 graph-replayed steps: 290 kernels each
names = [r["Kernel_Name"] for r in rows]
last = rows[-290 * 4:]
print("queues used in the last steps:", sorted(set(r.get("Queue_Id", "?") for r in last)))
st = [int(r["Start_Timestamp"]) for r in last]; en = [int(r["End_Timestamp"]) for r in last]
byend = sorted(range(len(last)), key=lambda i: en[i])
import statistics
dur = [en[i] - st[i] for i in range(len(last))]
print("kernel duration: mean %.2f us" % (statistics.mean(dur) / 1e3))
gaps = [en[byend[i + 1]] - en[byend[i]] for i in range(len(byend) - 1)]
print("end-to-end spacing: mean %.2f us, median %.2f" % (statistics.mean(gaps) / 1e3, statistics.median(gaps) / 1e3))
# overlap: does kernel k start before the previous (by end order) kernel has ended?
ov = [en[byend[i]] - st[byend[i + 1]] for i in range(len(byend) - 1)]
print("start(next) before end(prev) by: mean %.2f us, median %.2f us, fraction > 0: %.2f" % (statistics.mean(ov) / 1e3, statistics.median(ov) / 1e3, sum(o > 0 for o in ov) / len(ov)))
for i in byend[300:312]:
    print(last[i]["Kernel_Name"][:50], last[i].get("Queue_Id"), (st[i] - st[byend[300]]) / 1e3, (en[i] - st[byend[300]]) / 1e3)
PY
