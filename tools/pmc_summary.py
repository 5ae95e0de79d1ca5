"""Summarise the two rocprofv3 --pmc passes over tools/pmc_target.py (FETCH_SIZE, WRITE_SIZE; one counter per pass) into
a json under profiles/.  Usage: python tools/pmc_summary.py <fetch_csv> <write_csv> <kernel substring> <out json> [--wide]
(--wide: the target ran with --wide; the c_attn launch streams W x (2S + W) weights and is credited with W x 4S)."""
import csv
import json
import sys

fetch_csv, write_csv, key, out = sys.argv[1:5]
wide = "--wide" in sys.argv


def mean_kb(path, counter):
    vals = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if key in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


f_kb, n = mean_kb(fetch_csv, "FETCH_SIZE")
w_kb, _ = mean_kb(write_csv, "WRITE_SIZE")
N, W, S = 16, 1920, 480
JC = 4 * S if wide else 3 * S
alg = int(0.5 * ((W * W + W * JC) * 2 + 2 * N * W * 2 + N * (W + JC) * 2))
traffic = int(round((2 * f_kb + w_kb) * 1024))
json.dump({
    "kernel": f"{key} (LayerNorm-folded projection), shapes K=1920 J={2 * S + W if wide else 3 * S} / J=1920 alternating, 16 rows, cold weights",
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python tools/pmc_target.py" + (" --wide" if wide else "") + " (one pass per counter)",
    "launches_per_counter": n,
    "FETCH_SIZE_KB_mean": f_kb,
    "WRITE_SIZE_KB_mean": w_kb,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); "
                  "WRITE_SIZE checked against the exact output size (N*J*2 B) and used as is",
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": round(traffic / alg, 4),
}, open(out, "w"), indent=1)
print(open(out).read())
