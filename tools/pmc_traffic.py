"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_pass.sh) over tools/run_once.py
into profiles/<round>/pipeline_traffic.json: fabric bytes per configuration, per kernel.

FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports
half the bytes of coalesced streaming reads, so it is doubled; other access widths are uncalibrated,
which makes the fetch figure an upper estimate for the 4-byte gathers of the narrow-phase kernels.
Usage: python tools/pmc_traffic.py <fetch-dir> <write-dir> <configs-per-launch> <out.json>"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gear_hip" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].replace("void gear_hip::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def configs_per_launch(d):
    """One k_sweep_grid lane per configuration: its grid size is what one launch of every kernel covers.
    (The round-1 v7 figure divided per-launch counters of 128k-configuration launches by the 512k of the
    whole call and so understated the traffic four times: v7_pipeline_traffic_understated.json.)"""
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")[0]
    g = {int(r["Grid_Size"]) for r in csv.DictReader(open(f)) if "k_sweep" in r["Kernel_Name"]}
    return max(g) if g else None


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
n = int(sys.argv[3])
seen = configs_per_launch(sys.argv[1])
if seen and seen != n:
    print(f"configs per launch: {n} given, the sweep's grid says {seen}: using {seen}", file=sys.stderr)
    n = seen
out = {"configs_per_launch": n, "unit": "bytes per configuration", "kernels": {}, "corrections":
       "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of coalesced reads), WRITE_SIZE KiB x 1024"}
tot = 0.0
for k in sorted(set(fetch) | set(write)):
    fb, wb = fetch.get(k, 0.0) * 1024 * 2 / n, write.get(k, 0.0) * 1024 / n
    out["kernels"][k] = {"fetch": round(fb, 1), "write": round(wb, 1)}
    tot += fb + wb
out["total_bytes_per_config"] = round(tot, 1)
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
