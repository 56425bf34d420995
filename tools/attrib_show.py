"""Turns the passes of tools/attrib_sweep.sh into a per-region table of the sweep kernel (instructions per
wavefront of 64 configurations).  Usage: python tools/attrib_show.py <tag> [kernel-substring]"""
import collections, csv, glob, json, sys
tag = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_sweep"
rows = {}
for d in sorted(glob.glob(f"gpurun_out/{tag}/*/")):
    label = d.rstrip("/").split("/")[-1]
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if agg:
        m = {c: sum(v) / len(v) for c, v in agg.items()}
        w = m["SQ_WAVES"]
        rows[label] = {c: round(m[c] / w, 1) for c in m if c != "SQ_WAVES"}
        rows[label]["waves"] = int(w)
print(json.dumps(rows, indent=1))
def d(a, b, c="SQ_INSTS_VALU"):
    return round(rows[a][c] - rows[b][c], 1) if a in rows and b in rows else None
att = {
    "total VALU per wavefront": rows.get("full", {}).get("SQ_INSTS_VALU"),
    "stage 1 of the narrow phase inside the drains (full - no_stage1)": d("full", "no_stage1"),
    "ring pushes + drain bookkeeping (no_stage1 - no_push)": d("no_stage1", "no_push"),
    "oriented-box second cull, per-hit passes (no_push - no_push_obb)": d("no_push", "no_push_obb"),
    "cell lookup + first cull + block loop (no_push_obb - fk_self)": d("no_push_obb", "fk_self"),
    "self pairs: centre cull (fk_self - fk_only)": d("fk_self", "fk_only"),
    "limits + FK walk + shape poses (fk_only)": rows.get("fk_only", {}).get("SQ_INSTS_VALU"),
}
print(json.dumps(att, indent=1))
# issue-side summary of the unablated kernel.  SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles (4 clocks);
# a SIMD retires one wave64 VALU instruction per 2 clocks at its peak (157.3 TFLOP/s = 256 CU x 4 SIMD x 64 lanes x 2 flop
# / 2 clocks x 2.4 GHz), one wave alone issues one per 4 clocks.
if "full" in rows and len(sys.argv) > 3:
    f, b = rows["full"], rows.get("busy", {})
    waves_per_simd = int(sys.argv[3])
    cyc = 4.0 * f["SQ_WAVE_CYCLES"]
    issue = {
        "kernel": kern, "waves_per_simd": waves_per_simd,
        "valu_insts_per_wave": f["SQ_INSTS_VALU"], "salu_insts_per_wave": f["SQ_INSTS_SALU"],
        "lds_insts_per_wave": f["SQ_INSTS_LDS"], "smem_insts_per_wave": f["SQ_INSTS_SMEM"],
        "wave_cycles": cyc,
        "valu_issue_frac": round(waves_per_simd * f["SQ_INSTS_VALU"] * 2.0 / cyc, 4),
        "per_wave_time_split": {
            "issuing (SQ_ACTIVE_INST_ANY)": round(b.get("SQ_ACTIVE_INST_ANY", 0) / f["SQ_WAVE_CYCLES"], 3),
            "of which VALU (SQ_ACTIVE_INST_VALU)": round(f["SQ_ACTIVE_INST_VALU"] / f["SQ_WAVE_CYCLES"], 3),
            "of which scalar (SQ_ACTIVE_INST_SCA)": round(b.get("SQ_ACTIVE_INST_SCA", 0) / f["SQ_WAVE_CYCLES"], 3),
            "parked on s_waitcnt (SQ_WAIT_ANY)": round(f["SQ_WAIT_ANY"] / f["SQ_WAVE_CYCLES"], 3),
            "issue-stalled (SQ_WAIT_INST_ANY)": round(b.get("SQ_WAIT_INST_ANY", 0) / f["SQ_WAVE_CYCLES"], 3)},
        "source": "rocprofv3 --pmc passes of tools/attrib_sweep.sh over tools/run_once.py (512k configurations, metric scene, f32, one stream)",
    }
    print(json.dumps({"issue": issue}, indent=1))
    json.dump({"counters_per_wavefront": rows, "valu_attribution_per_wavefront": att, "issue": issue},
              open(sys.argv[4], "w"), indent=1)
