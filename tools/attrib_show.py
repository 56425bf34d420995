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
