"""Per-kernel register / LDS / occupancy table of libgear_hip's kernels (compiler remarks; no GPU needed).
Usage: python tools/kernel_resources.py [filter]"""
import os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", os.path.join(root, "gear_amd/csrc/gear_hip.hip")]
err = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r":\d+:\d+:\s+(\S.*?): (\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).replace("remark:", "").strip(), m.group(2)
    if k.endswith("Name"):
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'occ':>4s} {'LDS':>6s} {'sspill':>6s}")
for r in rows:
    nm = re.sub(r"\(.*", "", r["name"]).replace("void gear_hip::", "")
    if flt in nm:
        print(f"{nm[:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} "
              f"{r.get('ScratchSize [bytes/lane]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s} "
              f"{r.get('LDS Size [bytes/block]','?'):>6s} {r.get('SGPRs Spill','?'):>6s}")
