#!/usr/bin/env python3
"""Print a per-kernel resource table (VGPR/AGPR/scratch/LDS/occupancy) for one .hip file, from
hipcc -Rpass-analysis=kernel-resource-usage.  Usage: tools/kres.py csrc/gf_spmm.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{root}/include",
       f"-I{root}/graph-neural-networks_amd/csrc", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem); dem = re.sub(r"\(.*", "", dem); dem = dem.replace("void ", "")
        cur = {"name": dem}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print(f"{'kernel':48s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    if filt and filt not in r["name"]: continue
    print(f"{r['name'][:48]:48s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('SGPRs','?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('LDS Size [bytes/block]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
