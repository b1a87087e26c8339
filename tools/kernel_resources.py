#!/usr/bin/env python3
"""Register / spill / occupancy table of the library's kernels (hipcc -Rpass-analysis=kernel-resource-usage).
   python tools/kernel_resources.py [name filter]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "uf3_amd", "csrc", "uf3_hip.hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-shared",
       "-o", "/tmp/_res.so", src, "-Rpass-analysis=kernel-resource-usage"] + [a for a in sys.argv[1:] if a.startswith("-D")]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
flt = [a for a in sys.argv[1:] if not a.startswith("-D")]
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if flt and not all(f in r["name"] for f in flt):
        continue
    print(f'{r["name"][:70]:70s} VGPR {r.get("VGPRs", -1):4d} AGPR {r.get("AGPRs", -1):3d} SGPR {r.get("TotalSGPRs", -1):4d} '
          f'spillV {r.get("VGPRs Spill", -1):4d} spillS {r.get("SGPRs Spill", -1):4d} scratch {r.get("ScratchSize", -1):5d} occ {r.get("Occupancy", -1)}')
