#!/usr/bin/env python3
"""Resource usage (VGPRs, spills, occupancy, LDS) of the kernels of one csrc file, from hipcc's remarks.
usage: python tools/kres.py project.hip [name-filter] [extra hipcc flags...]"""
import os, re, subprocess, sys
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize",
       "-Rpass-analysis=kernel-resource-usage", *extra, "-c", os.path.join(here, "edgegaussians_amd", "csrc", src), "-o", "/tmp/_kres.o"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(\w[\w ]*\w)\s*(?:\[bytes/lane\]|\[waves/SIMD\]|\[bytes/block\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
if "error" in err and not rows:
    print(err)
for k, v in rows.items():
    if flt in k:
        print(f"{k[:70]:70s} VGPR {v.get('VGPRs')} AGPR {v.get('AGPRs')} spill {v.get('VGPRs Spill')} scratch {v.get('ScratchSize')} occ {v.get('Occupancy')} LDS {v.get('LDS Size')} SGPR {v.get('TotalSGPRs')}")
