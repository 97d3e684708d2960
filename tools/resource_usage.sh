#!/bin/bash
# prints per-kernel VGPR / SGPR / scratch / LDS / occupancy of libsphfluid's kernels (gfx950)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fvisibility=hidden \
  -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage \
  "$(dirname "$0")/../fluidsimulation_amd/csrc/sphfluid.hip" -o /tmp/_ru.so 2>&1 | python3 -c '
import re,sys
cur=None
for line in sys.stdin:
    m=re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur=t.split(":")[1].strip()[:40].ljust(42)
    elif any(t.startswith(k) for k in ("VGPRs:","TotalSGPRs:","ScratchSize","Occupancy","LDS Size","AGPRs:")):
        cur+=" "+t.replace(" [bytes/lane]","").replace(" [bytes/block]","").replace(" [waves/SIMD]","")
if cur: print(cur)
'
