#!/bin/bash
# dev helper: per-kernel register / scratch / occupancy report from hipcc (no GPU needed)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-result -Wno-unused-function -I include d3d12renderer_amd/csrc/world.hip -o /tmp/probe.so -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
out=[]
for line in sys.stdin:
    m=re.search(r"remark: +Function Name: (\S+)",line)
    if m: out.append([m.group(1)]); continue
    m=re.search(r"remark: +(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)",line)
    if m and out: out[-1].append(m.group(1).split()[0]+"="+m.group(2))
for o in out: print(" ".join(o))'
