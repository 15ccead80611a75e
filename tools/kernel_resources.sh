#!/bin/bash
# Per-kernel register / scratch / occupancy figures of one HIP source, from hipcc's own remarks (measurement tool):
#   tools/kernel_resources.sh openvoice_amd/csrc/conv1d_inst_f.hip
src=$1; dir=$(dirname "$src")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$(dirname "$0")/../include" -I"$dir" -Wno-unused-result -Wno-pass-failed \
  -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 | python3 -c "
import sys, re, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    for key in ('VGPRs', 'AGPRs', 'TotalSGPRs', 'ScratchSize \[bytes/lane\]', 'Occupancy \[waves/SIMD\]', 'LDS Size \[bytes/block\]'):
        m = re.search(r'\s' + key + r': (\d+)', line)
        if m and cur is not None:
            cur[key.split(' ')[0]] = int(m.group(1))
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'^void ', '', name).split('(')[0]
    print(f\"{r.get('VGPRs',0):4d} vgpr {r.get('AGPRs',0):3d} agpr {r.get('TotalSGPRs',0):4d} sgpr {r.get('ScratchSize',0):4d} scratch {r.get('Occupancy',0):2d} occ {r.get('LDS',0):6d} lds  {name}\")
"
