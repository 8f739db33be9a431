#!/bin/bash
# register / spill / LDS / occupancy summary per kernel of one csrc file: tools/kres.sh nerf_kernels.hip [name filter]
cd "$(dirname "$0")/../surface-irradiance-estimation-from-neural-radiance-fields_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -c "$1" -o /tmp/kres.$$.o -Rpass-analysis=kernel-resource-usage ${KRES_FLAGS} 2>&1 |
  python3 -c "
import re, sys
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r'remark: [^:]+:\d+:\d+: +([A-Za-z ]+[A-Za-z\]\[/]*): *(\S+)', line) or re.search(r': +(Function Name|Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|VGPR Spill|SGPR Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)', line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k in ('Function Name', 'Name'):
        cur = {'name': v}; rows.append(cur)
    else: cur[k] = v
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for r in rows:
    if flt in r['name']:
        print('%-90s VGPR %4s AGPR %3s SGPR %3s scratch %4s vspill %3s sspill %3s occ %s LDS %s' % (r['name'][:90], r.get('VGPRs'), r.get('AGPRs'), r.get('SGPRs'), r.get('ScratchSize [bytes/lane]'), r.get('VGPR Spill', r.get('VGPRs Spill')), r.get('SGPR Spill', r.get('SGPRs Spill')), r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
" "$2"
rm -f /tmp/kres.$$.o
