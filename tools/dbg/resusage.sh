#!/bin/bash
# register / scratch / occupancy per kernel of the product build: tools/dbg/resusage.sh [filter]
cd "$(dirname "$0")/../../oddio_amd/csrc" && mkdir -p /tmp/oddio_asm && make -s asm 2>&1 | python3 -c "
import sys, re, subprocess
cur = None; rows = {}
for ln in sys.stdin:
    m = re.search(r'Function Name: (\S+)', ln)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark: [^:]*:\d+:\d+:\s+(\w[\w /\[\]]*): (\d+)', ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()[:90]
    if flt not in name:
        continue
    print('%-90s vgpr %3d agpr %3d scratch %4d occ %d lds %6d' % (name, v.get('VGPRs', -1), v.get('AGPRs', -1), v.get('ScratchSize [bytes/lane]', -1), v.get('Occupancy [waves/SIMD]', -1), v.get('LDS Size [bytes/block]', -1)))
" "$1"
