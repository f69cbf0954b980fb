#!/bin/bash
# Compiles two instantiations of the lane kernel (L5-type two-arm HALF and the CBOC derived-arm one) and prints the
# instruction mix of their loops.  usage: scripts/lane_probe.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")/../cu-sdr-collection_amd/csrc"
mkdir -p /tmp/lane_probe
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -ffp-contract=off -DGC_LANE_PROBE "$@" \
  -c corr_lane.hip -o /tmp/lane_probe/probe.o --save-temps=obj 2>&1 | grep -E "error|warning: v" || true
python3 - <<'PY'
import re
txt = open('/tmp/lane_probe/corr_lane-hip-amdgcn-amd-amdhsa-gfx950.s').read()
L = txt.split('\n')
starts = [(i, m.group(1)) for i, l in enumerate(L) for m in [re.match(r'^(_Z\S*lane_kernel\S*):', l)] if m]
for n, (st, name) in enumerate(starts):
    en = next(i for i in range(st, len(L)) if L[i].startswith('.Lfunc_end'))
    body = L[st:en]
    vg = re.search(re.escape(name) + r'\.num_vgpr, (\S+)', txt)
    sc = re.search(re.escape(name) + r'\.private_seg_size, (\S+)', txt)
    print(name[40:95], 'vgpr', vg and vg.group(1), 'scratch', sc and sc.group(1))
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    for a, b in loops:
        ins = [x.strip() for x in body[a:b + 1] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        fma = sum(1 for x in ins if re.match(r'v_(fma_f32|fmac_f32|mul_f32|add_f32|sub_f32|add_u32|sub_u32)', x))
        v = sum(1 for x in ins if x.startswith('v_'))
        mov = sum(1 for x in ins if x.startswith('v_mov') or x.startswith('v_accvgpr'))
        ds = sum(1 for x in ins if x.startswith('ds_'))
        if ds >= 2 and fma >= 20:
            print(f'  loop {a}-{b}: insts {len(ins)} valu {v} (full-rate {fma}, movs {mov}) lds {ds} vmem {sum(1 for x in ins if x.startswith(("global_", "buffer_")))} salu {sum(1 for x in ins if x.startswith("s_"))}')
PY
