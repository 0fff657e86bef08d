#!/bin/bash
# compile k_deform.hip to gfx950 assembly and print the instruction mix of k_deform_dw_bf's main loop
# (M = MFMA, . = VALU, G = global load, r / w = LDS read / write, | = s_waitcnt, B = s_barrier)
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc $(python -c "from gftorf_amd import build; print(' '.join(build.flags()+build.FILE_FLAGS.get('k_deform.hip',[])))") $EXTRA --cuda-device-only -S gftorf_amd/csrc/k_deform.hip -o /tmp/k_deform_dev.s || exit 1
awk '/^_ZN12_GLOBAL__N_114k_deform_dw_bfENS_6DwArgsE:/,/\.set _ZN12_GLOBAL__N_114k_deform_dw_bfENS_6DwArgsE.num_vgpr/' /tmp/k_deform_dev.s > /tmp/dw.s
grep "num_vgpr\|num_agpr" /tmp/dw.s
python3 - <<'PY'
import re
lines=open('/tmp/dw.s').read().split('\n')
# find the loop with the most mfma between label and backward branch
best=None
labels={}
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: labels[m.group(1)]=i
for i,l in enumerate(lines):
    m=re.match(r'\s+s_cbranch\w+\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        a=labels[m.group(1)]
        n=sum('mfma' in x for x in lines[a:i])
        if best is None or n>best[0]: best=(n,a,i)
n,a,b=best
print("loop lines",a,b,"mfma",n)
out=''
for l in lines[a:b+1]:
    t=l.split()
    if not t or t[0].startswith(';') or t[0].startswith('.'): continue
    op=t[0]
    if 'mfma' in op: c='M'
    elif op.startswith('global_load') or op.startswith('buffer_load'): c='G'
    elif op.startswith('ds_read'): c='r'
    elif op.startswith('ds_write'): c='w'
    elif op.startswith('s_waitcnt'):
        c='|' if 'vmcnt' not in l else '['+re.search(r'vmcnt\((\d+)\)',l).group(1)+']'
    elif op.startswith('s_barrier'): c='B\n'
    elif op.startswith('v_'): c='.'
    elif op.startswith('s_nop'): c='n'
    else: c=''
    out+=c
for seg in out.split('\n'):
    for i in range(0,len(seg),150): print(seg[i:i+150])
    print('--')
PY
