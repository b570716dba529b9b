"""Instruction mix of a kernel's K loop from its assembly (no GPU): per interval between two s_barrier instructions that
contains MFMAs, the number of MFMA / LDS / LDS-DMA / other vector-memory / VALU / SALU / s_waitcnt instructions.
    hipcc --offload-arch=gfx950 -O3 -std=c++20 -S --cuda-device-only -Iinclude gan-aug-pfa_amd/csrc/conv_patch2.hip -o /tmp/k.s
    python tools/issue_mix.py /tmp/k.s k_conv_patch2ILi1ELi128ELi0ELi0
(what showed that the patch kernels' loops were bound by their own instruction stream: ~200 instructions per tap step for
16 MFMAs, two thirds of them the address arithmetic of 5 LDS-DMA pieces - profiles/r04_issue_mix.txt)"""
import re,sys
from collections import Counter
src=sys.argv[1]; pat=sys.argv[2]
lines=[l.rstrip() for l in open(src)]
starts=[i for i,l in enumerate(lines) if re.match(r'^_Z\S+:\s',l)]
def classify(l):
    m=re.match(r'\s+([a-z_0-9]+)',l)
    if not m: return None
    op=m.group(1)
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'ds'
    if op.startswith('global_load_lds') or (op.startswith('buffer_load') and ' lds' in l): return 'dma'
    if op.startswith('global_') or op.startswith('buffer_'): return 'vmem'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith('s_'): return 'salu'
    return 'other'
for si,s in enumerate(starts):
    name=lines[s].split(':')[0]
    if pat not in name: continue
    e=starts[si+1] if si+1<len(starts) else len(lines)
    body=lines[s:e]
    idx=[i for i,l in enumerate(body) if re.match(r'\s+s_barrier',l)]
    print(name, len(idx),'barriers')
    for a,b in zip(idx[:-1],idx[1:]):
        c=Counter(classify(l) for l in body[a+1:b]); c.pop(None,None)
        if c.get('mfma',0)>0: print('   ',dict(sorted(c.items())), 'total', sum(c.values()))
