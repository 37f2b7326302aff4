"""Instruction histogram of the loop that contains a given instruction (default: s_barrier) in one kernel's assembly
(hipcc -S --cuda-device-only, one kernel cut out with awk): how many VALU / scalar / MFMA / LDS / DMA instructions a K-loop
iteration issues per wave -- the method of DESIGN 12.4.   python tools/loopstat.py kernel.s [instruction]"""
import re, sys
from collections import Counter
def loopstat(path, must='s_barrier'):
    lines=open(path).read().split('\n')
    # split into blocks by labels
    blocks=[]; cur=None
    for l in lines:
        m=re.match(r'^(\.LBB\d+_\d+):(.*)$', l)
        if m:
            cur=dict(label=m.group(1), note=m.group(2), ins=[])
            blocks.append(cur)
        elif cur is not None:
            t=l.strip()
            if t and not t.startswith(';') and not t.startswith('.'):
                cur['ins'].append(t)
            elif '%bb.' in t and cur is not None:
                pass
    # find header of the loop containing `must`
    hdr=None
    for b in blocks:
        if any(must in i for i in b['ins']):
            m=re.search(r'Header=(BB\d+_\d+)', b['note'])
            hdr='.L'+m.group(1) if m else b['label']
    c=Counter()
    for b in blocks:
        if b['label']==hdr or ('Header='+hdr[2:]) in b['note']:
            for i in b['ins']: c[i.split()[0]]+=1
    valu=sum(v for k,v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    slow=sum(v for k,v in c.items() if k in ('v_mul_lo_u32','v_mul_hi_u32','v_mad_u64_u32','v_mad_i64_i32','v_mul_hi_i32'))
    trans=sum(v for k,v in c.items() if re.match(r'v_(exp|rcp|log|rsq|sqrt)_',k))
    print(path,'loop',hdr,'total',sum(c.values()),'valu',valu,'(slow-mul',slow,'trans',trans,') mfma',sum(v for k,v in c.items() if k.startswith('v_mfma')),'salu',sum(v for k,v in c.items() if k.startswith('s_')),'ds_read',sum(v for k,v in c.items() if k.startswith('ds_read')),'dma',c['buffer_load_dwordx4'],'branches',sum(v for k,v in c.items() if 'branch' in k))
    return c
if __name__=='__main__':
    c=loopstat(sys.argv[1], sys.argv[2] if len(sys.argv)>2 else 's_barrier')
    print(c.most_common(16))
