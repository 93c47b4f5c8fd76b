#!/usr/bin/env python3
"""Static ISA attribution: VALU / FP64 / scalar / memory instructions of one kernel per source line.
Compile the translation unit with `-gline-tables-only --save-temps` and pass the gfx950 .s file:
  python tools/isa_by_line.py <file.s> <mangled-name pattern> <min instructions per line>"""
import re, sys, collections
S=open(sys.argv[1]).read().splitlines()
pat=sys.argv[2]
files={}
for l in S:
    m=re.match(r'\s+\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"',l)
    if m: files[int(m.group(1))]=m.group(2)
start=next(i for i,l in enumerate(S) if re.match(r'^_Z.*'+pat+r'.*:',l))
end=next(i for i in range(start,len(S)) if S[i].startswith('.Lfunc_end'))
cur=None
cnt=collections.Counter(); cat=collections.defaultdict(collections.Counter)
def kind(k):
    if k.startswith('v_') and 'f64' in k and not k.startswith('v_cmp') and not k.startswith('v_mov'): return 'f64'
    if k.startswith('v_'): return 'valu_other'
    if k.startswith('s_'): return 'salu'
    return 'mem'
for l in S[start:end]:
    m=re.match(r'\s+\.loc\s+(\d+)\s+(\d+)',l)
    if m: cur=(files.get(int(m.group(1)),'?'),int(m.group(2))); continue
    m=re.match(r'^\s+([a-z_0-9]+)\s',l)
    if m and not l.strip().startswith(('.',';')):
        cnt[cur]+=1; cat[cur][kind(m.group(1))]+=1
# aggregate per file:line, print those in fg_pipeline.h mu_step range and ndpp_math.h
tot=collections.Counter()
rows=[]
for k,v in cnt.items():
    rows.append((k,v,cat[k]))
rows.sort(key=lambda r:(r[0][0],r[0][1]))
for k,v,c in rows:
    if v>=int(sys.argv[3]) : print(f"{k[0]}:{k[1]:5d}  total {v:4d}  f64 {c['f64']:4d} valu_other {c['valu_other']:4d} salu {c['salu']:4d} mem {c['mem']:3d}")
