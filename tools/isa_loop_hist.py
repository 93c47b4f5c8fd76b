#!/usr/bin/env python3
"""Static ISA histogram of the loops of one kernel (largest backward-branch regions of its .s):
  python tools/isa_loop_hist.py <file.s> <mangled-name pattern> [loop index]"""
import re, sys, collections
S=open(sys.argv[1]).read().splitlines()
pat=sys.argv[2]
start=None
for i,l in enumerate(S):
    if re.match(r'^_Z.*'+pat+r'.*:\s', l+' ') and start is None: start=i
assert start is not None
end=next(i for i in range(start,len(S)) if S[i].startswith('.Lfunc_end'))
body=S[start:end]
# labels
lab={}
for i,l in enumerate(body):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: lab[m.group(1)]=i
# find backward branches, choose the largest loop
loops=[]
for i,l in enumerate(body):
    m=re.match(r'^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)',l)
    if m and m.group(1) in lab and lab[m.group(1)]<i: loops.append((i-lab[m.group(1)],lab[m.group(1)],i))
loops.sort(reverse=True)
print("loops (len,start,end):",loops[:5])
n,a,b=loops[int(sys.argv[3]) if len(sys.argv)>3 else 0]
cnt=collections.Counter()
for l in body[a:b+1]:
    m=re.match(r'^\s+([a-z_0-9]+)',l)
    if m and not l.strip().startswith(('.',';')): cnt[m.group(1)]+=1
tot=sum(cnt.values())
def cat(k):
    if k.startswith('v_') and ('f64' in k): return 'valu_f64'
    if k.startswith('v_mov_b64') or k.startswith('v_mov') or k.startswith('v_accvgpr'): return 'valu_mov'
    if k.startswith('v_cndmask'): return 'valu_sel'
    if k.startswith('v_cmp'): return 'valu_cmp'
    if k.startswith('v_'): return 'valu_int/other'
    if k.startswith('s_'): return 'salu'
    if k.startswith('ds_'): return 'lds'
    if k.startswith(('global_','flat_','buffer_','scratch_')): return 'vmem'
    return 'other'
cc=collections.Counter()
for k,v in cnt.items(): cc[cat(k)]+=v
print("total",tot); 
for k,v in cc.most_common(): print(f"  {k:16s}{v}")
for k,v in cnt.most_common(60): print(f"{k:28s}{v}")
