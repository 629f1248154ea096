"""Instruction stream between the MFMAs of a source-line range of one kernel (gfx950 assembly built with -g1), condensed:
   hipcc --offload-arch=gfx950 -O3 -g1 --cuda-device-only -S file.hip -o /tmp/pl/g.s; python tools/mfma_schedule.py <mangled kernel> <first line> <last line> [source file name]
   (reads /tmp/pl/g.s)"""
import re,sys
s=open('/tmp/pl/g.s').read()
kern=sys.argv[1]; lo=int(sys.argv[2]); hi=int(sys.argv[3]); fname=sys.argv[4] if len(sys.argv)>4 else 'gmlp.hip'
m=re.search(r"^("+re.escape(kern)+r"):[^\n]*\n(.*?)\.Lfunc_end", s, re.S|re.M)
body=m.group(2).split("\n")
files={}
for mm in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s, re.M):
    files[int(mm.group(1))]=(mm.group(3) or mm.group(2)).split("/")[-1]
seq=[];cur=None
for l in body:
    t=l.strip()
    if t.startswith(".loc"):
        f=t.split(); cur=(files.get(int(f[1]),''),int(f[2])); continue
    if not t or t.startswith((".",";","//")) or t.split()[0].endswith(":"): continue
    seq.append((cur,t.split()[0], t))
idx=[i for i,(c,o,t) in enumerate(seq) if c and c[0]==fname and lo<=c[1]<=hi and o.startswith('v_mfma')]
print(len(idx))
a,b=idx[0]-10, idx[-1]+3
comp=[]; run=0
for c,o,t in seq[a:b]:
    if o.startswith('v_mfma'): run+=1; continue
    if run: comp.append(f"[mfma x{run}]"); run=0
    if o.startswith(('s_waitcnt','ds_read','ds_write','scratch_','s_nop','s_cbranch','global_')): comp.append(o+(" "+t.split(None,1)[1][:14] if o.startswith('s_waitcnt') else ''))
if run: comp.append(f"[mfma x{run}]")
res=[]
for x in comp:
    if res and res[-1][0]==x: res[-1][1]+=1
    else: res.append([x,1])
print(" ".join(f"{x}{'*'+str(n) if n>1 else ''}" for x,n in res))
