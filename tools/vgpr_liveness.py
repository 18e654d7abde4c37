#!/usr/bin/env python3
"""Build-time diagnostic (no GPU): control-flow-aware VGPR liveness of one kernel, from the compiler's own assembly.

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -c --save-temps x.hip        (x.hip instantiates the kernel)
    python tools/vgpr_liveness.py kernel.s [first_line last_line]

kernel.s = the text of ONE kernel cut out of *-hip-amdgcn-amd-amdhsa-gfx950.s (from its label to .amdhsa_kernel).  Parses
definitions / uses of v-registers per instruction (approximately: the first operand of VALU / ds_read / load instructions is
the definition, DPP and v_fmac destinations are also read), builds the CFG from the .LBB labels and branches, iterates the
backward dataflow to a fixed point and prints the maximum number of simultaneously live VGPRs per stretch of the listing,
with the markers (s_setprio, stores, LDS-DMA) that tell the phases of a row step apart.  The allocated count minus this
maximum is what the allocator loses to tuple alignment and fragmentation (typically 40 - 50 registers in the strip kernels);
a stretch far above its twin (the two copies of a templated step) is a live range the code layout keeps open -- that is how
the 44 - 60 spills of the first hadi_pass_a_pairs were traced to its duplicated step."""
import re,sys
F=sys.argv[1]
lines=open(F).read().split("\n")
def regs(tok):
    out=[]
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b",tok):
        if m.group(1): out+=list(range(int(m.group(1)),int(m.group(2))+1))
        else: out.append(int(m.group(3)))
    return out
# parse instructions, labels
ins=[]; label_at={}
for i,l in enumerate(lines):
    l0=l.split(";")[0].rstrip()
    m=re.match(r"^(\.LBB\d+_\d+):",l0)
    if m: label_at[m.group(1)]=len(ins); continue
    l0=l0.strip()
    if not l0 or l0.startswith("."): continue
    parts=l0.split(None,1); op=parts[0]; args=parts[1] if len(parts)>1 else ""
    toks=[t.strip() for t in args.split(",")]
    nodst = op.startswith(("global_store","ds_write","scratch_store","buffer_store","global_load_lds","s_","v_cmp","v_readlane","v_readfirstlane"))
    d=[];u=[]
    if toks and not nodst and (op.startswith(("v_","ds_read","ds_bpermute","global_load","scratch_load","buffer_load","ds_swizzle"))):
        d=regs(toks[0])
        for t in toks[1:]: u+=regs(t)
        if op.startswith(("v_fmac","v_mac")): u+=d
        if "dpp" in op or "row_" in args or "quad_perm" in args: u+=d
    else:
        for t in toks: u+=regs(t)
    tgt=None
    m=re.search(r"(\.LBB\d+_\d+)",args)
    if op.startswith(("s_cbranch","s_branch")) and m: tgt=m.group(1)
    ins.append(dict(line=i,op=op,d=set(d),u=set(u),tgt=tgt))
n=len(ins)
succ=[[] for _ in range(n)]
for k,I in enumerate(ins):
    if I["op"]=="s_endpgm": continue
    if I["op"]=="s_branch": succ[k]=[label_at[I["tgt"]]]; continue
    if k+1<n: succ[k].append(k+1)
    if I["tgt"]: succ[k].append(label_at[I["tgt"]])
livein=[set() for _ in range(n)]
changed=True
it=0
while changed:
    changed=False; it+=1
    for k in range(n-1,-1,-1):
        out=set()
        for s in succ[k]:
            if s<n: out|=livein[s]
        new=(out-ins[k]["d"])|ins[k]["u"]
        if new!=livein[k]: livein[k]=new; changed=True
print("iterations",it,"max live",max(len(x) for x in livein))
lo=int(sys.argv[2]) if len(sys.argv)>2 else 0; hi=int(sys.argv[3]) if len(sys.argv)>3 else 10**9
sel=[k for k in range(n) if lo<=ins[k]["line"]<hi]
step=max(1,len(sel)//70)
for a in range(0,len(sel),step):
    seg=sel[a:a+step]
    marks=[ins[k]["op"] for k in seg if ins[k]["op"].startswith(("s_setprio","s_sleep","global_store","global_load_lds","s_cbranch","v_rcp"))]
    print(ins[seg[0]]["line"], max(len(livein[k]) for k in seg), " ".join(marks[:6]))
