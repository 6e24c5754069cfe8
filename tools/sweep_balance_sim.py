"""Load balance of the MFMA sweep's work items over the chip's 1 024 SIMDs under different item-to-SIMD assignments (CPU, numpy; round 6).

A launch of the correlate-then-interpolate sweep lasts as long as its busiest SIMD, and a SIMD's busy time follows the 16-cell tiles of the items it got
(profiles/r06_sweep_mfma_v3_timeline.txt: correlation 0.96 - 0.98).  Item cost here = exact tile count of the item (boxes enumerated from the sample
positions, as tools/sweep_mfma_model.py does) + a fixed per-item overhead in tile equivalents.  Assignments compared, per keyframe pair, as
(max SIMD load) / (mean SIMD load):
  band_greedy   round 5: XCD x owns a contiguous band of group rows, items handed to free wave slots in launch order (list scheduling)
  rows_greedy   XCD x owns rows x, x + 8, ... (interleaved)
  sym_greedy    XCD x owns the point-symmetric row set {x, 15 - x, 16 + x, 31 - x}
  *_mod         persistent workgroups, list l of an XCD gets chunk c of group (l - 32 c) mod 128 (+ one more)
  sym_pair      persistent workgroups, list l gets chunks 0 / 3 of group l and chunks 1 / 2 of its mirror image (+ one more): the shipped static part
  sym_pair_lptextra   ... the 128 extra items of an XCD given, heaviest first, to the least loaded lists (an XCD-wide queue: not built)
  lpt_global    longest-processing-time-first over all 1 024 SIMDs with the true costs: the bound
      python tools/sweep_balance_sim.py [--every 12]
"""
import numpy as np, sys, heapq
from sweep_geometry import index_lines, sample_positions, syn, H, W, D
from sweep_mfma_model import chunk_boxes
lines=index_lines(2); poses=syn.sample_poses(); K=syn.scaled_K(syn.full_K(),2.0)[0].double().numpy()
OVH=14.0
def tilemap(li):
    ref,*meas=lines[li]
    tot=np.zeros((4,32,40))
    for m in meas:
        sx,sy,_=sample_positions(poses[ref],poses[m],K)
        cells16,_=chunk_boxes(sx,sy,4,4,16); cells4,_=chunk_boxes(sx,sy,4,4,4)
        t16=np.ceil(cells16/16); t4=np.ceil(cells4/16).reshape(4,4,32,40).sum(axis=1)
        tot+=np.where(cells16<=256,t16,t4)
    return tot
def greedy(costs_in_order, nsimd=128, per=4):
    slots=[(0.0,s) for s in range(nsimd*per)]
    heapq.heapify(slots); simd=np.zeros(nsimd)
    for c in costs_in_order:
        t,s=heapq.heappop(slots); heapq.heappush(slots,(t+c*per,s)); simd[s%nsimd]+=c
    return simd
def rows_of(mode,x):
    if mode=="band": return list(range(4*x,4*x+4))
    if mode=="rows": return [x,x+8,x+16,x+24]
    if mode=="sym": return sorted([x,15-x,16+x,31-x])
def lists_mod(cost, rows):
    g=[(r,c) for r in rows for c in range(40)]; L=128; lists=np.zeros(L)
    for l in range(L):
        for n in range(8):
            c=n%4; r=n//4; gi=(l-32*c)%L + r*L
            if gi<160: lists[l]+=cost[(c,)+g[gi]]
    return lists
def lists_sym(cost, rows, variant=0):
    g=[(r,c) for r in rows for c in range(40)]; L=128; lists=np.zeros(L); seen=set()
    def add(l,gi,c):
        assert (gi,c) not in seen; seen.add((gi,c)); lists[l]+=cost[(c,)+g[gi]]
    for l in range(L):
        u=l
        add(l,u,0); add(l,u,3); add(l,159-u,1); add(l,159-u,2)
    # extras: chunk0,3 of groups 128..159 ; chunk 1,2 of groups 0..31 : 128 items
    extras=[(gi,0) for gi in range(128,160)]+[(gi,3) for gi in range(128,160)]+[(gi,1) for gi in range(0,32)]+[(gi,2) for gi in range(0,32)]
    if variant==0:
        for l,(gi,c) in enumerate(extras): add(l,gi,c)
    else:
        # heavy extras (by chunk weight) to lists far from ... simple: sort extras by nominal weight desc, lists by base asc (true costs unknown in kernel: use nominal position-independent) -> here cheat with true cost to see the bound
        order=np.argsort(lists); ex=sorted(extras,key=lambda e:-cost[(e[1],)+g[e[0]]])
        for l,(gi,c) in zip(order,ex): add(l,gi,c)
    assert len(seen)==640
    return lists
def lpt(cost_items, nbins=1024):
    bins=[(0.0,i) for i in range(nbins)]; heapq.heapify(bins)
    for c in sorted(cost_items,reverse=True):
        t,i=heapq.heappop(bins); heapq.heappush(bins,(t+c,i))
    return max(t for t,_ in bins)
res={}
import sys
every=int(sys.argv[sys.argv.index("--every")+1]) if "--every" in sys.argv else 12
sel=list(range(0,285,every))+[170,202]
for li in sel:
    tm=tilemap(li); cost=tm+OVH; mean=cost.sum()/1024
    row={}
    for mode in ("band","rows","sym"):
        wg=0; wm=0; ws=0; ws1=0
        for x in range(8):
            rows=rows_of(mode,x)
            order=[cost[c,r,gx] for c in range(4) for r in rows for gx in range(40)]
            wg=max(wg,greedy(order).max()); wm=max(wm,lists_mod(cost,rows).max())
            if mode=="sym": ws=max(ws,lists_sym(cost,rows).max()); ws1=max(ws1,lists_sym(cost,rows,1).max())
        row[mode+"_greedy"]=wg/mean; row[mode+"_mod"]=wm/mean
        if mode=="sym": row["sym_pair"]=ws/mean; row["sym_pair_lptextra"]=ws1/mean
    row["lpt_global"]=lpt(cost.ravel())/mean
    res[li]=row
    print(li, f"mean {mean:.0f}", " ".join(f"{k}={v:.2f}" for k,v in row.items()), flush=True)
keys=list(res[sel[0]].keys())
print("MEAN", " ".join(f"{k}={np.mean([res[l][k] for l in sel]):.3f}" for k in keys))
print("MAX ", " ".join(f"{k}={np.max([res[l][k] for l in sel]):.3f}" for k in keys))

print("---- pooled extras ----")
def lists_sym_pool(cost, rows, pool):
    """static: list l = chunks 0,3 of group l, chunks 1,2 of group 159-l; extras pooled per CU (4 lists) or per XCD, given greedily (in enumeration order) to the least loaded list of the pool"""
    g=[(r,c) for r in rows for c in range(40)]; L=128; lists=np.zeros(L)
    for l in range(L):
        lists[l]+=cost[(0,)+g[l]]+cost[(3,)+g[l]]+cost[(1,)+g[159-l]]+cost[(2,)+g[159-l]]
    extras=[(gi,0) for gi in range(128,160)]+[(gi,1) for gi in range(0,32)]+[(gi,2) for gi in range(0,32)]+[(gi,3) for gi in range(128,160)]
    if pool=="cu":
        # CU j owns lists 4j..4j+3 and extras j, 32+j, 64+j, 96+j (one of each chunk), heavy chunk first
        for j in range(32):
            ls=list(range(4*j,4*j+4))
            for e in (j,32+j,64+j,96+j):
                gi,c=extras[e]; k=min(ls,key=lambda i:lists[i]); lists[k]+=cost[(c,)+g[gi]]
    else:
        for gi,c in extras:
            k=int(np.argmin(lists)); lists[k]+=cost[(c,)+g[gi]]
    return lists
res2={}
for li in sel:
    tm=tilemap(li); cost=tm+OVH; mean=cost.sum()/1024
    row={}
    for pool in ("cu","xcd"):
        w=0
        for x in range(8):
            w=max(w,lists_sym_pool(cost,rows_of("sym",x),pool).max())
        row[pool]=w/mean
    tot=[sum(cost[c,r,gx] for c in range(4) for r in rows_of("sym",x) for gx in range(40)) for x in range(8)]
    row["xcd_total_max"]=max(tot)/np.mean(tot)
    res2[li]=row
    print(li," ".join(f"{k}={v:.2f}" for k,v in row.items()),flush=True)
for k in ("cu","xcd","xcd_total_max"):
    print(k,"mean",np.mean([res2[l][k] for l in sel]),"max",np.max([res2[l][k] for l in sel]))
