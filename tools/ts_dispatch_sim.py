"""HERE (no GPU): greedy dispatch of the batched weight-gradient launch (gemm_tn_split.hpp) on 256 CUs with the per-tile costs measured on the GPU
(tools/ts_wg_times.py: first_conv 2.45 us per 32-step tile in lock step / 2.2 staggered, layer workgroups 2.75 plain / 3.1 with two gradient images + dropout replay) and
a per-workgroup overhead of 5 - 9 us: makespan per chunk cap of the layer jobs, in launch order and sorted by duration (LPT), and the average load per CU."""
import heapq, itertools, math
B=8
levels=[4096,4096,2048,1024,1024,512,512,512,512,256,256]   # rows per video of layer l
def pick_mc(rows_total,kchunks,cap,target=64):
    want=(rows_total*kchunks+target-1)//target
    mc=((want+31)//32)*32
    mc=max(mc,128); mc=min(mc,cap)
    return mc
def wgs(cap_layer, fc_cost=2.45, ovh=7.0, cap_fc=2048, c_plain=2.75, c_two=3.1, fc_mc=None):
    out=[]
    # first_conv: Ktot 2048 -> 16 kchunks -> 8 col WGs
    mc=fc_mc or pick_mc(B*4096,16,cap_fc)
    for b in range(B):
        t=4096
        while t>0:
            n=min(mc,t); t-=n
            for kc in range(8): out.append(('fc',ovh+math.ceil(n/32)*fc_cost))
    for l,T in enumerate(levels):
        mc=pick_mc(B*T,4,min(2048,cap_layer))
        for b in range(B):
            t=T
            while t>0:
                n=min(mc,t); t-=n
                out.append(('L%d'%l,ovh+math.ceil(n/32)*c_plain))
                out.append(('L%d'%l,ovh+math.ceil(n/32)*c_two))
    # last conv: 256 rows, 1 kchunk -> 1 WG (half)
    mc=pick_mc(B*256,1,2048)
    for b in range(B):
        t=256
        while t>0:
            n=min(mc,t); t-=n
            out.append(('last',ovh+math.ceil(n/32)*2.6))
    return out
def makespan(w, ncu=256, sort=False):
    if sort: w=sorted(w,key=lambda x:-x[1])
    h=[0.0]*ncu; heapq.heapify(h)
    for _,d in w:
        t=heapq.heappop(h); heapq.heappush(h,t+d)
    return max(h), sum(d for _,d in w)/ncu
for ovh in (5,7,9):
    print('ovh',ovh)
    for cap in (2048,1024,768,512,384,256):
        w=wgs(cap,ovh=ovh)
        m,avg=makespan(w); ms,_=makespan(w,sort=True)
        w2=wgs(cap,fc_cost=2.2,ovh=ovh); m2,avg2=makespan(w2); m2s,_=makespan(w2,sort=True)
        print(f'  cap {cap:5d}: lock-step fc: makespan {m:6.1f} (sorted {ms:6.1f}) avg {avg:6.1f} | staggered fc: {m2:6.1f} (sorted {m2s:6.1f}) avg {avg2:6.1f}  nWG {len(w)}')
