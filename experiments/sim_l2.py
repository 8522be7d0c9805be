import sys; sys.path.insert(0,'graphsage-simple_amd'); sys.path.insert(0,'.')
import numpy as np
from collections import OrderedDict
from sage355.graph import rmat_graph
from oracle import sampler_ref
g=rmat_graph(20,16_000_000,cache_dir='/tmp/sage_cache')
deg=g.degrees()
seeds=np.random.default_rng(1).choice(np.nonzero(deg>0)[0],4096,replace=False).astype(np.int32)
n2,c2=sampler_ref.sample_neighbors(g.rowptr,g.col,seeds,25,1,2)
u2=np.unique(n2[n2>=0]).astype(np.int32)
rng=np.random.default_rng(0); u2=u2[rng.permutation(u2.size)]   # hash-like order
n1,c1=sampler_ref.sample_neighbors(g.rowptr,g.col,u2,15,1,1)
def simulate(order, cap_rows, nx=8, tile=64, interleave=True):
    # tiles of `tile` dests dealt round-robin (blockIdx%8) or contiguous ranges per XCD; per-XCD LRU of cap_rows rows
    caches=[OrderedDict() for _ in range(nx)]
    hits=0; total=0
    ntile=(len(order)+tile-1)//tile
    for t in range(ntile):
        x = t % nx if interleave else min(nx-1, t*nx//ntile)
        c=caches[x]
        for d in order[t*tile:(t+1)*tile]:
            for s in n1[d,:c1[d]]:
                total+=1
                if s in c:
                    hits+=1; c.move_to_end(s)
                else:
                    c[s]=1
                    if len(c)>cap_rows: c.popitem(last=False)
    return hits/total
idx=np.arange(u2.size)
nb=np.where(n1>=0,n1,2**31-1)
minnb=nb.min(1)
order_min=np.argsort(minnb,kind='stable')
# sort by the neighbour with the highest degree
degn=np.where(n1>=0,deg[np.clip(n1,0,None)],-1)
top=n1[np.arange(len(n1)),degn.argmax(1)]
order_top=np.argsort(top,kind='stable')
# lexicographic on sorted neighbour lists (first 3)
sn=np.sort(nb,1)
order_lex=np.lexsort((sn[:,2],sn[:,1],sn[:,0]))
for cap in (2048,3500):
    print('cap',cap,'random order rr: %.3f'%simulate(idx,cap),' contiguous: %.3f'%simulate(idx,cap,interleave=False))
    print('   min-nbr sorted rr: %.3f contiguous: %.3f'%(simulate(order_min,cap),simulate(order_min,cap,interleave=False)))
    print('   top-deg-nbr sorted contiguous: %.3f'%simulate(order_top,cap,interleave=False))
    print('   lex sorted contiguous: %.3f'%simulate(order_lex,cap,interleave=False))
print('one big shared cache (no XCD split) cap 4096x8: %.3f'%simulate(idx,32768,nx=1))

# --- sorted-by-source sweep: beyond-L2 traffic = distinct sources per XCD partition
E=int(c1.sum())
for nparts,label in ((8,'XCD'),(256,'CU')):
    part=(np.arange(len(n1))*nparts//len(n1))
    tot=0
    for x in range(nparts):
        rows=n1[part==x]; rows=rows[rows>=0]
        tot+=np.unique(rows).size
    print(f'{label} partitions={nparts}: sum of distinct sources = {tot} of {E} edges ({tot/E:.3f}); global distinct {np.unique(n1[n1>=0]).size}')
