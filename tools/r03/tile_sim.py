#!/usr/bin/env python3
"""Build-container simulation behind the block shapes of the binned sweep: the twitter stand-in's window,
numbered like the engine numbers it (hot blocks by in-degree, hashed inside), cut into A- / B-blocks like dppr_engine.hip bin_cut
does; prints how many tiles are populated and which share of the edges sits in tiles of <= 8 / 16 / 64 edges (short runs =
partial-line stores in k_bin_scatter). Needs the stand-in prefix under /tmp/dppr_data (tests/golden/make_fullsize_golden.py twitter)."""
sys.path.insert(0,'/root/repo')
from dynamicppr_amd import datagen
W=146836518
V,e1,e2=datagen.read_bin('/tmp/dppr_data/twitter-2010.rmat25.s4.first149773248.bin')
e1=e1[:W]; e2=e2[:W]
indeg=np.bincount(e2,minlength=V); outdeg=np.bincount(e1,minlength=V)
live=np.nonzero((indeg+outdeg)>0)[0]; n=len(live)
rng=np.random.default_rng(1)
d=indeg[live]; order=np.argsort(-d,kind='stable')
bounds=[0,8192,16384,32768,65536,131072,262144,524288,n]
newid=np.empty(n,np.int64)
for a,b in zip(bounds[:-1],bounds[1:]):
    idx=order[a:b]; newid[idx]=a+rng.permutation(b-a)
ext2int=np.full(V,-1,np.int64); ext2int[live]=newid
u=ext2int[e2]; v=ext2int[e1]
del e1,e2
ind=np.bincount(u,minlength=n); outd=np.bincount(v,minlength=n)
Ed=len(u)
def cut(deg,cap,target):
    pre=np.concatenate([[0],np.cumsum(deg)])
    K=(Ed+target-1)//target
    q=np.searchsorted(pre,np.arange(1,K)*target,side='left')
    big=np.nonzero(deg>=target//4)[0]
    c=np.unique(np.concatenate([np.arange(0,n,cap),q,big,big+1,[n]]))
    c=c[c<=n]
    return c
for ha,hb,ta,tb in ((128,48,98304,98304),(288,48,98304,98304),(288,48,4<<20,98304),(288,48,4<<20,196608),(288,112,4<<20,196608),(288,112,4<<20,393216),(1024,112,4<<20,196608),(1024,256,4<<20,196608)):
    ca=cut(ind,ha*64,ta); cb=cut(outd,hb*64,tb)
    ba=np.searchsorted(ca,u,side='right')-1; bb=np.searchsorted(cb,v,side='right')-1
    key=ba*len(cb)+bb
    uniq,cnt=np.unique(key,return_counts=True)
    eb=np.bincount(bb,minlength=len(cb)-1)
    print(f'ha {ha} hb {hb} ta {ta} tb {tb}: nA {len(ca)-1} nB {len(cb)-1} tiles {len(uniq)} avg {Ed/len(uniq):.1f} maxB {eb.max()}', ' frac<=8 %.3f <=16 %.3f <=64 %.3f'%tuple(cnt[cnt<=t].sum()/Ed for t in (8,16,64)))
