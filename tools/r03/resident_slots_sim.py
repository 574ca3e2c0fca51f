#!/usr/bin/env python3
"""Build-container simulation behind the slot table of the resident sweep (dppr_resident.hpp): the configs[1] stand-in window, hashed
ids, <= 256 sweep groups; counts the distinct 64- / 128-byte sectors the wave instructions of the gather phase touch with the
slots in CSR order, sorted by gather position, and sorted with the H hottest vertices packed behind the vectors (measured on
the GPU afterwards: sorting wins, packing loses -- see the header comment there). usage: resident_slots_sim.py [youtube|dblp]"""
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from dynamicppr_amd import datagen, stream as st
key='youtube' if len(sys.argv)<2 else sys.argv[1]
cfg=datagen.STAND_INS[key]
V,e1,e2,_=datagen.stand_in_stream(key,'/tmp/dppr_data')
f=cfg.flags.split(); opt={f[i]:f[i+1] for i in range(0,len(f),2)}
wl=st.workload_config(cfg.edges,0.1,int(opt.get('-n',0)),float(opt.get('-r',-1.0)),int(opt.get('-b',0)),int(opt.get('-c',0)),int(opt.get('-l',0)))
W=wl.window; print('V',V,'W',W,'c',wl.per_batch)
a=e1[:W].astype(np.int64); b=e2[:W].astype(np.int64)
# undirected: both directions. out-CSR rows v, cols u: edge (v->u) means x[u] pulled into v
v=np.concatenate([a,b]); u=np.concatenate([b,a])
deg=np.bincount(v,minlength=V)+0
live=np.nonzero(np.bincount(v,minlength=V)+np.bincount(u,minlength=V))[0]; n=len(live)
rng=np.random.default_rng(1)
newid=rng.permutation(n); ext2int=np.full(V,-1,np.int64); ext2int[live]=newid
v=ext2int[v]; u=ext2int[u]
Ed=len(v); print('n',n,'Ed',Ed)
indeg=np.bincount(u,minlength=n)
# groups: consecutive tiles of 64, <=1024 vertices, balanced by edges (approx: equal-weight greedy with weight = edges + vertices)
outd=np.bincount(v,minlength=n)
wt=outd+1
G=min(256,(n+1023)//1024)
G=256
pre=np.concatenate([[0],np.cumsum(wt)])
cuts=[0]
tiles=(n+63)//64
# simple: choose cuts on tile boundaries near equal weight, respecting <=16 tiles
tw=np.add.reduceat(wt,np.arange(0,n,64))
tp=np.concatenate([[0],np.cumsum(tw)])
t=0
for g in range(G):
    rem=G-g
    target=(tp[-1]-tp[t])/rem
    t2=t+1
    while t2<tiles and t2-t<16 and tp[t2+1-0]-tp[t]<=target and tiles-t2>rem-1: t2+=1
    # must keep enough groups for remaining tiles: tiles - t2 <= (rem-1)*16
    while tiles-t2>(rem-1)*16: t2+=1
    cuts.append(t2); t=t2
    if t>=tiles: break
cuts=np.array(cuts); print('groups',len(cuts)-1,'max tiles',np.diff(cuts).max())
grp_of_v=np.searchsorted(cuts*64,np.arange(n),side='right')-1
order=np.lexsort((u,v))
v=v[order]; u=u[order]
ge=grp_of_v[v]
eg=np.bincount(ge,minlength=len(cuts)-1); print('edges per group: max',eg.max(),'mean',eg.mean())
def count(colpos, sort):
    # per group: edge list (CSR order or sorted by colpos), instruction = 64 consecutive
    tot64=0; tot128=0; totinstr=0
    start=np.concatenate([[0],np.cumsum(eg)])
    for g in range(len(eg)):
        c=colpos[start[g]:start[g+1]]
        if sort: c=np.sort(c)
        m=len(c); pad=(-m)%64
        c=np.concatenate([c,np.full(pad,-1)]).reshape(-1,64)
        s64=c>>3; s128=c>>4
        for arr,name in ((s64,'64'),(s128,'128')):
            srt=np.sort(arr,axis=1)
            distinct=(np.diff(srt,axis=1)!=0).sum(axis=1)+1
            # remove the pad (-1) class
            distinct-= (srt[:,0]<0)&(srt[:,-1]>=0)
            if name=='64': tot64+=distinct.sum()
            else: tot128+=distinct.sum()
        totinstr+=len(c)
    return tot64,tot128,totinstr
base=count(u,False); print('baseline CSR order: 64B requests',base[0],'128B',base[1],'instr',base[2],'per edge',base[0]/Ed)
s=count(u,True); print('sorted by col: ',s[0],s[1], 'ratio',s[0]/base[0])
for H in (1024,4096,16384,65536):
    rank=np.argsort(-indeg,kind='stable')
    pos=np.arange(n); pos[rank[:H]]=n+np.arange(H)
    share=indeg[rank[:H]].sum()/Ed
    h=count(pos[u],True); print('hot array H',H,'share of gathers %.3f'%share,': ',h[0],h[1],'ratio',h[0]/base[0], h[1]/base[1])
# full degree-sorted positions (upper bound of clustering)
rank=np.argsort(-indeg,kind='stable'); pos=np.empty(n,np.int64); pos[rank]=np.arange(n)
h=count(pos[u],True); print('all positions degree-sorted + sorted slots:',h[0],h[1],'ratio',h[0]/base[0],h[1]/base[1])
