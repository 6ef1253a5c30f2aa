#!/usr/bin/env python3
"""L2 model of the program-driven aggregation (CPU only): one XCD's band of the cfg2 bench pair, `nslots` waves in
flight executing their programs' LOAD units in dispatch order (event-driven, a unit costs t_op per op + t_slot per slot +
t_add per add), an LRU of `cap` pixel records (4 MiB of 1 KiB records at four disparities per lane).  Output: window
loads, misses, and misses over compulsory misses.  Round 4 (measured FETCH_SIZE in brackets): 384 waves in flight 2.3 x
compulsory [2.09], 256 waves 1.8 [1.57], two disparities per lane at 384 waves 1.8 [1.37]: the resident patches' own
pixels (384 x 20 KiB) are twice the L2, whatever the dispatch order or the band geometry.
    python tools/model/cbca_l2.py"""
import sys, os, heapq, numpy as np
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[ROOT+'/mc-cnn-python_amd/src', ROOT+'/oracle', ROOT+'/tests/asmtools', ROOT+'/mc-cnn-python_amd/csrc/asm']
import oracle as o, synthetic, cbca_prog_ref as ref
H,W,D=500,750,256
L_,R_,_,_,_=synthetic.make_pair(H,W,D,seed=100)
def words(img):
    arms,cnt=o.cross_arms(img,0.02,14)
    a=arms.astype(np.uint32)
    return (a[...,0]|(a[...,1]<<5)|(a[...,2]<<10)|(a[...,3]<<15)|(cnt.astype(np.uint32)<<20)).astype(np.uint32)
sup=words(L_)
R=13
_cache={}
def patch_units(K,G,WW,rg,cg):
    key=(K,G,WW,rg,cg)
    if key in _cache: return _cache[key]
    L=dict(K=K,G=G,W=WW,MAXD=min(WW,14),MAXA=min(WW,13))
    y0,x0=rg*K,cg*G
    row0=max(y0-R,0)
    out=[]
    for lo,hi,p,runs,_ in ref.plan_units(sup,H,W,y0,x0,L):
        yq=row0+p//W; xhi=p%W   # p=(yq-row0)*W+(x0-R+hi)
        n=hi-lo+1
        adds=sum(r[4]*bin(r[2]).count('1') for r in runs)
        nops=1+sum(len(ref.decompose(r[2],K)) for r in runs)
        out.append((yq,xhi-n+1,n,adds,nops))
    _cache[key]=out
    return out

def simulate(K,G,WW,order,nslots=384,cap=4096,band=(0,63),store_alloc=True, t_op=1.0,t_slot=0.5,t_add=0.1, verbose=False):
    """order: list of (rg,cg) in dispatch order for one XCD. Returns (loads, misses, compulsory)."""
    lru=OrderedDict(); misses=0; loads=0
    seen=set()
    heap=[]  # (time, seq, waveid)
    waves={}
    nxt=0; seq=0; now=0.0
    def start(t):
        nonlocal nxt,seq
        while nxt<len(order) and len(waves)<nslots:
            rg,cg=order[nxt]; nxt+=1
            us=patch_units(K,G,WW,rg,cg)
            waves[seq]=[us,0,(rg,cg)]
            heapq.heappush(heap,(t,seq,seq)); seq+=1
    start(0.0)
    while heap:
        t,_,w=heapq.heappop(heap)
        us,i,pc=waves[w]
        if i>=len(us):
            # epilogue: stores of the K x G outputs
            if store_alloc:
                rg,cg=pc
                for k in range(K):
                    for j in range(G):
                        key=('o',rg*K+k,cg*G+j)
                        lru[key]=1
                        if len(lru)>cap: lru.popitem(last=False)
            del waves[w]
            start(t)
            continue
        yq,x,n,adds,nops=us[i]
        for xx in range(x,x+n):
            key=(yq,xx); loads+=1
            if key in lru: lru.move_to_end(key)
            else:
                misses+=1; lru[key]=1; seen.add(key)
                if len(lru)>cap: lru.popitem(last=False)
        waves[w][1]=i+1
        heapq.heappush(heap,(t+t_op*nops+t_slot*n+t_add*adds, w, w))
    return loads,misses,len(seen)

def order_colmajor(K,G,band_rows=63,H0=0):
    rgs=-(-band_rows//K); cgs=-(-W//G)
    return [(H0//K+r,c) for c in range(cgs) for r in range(rgs)]
def order_rowmajor(K,G,band_rows=63,H0=0):
    rgs=-(-band_rows//K); cgs=-(-W//G)
    return [(H0//K+r,c) for r in range(rgs) for c in range(cgs)]
if __name__=="__main__":
    for name,ordf in (("column-group-major",order_colmajor),("row-group-major",order_rowmajor)):
        for ns in (128,256,384):
            l,m,c=simulate(4,5,20,ordf(4,5,64,64),nslots=ns,store_alloc=False)
            print(name,"waves in flight",ns,"loads",l,"misses",m,"compulsory",c,"miss rate %.3f"%(m/l),"x compulsory %.2f"%(m/c))
    for ns in (384,512,640):
        l,m,c=simulate(4,5,20,order_colmajor(4,5,64,64),nslots=ns,cap=8192,store_alloc=False)
        print("two disparities per lane (512-byte records), waves in flight",ns,"x compulsory %.2f"%(m/c))
