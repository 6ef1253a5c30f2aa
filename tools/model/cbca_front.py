#!/usr/bin/env python3
"""How wide is the front of resident patches, and what would a narrower one buy?  (CPU only; the L2 model of cbca_l2.py
with a dispatch throttle: a patch of column group c starts only once every patch of column group c - delta has ended.)
Round 4: unthrottled, the 384 resident waves of an XCD spread over more than 48 column groups - light patches end at
once, the slots fill up with the long serial chains (LOAD -> wait -> ADD, ~100-300 units) of heavy patches far ahead -
and the misses per pixel fall from 2.5 (full programs) / 2.0 (skip programs) to 1.4 / 0.8 at delta = 24 and 1.1 / 0.7
at delta = 12, but the slots idle: the throttle alone does not pay.  What it says is that the over-fetch is a
chain-length problem: a wave that keeps several windows in flight (a software pipeline through LDS, one fat wave per
SIMD) shortens the chains, narrows the front and needs a third of the resident patches.
    python tools/model/cbca_front.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tools', 'model')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'asmtools'))
import numpy as np, heapq
from collections import OrderedDict
import cbca_l2
from cbca_l2 import sup,H,W
import cbca_prog_ref as ref
K,G,WW=4,5,20
L=dict(K=K,G=G,W=WW,MAXD=14,MAXA=13)
_c={}
def units(rg,cg,skip):
    key=(rg,cg,skip)
    if key in _c: return _c[key]
    y0,x0=rg*K,cg*G
    us=ref.plan_units(sup,H,W,y0,x0,L,skip_unit=skip)
    row0=max(y0-13,0); out=[]
    for lo,hi,p,runs,_ in us:
        yq=row0+p//W; xhi=p%W; n=hi-lo+1
        adds=sum(r[4]*bin(r[2]).count('1') for r in runs)
        nops=1+sum(len(ref.decompose(r[2],K)) for r in runs)
        out.append((yq,xhi-n+1,n,adds,nops))
    _c[key]=out
    return out
def sim(skip,delta,nslots=384,cap=4096):
    rgs=16; cgs=150; r0=16
    order=[(r0+r,c) for c in range(cgs) for r in range(rgs)]
    remaining=[rgs]*cgs   # unfinished patches per column group
    lru=OrderedDict(); misses=0; loads=0; heap=[]; waves={}; nxt=0; seq=0
    idle=0.0; tlast=0.0
    def lowest_unfinished():
        for c in range(cgs):
            if remaining[c]>0: return c
        return cgs
    def start(t):
        nonlocal nxt,seq
        lo=lowest_unfinished()
        while nxt<len(order) and len(waves)<nslots and order[nxt][1] < lo+delta:
            rg,cg=order[nxt]; nxt+=1
            waves[seq]=[units(rg,cg,skip),0,cg]; heapq.heappush(heap,(t+4.0,seq,seq)); seq+=1
    start(0.0)
    busy_integral=0.0
    while heap:
        t,_,w=heapq.heappop(heap)
        busy_integral+=(t-tlast)*len(waves); tlast=t
        us,i,cg=waves[w]
        if i>=len(us):
            del waves[w]; remaining[cg]-=1; start(t); continue
        yq,x,n,a,nops=us[i]
        for xx in range(x,x+n):
            key=(yq,xx); loads+=1
            if key in lru: lru.move_to_end(key)
            else:
                misses+=1; lru[key]=1
                if len(lru)>cap: lru.popitem(last=False)
        waves[w][1]=i+1
        heapq.heappush(heap,(t+1.0*nops+0.5*n+0.1*a,w,w))
    return loads,misses,tlast,busy_integral/(tlast*nslots)
for skip in (False,True):
    for delta in (1000,48,32,24,16,12,8):
        l,m,T,util=sim(skip,delta)
        print("skip %d delta %4d: misses/pixel %.2f  makespan %.0f  slot utilisation %.3f"%(skip,delta,m/(64*750),T,util),flush=True)
