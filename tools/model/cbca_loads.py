#!/usr/bin/env python3
"""Load-count model of the program-driven aggregation (CPU only): for a patch shape K x G and a window of W slots, how
many window slots (1 KiB loads at four disparities per lane), LOAD / ADD ops and float32 adds the programs of the cfg2
bench pair need per pixel - the plain-Python program builder (tests/asmtools/cbca_prog_ref.py) run over a sample of the
patches.  Round 4: 4 x 5 / 20 needs 4.8 slots per pixel, 8 x 5 4.1, 16 x 5 3.3, 4 x 10 3.7 - taller or wider patches
pay little, which is why the patch shape stayed.
    python tools/model/cbca_loads.py [K G W ...]"""
import sys, os, numpy as np
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
cnt=(sup>>20)
print("mean region size",cnt.mean(), "frac all-zero arms", ((sup&0xfffff)==0).mean())
rng=np.random.default_rng(0)
def model(K,G,WW,nsamp=600):
    L=dict(K=K,G=G,W=WW,MAXD=min(WW,14),MAXA=min(WW,13))
    rgs=-(-H//K); cgs=-(-W//G)
    idx=rng.choice(rgs*cgs,size=min(nsamp,rgs*cgs),replace=False)
    slots=units=runs=adds=0; pix=0
    for i in idx:
        rg,cg=divmod(int(i),cgs)
        us=ref.plan_units(sup,H,W,rg*K,cg*G,L)
        for lo,hi,p,rr,_ in us:
            slots+=hi-lo+1; units+=1
            for d,j,aset,first,n in rr:
                for m in ref.decompose(aset,K):
                    runs+=1; adds+=n*bin(m).count('1')
        pix+=min(K,H-rg*K)*min(G,W-cg*G)
    return slots/pix, units/pix, runs/pix, adds/pix
if __name__ == "__main__":
  shapes = [(2,5,20),(4,5,20),(8,5,20),(8,6,24),(4,8,24),(4,10,28),(16,5,20),(8,8,24)]
  if len(sys.argv) > 3:
    v = [int(x) for x in sys.argv[1:]]
    shapes = list(zip(v[0::3], v[1::3], v[2::3]))
  for K,G,WW in shapes:
    s,u,r,a=model(K,G,WW,nsamp=2000)
    print("K=%d G=%d W=%d: slots/pixel %.2f  LOAD ops/pixel %.3f  ADD ops/pixel %.3f  adds/pixel %.1f  ops/pixel %.3f"%(K,G,WW,s,u,r,a,u+r))
