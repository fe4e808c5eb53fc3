"""Offline fuzz of the interval decision (csrc/seqscan.h: lane_tight) against the sequential float32 chain, on the host:
random / clustered / periodic rows, ten bias pairs, targets uniform and on / one ulp around float32 and exact partial sums.
usage: fuzz_lane_tight.py <seed> <seconds> [device]     (prints WRONG lines, if any, and a summary; needs the built library;
`device`: one GPU thread per target instead of the host build of the routines)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pecanpy_amd import _lib
lib=_lib.load()
AMB=0xfffffffd
def run(cls,w_out,w_prev,r):
    cls=np.ascontiguousarray(cls,dtype=np.uint8); r=np.ascontiguousarray(r,dtype=np.float64)
    ch=np.zeros(r.size,np.uint32); ln=np.zeros_like(ch); km=np.zeros_like(ch); tg=np.zeros_like(ch)
    rc=lib.pw_selftest_lane(DEVICE,0,cls.ctypes.data,cls.size,w_out,w_prev,r.ctypes.data,r.size,ch.ctypes.data,ln.ctypes.data,km.ctypes.data,tg.ctypes.data,None)
    assert rc==0
    return ch,ln,tg
seed=int(sys.argv[1]); tmax=float(sys.argv[2]); DEVICE=int(len(sys.argv)>3 and sys.argv[3]=='device')
rng=np.random.default_rng(seed)
t0=time.time(); it=0; amb=0; settled=0; wrong=0
biases=[(0.5,2.0),(2.0,0.5),(0.25,4.0),(1.0,1.0),(4.0,0.125),(0.0625,16.0),(8.0,1.0),(0.5,0.5),(0.125,8.0),(1.0,4.0)]
while time.time()-t0<tmax:
    it+=1
    n=int(2**rng.uniform(1,17.5))
    w_out,w_prev=biases[rng.integers(len(biases))]
    pc=10**rng.uniform(-4,0)
    cls=(rng.random(n)<pc).astype(np.uint8)
    mode=rng.integers(4)
    if mode==1:  # clustered
        a=rng.integers(0,n); b=rng.integers(a,n+1); m=np.zeros(n,bool); m[a:b]=True; cls=(cls.astype(bool)&m).astype(np.uint8) if rng.random()<0.5 else m.astype(np.uint8)
    if mode==2 and n>8:  # periodic
        cls=np.zeros(n,np.uint8); cls[::int(rng.integers(2,64))]=1
    if rng.random()<0.7:
        cls[rng.integers(n)]=2
    w=np.where(cls==1,1.0,np.where(cls==0,w_out,w_prev))
    if w.sum()>2**24*min(1.0,w_out,w_prev): continue
    tot=np.float32(w.sum()); x=(w.astype(np.float32)/tot).astype(np.float32)
    c=np.cumsum(x,dtype=np.float32).astype(np.float64)
    ex=np.cumsum(w)/w.sum()
    k=rng.integers(0,n,size=min(n,300))
    r=np.concatenate([rng.random(500),c[k],np.nextafter(c[k],0),np.nextafter(c[k],2),ex[k],np.nextafter(ex[k],0),np.nextafter(ex[k],2)])
    r=np.clip(r,0,np.nextafter(1.0,0.0))
    ch,ln,tg=run(cls,w_out,w_prev,r)
    a=ln==AMB; s=a&(tg!=AMB)
    amb+=a.sum(); settled+=s.sum()
    bad=s&(tg!=ch)
    if bad.any():
        wrong+=bad.sum(); print('WRONG',seed,it,n,w_out,w_prev,pc,mode,r[bad][:2],tg[bad][:2],ch[bad][:2],flush=True)
        np.save(f'/tmp/fuzz_bad_{seed}_{it}.npy',cls)
print(f'seed {seed} iters {it} amb {amb} settled {settled} wrong {wrong}')
