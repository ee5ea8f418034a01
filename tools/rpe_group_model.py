#!/usr/bin/env python3
"""How many groups of equal cell signature do S Morton-consecutive keys of one query hold?  (numpy model of the table-gradient
kernels' grouping, csrc/attn_bwd_box*.hip: a pair's signature = the base cells of its axis taps.)  Decides what a sort over a
larger key range can save: 12.7 groups per 64 keys (5-axis signature of one z-half), 97 per 1024, ~300 per 4096.
    python tools/rpe_group_model.py"""
import numpy as np
rng=np.random.default_rng(0)
pts=rng.uniform([0,0,0],[8,6,3],(40000,3))+1.0
vox=np.unique(np.round(pts/0.04).astype(np.int64),axis=0)
rng.shuffle(vox)
xyz=(vox*0.04).astype(np.float32)
# crude FPS substitute: random subset of 4096 (FPS is more uniform, fine)
keys=xyz[rng.choice(len(xyz),4096,replace=False)]
# morton order
def morton(p):
    q=((p-p.min(0))/(p.max(0)-p.min(0)+1e-9)*1023).astype(np.int64)
    def spread(v):
        r=np.zeros_like(v)
        for i in range(10): r|=((v>>i)&1)<<(3*i)
        return r
    return spread(q[:,0])|(spread(q[:,1])<<1)|(spread(q[:,2])<<2)
keys=keys[np.argsort(morton(keys))]
def base(d):
    L=np.log2(np.abs(d)*512+1)
    pix=np.copysign(L,d)*(1/12*5)+4.5
    return np.clip(np.floor(pix),0,8).astype(np.int64)
nq=64
res={}
for size_mean in (0.5,1.0,2.0):
  for S in (64,128,256,512,1024,4096):
    g5=[];g6=[];g4=[]
    for qi in range(nq):
        c=keys[rng.integers(4096)]+rng.normal(0,0.1,3)
        sz=size_mean*np.exp(rng.normal(0,0.3,3))
        lo=c-sz/2;hi=c+sz/2
        bx0=base(lo[0]-keys[:,0]);bx1=base(hi[0]-keys[:,0])
        by0=base(lo[1]-keys[:,1]);by1=base(hi[1]-keys[:,1])
        bz0=base(lo[2]-keys[:,2]);bz1=base(hi[2]-keys[:,2])
        J5=bz0+16*(by0+16*(by1+16*(bx0+16*bx1)))
        J6=J5*16+bz1
        J4=bz0+16*(bz1+16*(by0+16*by1))
        for t in range(0,4096,S):
            g5.append(len(np.unique(J5[t:t+S])));g6.append(len(np.unique(J6[t:t+S])));g4.append(len(np.unique(J4[t:t+S])))
    print(f"size {size_mean} S={S}: groups/tile 5-axis {np.mean(g5):.1f} 6-axis {np.mean(g6):.1f} zy-only {np.mean(g4):.1f}; per query 5ax {np.mean(g5)*4096/S:.0f} 6ax {np.mean(g6)*4096/S:.0f}")
