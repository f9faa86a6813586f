"""Soak test on the GPU box: random rasters of a few hundred to 1600 pixels a side (all data types, ragged sizes, masks, nDepth,
raw blocks) through the product library and the oracle for 150 s; prints every mismatch.  Round 1: 7 600 cases, none.
    gpurun -- 'python tools/fuzz_against_oracle.py [seed] [seconds] [bytes]'      (bytes: 8-bit data types only)"""
import sys, os, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, capi, cases
P=capi.product(); O=capi.oracle()
def same(a,b):
    if a is None or b is None: return a is None and b is None
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
t0=time.time(); n=0
budget=float(sys.argv[2]) if len(sys.argv)>2 else 150
only_bytes=len(sys.argv)>3 and sys.argv[3]=='bytes'
while time.time()-t0 < budget:
    dt=cases.ALL_DTYPES[rng.integers(0,8)] if not only_bytes else (np.uint8 if rng.random()<0.6 else np.int8)
    r,c=int(rng.integers(200,1600)),int(rng.integers(200,1600))
    if rng.random()<0.4: r-=r%8; c-=c%8
    nd=int(rng.choice([1,1,1,2,3]))
    kind=np.dtype(dt).kind
    x=cases.terrain(r,c,rng,amp=float(rng.choice([5,50,500])),base=float(rng.choice([0,100,1000])),sigma=float(rng.choice([0,0.3,3])))
    style=rng.integers(0,5)
    if style==1: x=np.floor(x/16)*16
    if style==2: x=np.round(x,1)
    if style==3: x[::9,::7]*=1e6
    x=np.stack([x+k for k in range(nd)],axis=-1) if nd>1 else x
    if np.dtype(dt).itemsize==1: x=x/8
    x=cases._cast(x,dt)
    e=float(rng.choice([0,0.001,0.01,0.5,1,3])) if kind=='f' else float(rng.choice([0,0,1,4]))
    kw=dict(n_depth=nd)
    if rng.random()<0.5:
        m=np.ones((r,c),np.uint8)
        for _ in range(int(rng.integers(1,30))):
            i0,j0=int(rng.integers(0,r)),int(rng.integers(0,c)); m[i0:i0+int(rng.integers(1,200)), j0:j0+int(rng.integers(1,200))]=0
        if rng.random()<0.5: m&=(rng.random((r,c))>0.02).astype(np.uint8)
        kw['mask']=m
    tag=f"{np.dtype(dt).name} {r}x{c}x{nd} e={e} style={style} mask={'mask' in kw}"
    r1,b1=O.encode(x,e,**kw); r2,b2=P.encode(x,e,**kw)
    if r1!=r2 or (b1!=b2 and not (kind=='f' and e==0)):
        print("ENC MISMATCH",tag,r1,r2,len(b1),len(b2)); continue
    if r1==0:
        d1,d2=O.decode(b1),P.decode(b1)
        if d1[0]!=d2[0] or not same(d1[1],d2[1]) or not same(d1[2],d2[2]): print("DEC MISMATCH",tag,d1[0],d2[0])
    n+=1
print("cases",n,"done")
