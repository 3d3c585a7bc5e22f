"""GARBAGE record fields and event words through the kernel phases on the CPU, under AddressSanitizer (run by
tests/test_sim_kernels.py in a child process with libasan preloaded).  tests/sim/hostile_replay.py attacks WHERE the waves read
(offsets, bases, block indices); this one attacks WHAT they read: macroblock types 6..255, quantisers 0 and 32..255, every
coded-block-pattern and kill byte, vectors over the whole int16 range, INTRADC codes 0 and 128 (types.rs:930-936: never in a
stream), and event words with positions beyond 63 and levels over the whole int16 range -- in P pictures that predict from a
reference frame and in I pictures, dense and event transport, with the bounds a CHECKED launch has.  The reference cannot be
given such input (its parser never makes it), so there is no expected picture for a stream that was hit: the contract is that
nothing outside the caller's arrays and the frame store is touched, and that the streams left alone decode exactly.
usage: python garbage_fields_replay.py <seed> <cases>"""
import sys; sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/h263-rs_amd']
import numpy as np, ctypes as C, simlib, recgen, h263mi
from oracle import oracle as orc
lib=simlib.lib(asan=True)
lib.sim_set_n_events.argtypes=[C.c_uint32]
def device_decode(w,h,n,mbs,co,first,ev,base,ref):
    L=simlib.layout(w,h)
    cur=np.full(L.frame_bytes*n,0xC3,np.uint8); status=np.zeros(n,np.uint32)
    if first is not None:
        lib.sim_set_n_events(len(ev))
        rc=lib.sim_recon_ex(w,h,n,simlib._p(mbs),simlib._p(np.zeros((1,64),np.int16)),len(first)-1,simlib._p(base),simlib._p(ref),1 if ref is not None else 0,
                            simlib._p(cur),simlib._p(status),simlib._p(first),simlib._p(ev))
    else:
        rc=lib.sim_recon(w,h,n,simlib._p(mbs),simlib._p(co),len(co),simlib._p(base),simlib._p(ref),1 if ref is not None else 0,simlib._p(cur),simlib._p(status))
    assert rc==0
    return status,[simlib.unpack_frame(L,cur[s*L.frame_bytes:(s+1)*L.frame_bytes]) for s in range(n)]
seed=int(sys.argv[1]) if len(sys.argv)>1 else 11
n_cases=int(sys.argv[2]) if len(sys.argv)>2 else 3
rng=np.random.default_rng(seed)
for case in range(n_cases):
    w=int(rng.choice([17,33,48,100,176,255,300])); h=int(rng.choice([17,32,50,144,200]))
    n=int(rng.choice([1,2,5])); inter=bool(rng.integers(0,2)); events=bool(rng.integers(0,2))
    L=simlib.layout(w,h)
    recs,at,base,refs,want=[],0,[],[],[]
    for s_ in range(n):
        ref=None
        if inter:
            m0,c0=recgen.intra_picture(w,h,seed=int(rng.integers(0,1<<30)),max_level=40)
            rc,ref=orc.decode_picture(w,h,m0,c0,None); assert rc==0
            m,c=recgen.inter_picture(w,h,seed=int(rng.integers(0,1<<30)),mv_range=32,p_4v=0.3,p_intra=0.15,p_coded=0.5,quant=0,max_level=60,sparse_low=False)
        else:
            m,c=recgen.intra_picture(w,h,seed=int(rng.integers(0,1<<30)),max_level=int(rng.choice([40,1023])))
        rc,out=orc.decode_picture(w,h,m,c,ref); assert rc==0
        m=simlib.pad_records(m,w,h)
        bi=np.zeros(len(c),bool)
        for r in m:
            if int(r["mb_type"]) in (3,4):
                k=int(r["coeff_index"]); bi[k:k+bin(int(r["cbp"])).count("1")]=True
        recs.append((m,c,bi)); refs.append(ref); want.append(out); base.append(at); at+=len(c)
    mbs=np.concatenate([r[0] for r in recs]); co=np.concatenate([r[1] for r in recs]) if at else np.zeros((0,64),np.int16)
    reff=np.concatenate([simlib.pack_frame(L,r) for r in refs]) if inter else None
    cod=co if at else np.zeros((1,64),np.int16)
    first=ev=None
    if events:
        first,ev=h263mi.events_from_dense(co,np.concatenate([r[2] for r in recs]) if at else None)
        ev=ev if len(ev) else np.zeros(4,np.uint32)
    print("case",case,"w",w,"h",h,"n",n,"inter",inter,"events",events,"blocks",at,flush=True)
    b0=np.array(base,np.uint64)
    st,got=device_decode(w,h,n,mbs.copy(),cod.copy(),None if first is None else first.copy(),None if ev is None else ev.copy(),b0.copy(),reff)
    assert not st.any(),("clean rejected",st)
    for s_ in range(n):
        for g,e in zip(got[s_],want[s_]): assert (g==e).all(),("clean differs",s_)
    per=len(mbs)//n
    for attempt in range(4):
        hit=set(int(v) for v in rng.choice(n,size=int(rng.integers(1,n+1)),replace=False))
        m2=mbs.copy(); e2=None if ev is None else ev.copy(); c2=cod.copy(); kinds=[]
        for s_ in hit:
            kind=int(rng.integers(0,7)); kinds.append((s_,kind))
            k=s_*per+rng.integers(0,per,size=max(1,per//3))
            lo,hi=base[s_],(base[s_+1] if s_+1<n else at)
            if kind==0: m2["mb_type"][k]=rng.integers(6,256,size=len(k))
            elif kind==1: m2["quant"][k]=rng.choice([0,32,127,128,255],size=len(k))
            elif kind==2:
                # (more blocks than the macroblock has: the coded blocks then run into the next macroblock's, or past the pool)
                m2["cbp"][k]=rng.integers(0,256,size=len(k)); m2["kill"][k]=rng.integers(0,256,size=len(k))
            elif kind==3: m2["mv"][k]=rng.choice([-32768,-32767,-4097,-1025,1024,4096,32766,32767],size=(len(k),4,2))
            elif kind==4: m2["intradc"][k]=rng.choice([0,128,255],size=(len(k),6)); m2["reserved"][k]=255
            elif kind==5 and e2 is not None and first[hi]>first[lo]:
                j=rng.integers(int(first[lo]),int(first[hi]),size=16)
                e2[j]=rng.integers(0,1<<32,size=16,dtype=np.uint64).astype(np.uint32)
            elif hi>lo:
                j=rng.integers(lo,hi,size=4); c2[j]=rng.integers(-32768,32768,size=(4,64))
        print("  attempt",attempt,"hit",kinds,flush=True)
        st,got=device_decode(w,h,n,m2,c2,None if first is None else first.copy(),e2,b0.copy(),reff)
        for s_ in range(n):
            if s_ in hit: continue
            assert st[s_]==0,(s_,st)
            for g,e in zip(got[s_],want[s_]): assert (g==e).all(),("untouched stream differs",s_)
print("garbage-fields-cpu-ok")
