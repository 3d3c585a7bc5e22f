"""HOSTILE device arrays through the kernel phases on the CPU, under AddressSanitizer (run by tests/test_sim_kernels.py in a
child process with libasan preloaded).  The same cases tools/fuzz_gpu.py: fuzz_hostile makes on the GPU -- garbage block
offsets, descending pairs, coded-block indices and per-stream bases far outside the pool, with the bounds a CHECKED launch of
h263mi_batch_decode_events has (n_events and pool size = what the caller's allocations hold) -- plus the two values that
faulted on the MI355X in round 6: a block offset of 0xffffffff (`first + lane` wrapped) and a base of 2^64 - 2^20.  Exact-size
numpy arrays stand for the caller's allocations: one word beyond them is a report.
usage: python hostile_replay.py <seed> <cases>"""
import sys; sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/h263-rs_amd']
import numpy as np, ctypes as C, simlib, recgen, h263mi
from oracle import oracle as orc
lib=simlib.lib(asan=True)
lib.sim_set_n_events.argtypes=[C.c_uint32]
def device_decode(w,h,n,mbs,first,ev,base,n_events_alloc,pool_alloc):
    """what the launch of h263mi_batch_decode_events with NO sizes given reads: bounds from the allocations"""
    L=simlib.layout(w,h)
    cur=np.full(L.frame_bytes*n,0xC3,np.uint8); status=np.zeros(n,np.uint32)
    lib.sim_set_n_events(n_events_alloc)
    rc=lib.sim_recon_ex(w,h,n,simlib._p(mbs),simlib._p(np.zeros((1,64),np.int16)),pool_alloc,simlib._p(base),None,0,simlib._p(cur),simlib._p(status),simlib._p(first),simlib._p(ev))
    assert rc==0
    return status
seed=int(sys.argv[1]) if len(sys.argv)>1 else 6
rng=np.random.default_rng(seed)
cases=0
while cases<int(sys.argv[2]) if len(sys.argv)>2 else cases<3:
    w = int(rng.choice([rng.integers(1, 64), rng.integers(1, 420), 16 * rng.integers(1, 30), 176, 352, rng.integers(260, 800), 4 * rng.integers(65, 200)]))
    h = int(rng.choice([rng.integers(1, 64), rng.integers(1, 300), 16 * rng.integers(1, 20), 144, 288, rng.integers(64, 420), 4 * rng.integers(16, 100)]))
    rng.random(); rng.random()
    w,h=min(max(w,17),300),min(max(h,17),200)
    n=int(rng.choice([1,2,5,8])); pipeline=bool(rng.integers(0,2))
    recs,at,base=[],0,[]
    for s_ in range(n):
        m,c=recgen.intra_picture(w,h,seed=int(rng.integers(0,1<<30)),max_level=int(rng.choice([40,1023])))
        recs.append((simlib.pad_records(m,w,h),c)); base.append(at); at+=len(c)
    mbs=np.concatenate([r[0] for r in recs]); co=np.concatenate([r[1] for r in recs])
    first,ev=h263mi.events_from_dense(co,np.ones(len(co),bool))
    evd = ev if len(ev) else np.zeros(4,np.uint32)
    print("case",cases,"w",w,"h",h,"n",n,"pipeline",pipeline,"blocks",at,"events",len(ev),flush=True)
    # exact-size copies (ASan sees one word past them)
    st=device_decode(w,h,n,mbs.copy(),first.copy(),evd.copy(),np.array(base,np.uint64),len(evd),len(first)-1)
    assert not st.any(), ("clean rejected",st)
    per=len(mbs)//n
    for attempt in range(int(rng.integers(2,6))):
        hit=set(int(v) for v in rng.choice(n,size=int(rng.integers(1,n+1)),replace=False))
        m2,f2,b2=mbs.copy(),first.copy(),np.array(base,np.uint64)
        kinds=[]
        for s_ in hit:
            lo,hi=base[s_],(base[s_+1] if s_+1<n else at)
            kind=int(rng.integers(0,5))
            junk=[0xffffffff,0xfffffffe,0xfffffff9,0xfffffff0,len(ev),len(ev)+1,len(ev)+64,1<<28,0x7fffffff]
            if kind in (0,1,4) and hi-lo<4: kind=2
            if kind==0:
                k=rng.integers(lo+1,hi,size=min(4,hi-lo-1)); f2[k]=rng.choice(junk,size=len(k))
            elif kind==1:
                k=int(rng.integers(lo+1,hi-1)); f2[k],f2[k+1]=f2[k+1]+7,f2[k]
            elif kind==2:
                k=s_*per+rng.integers(0,per,size=3); m2["coeff_index"][k]=rng.choice([at,at+1,1<<24,0xffffffff],size=3); m2["cbp"][k]|=1
            elif kind==3:
                b2[s_]=[at,at+5,1<<40,(1<<64)-(1<<20),(1<<64)-1][int(rng.integers(0,5))]
            else:
                f2[lo+1:hi]=(f2[lo+1:hi].astype(np.uint64)+int(rng.choice([len(ev),1<<27]))).astype(np.uint32)
            kinds.append((s_,kind))
        print("  attempt",attempt,"hit",kinds,flush=True)
        st=device_decode(w,h,n,m2,f2,evd.copy(),b2,len(evd),len(first)-1)
        for s_ in range(n):
            if s_ not in hit: assert st[s_]==0,(s_,st)
        st=device_decode(w,h,n,mbs.copy(),first.copy(),evd.copy(),np.array(base,np.uint64),len(evd),len(first)-1)
        assert not st.any()
    cases+=1
print("hostile-cpu-ok")
