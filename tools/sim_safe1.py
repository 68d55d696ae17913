#!/usr/bin/env python3
"""What S1 (filtlong_amd/csrc/kmerset.h: safe1) is worth, before it was built: on the synthetic C3 reads, the share of 16-mer windows
the text settles without a lookup — text match, U13 refutation, S1 refutation (exactly one mismatch against the text and no 16-mer one
base away from the text's window is a member) — and the prefilter pair slots the unsettled windows still touch.
Result at 5 Mbp, 300 reads: windows known 0.486, U13 0.108, S1 0.164, left 0.242 (0.406 without S1); pair slots touched per position
0.294 (all windows with a mismatch) -> 0.252 (U13) -> 0.176 (U13 + S1); 10.6 % of the text's windows have a member one base away.
usage: python tools/sim_safe1.py   (CPU, ~2 min)"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from filtlong_amd import synth
rng=np.random.default_rng(1)
G=5_000_000
ref_codes=rng.integers(0,4,G).astype(np.uint8)
ref=np.frombuffer(b"ACGT",dtype=np.uint8)[ref_codes]
# both strands text
def codes_of(a):
    lut=np.zeros(256,np.uint8); lut[ord('C')]=1; lut[ord('G')]=2; lut[ord('T')]=3
    return lut[a]
fw=ref_codes.astype(np.uint64)
def kmer_vals(c,k):
    v=np.zeros(len(c)-k+1,np.uint64)
    for i in range(k):
        v=(v<<np.uint64(2))|c[i:len(c)-k+1+i].astype(np.uint64)
    return v
rc=(3-ref_codes[::-1]).astype(np.uint8)
k16=np.concatenate([kmer_vals(ref_codes,16),kmer_vals(rc,16)])
members=np.unique(k16)
print("members",len(members))
k13f=kmer_vals(ref_codes,13); k13=np.concatenate([k13f,kmer_vals(rc,13)])
u,cnt=np.unique(k13,return_counts=True)
uniq13=u[cnt==1]
u13_flag=np.isin(k13f,uniq13)  # per forward text position
print("U13 frac",u13_flag.mean())
# 12-mers present
k12=np.unique(np.concatenate([kmer_vals(ref_codes,12),kmer_vals(rc,12)]))
pres12=np.zeros(1<<24,bool); pres12[k12.astype(np.int64)]=True
print("12mer density",pres12.mean())
# near1 flag per forward text position: any middle(3..12)/all substitution neighbour is a member
def near1_flags(offsets):
    T=kmer_vals(ref_codes,16)
    flag=np.zeros(len(T),bool)
    for j in offsets:
        sh=np.uint64(2*(15-j))
        for x in (1,2,3):
            nb=T^(np.uint64(x)<<sh)
            idx=np.searchsorted(members,nb); idx[idx>=len(members)]=0
            flag|=members[idx]==nb
    return flag
near_mid=near1_flags(range(3,13))
near_all=near1_flags(range(16))
print("near1 flag set: mid",near_mid.mean(),"all",near_all.mean())
tot=dict(pos=0,known=0,u13=0,n1mid=0,n1all=0,rest=0,rest_all=0,pairs0=0,pairs_u13=0,pairs_mid=0,pairs_all=0)
for r in range(300):
    L=int(synth.lengths(1,first=r)[0]) if hasattr(synth,'lengths') else 10000
    L=min(L,30000)
    if L<100: continue
    seq=synth.seq_read(np.uint64(r),L,ref)
    c=codes_of(seq)
    start=int(synth.mix(synth.SEED,synth.STREAM_START,np.uint64(r),0)%np.uint64(G-L)) if G>L else 0
    mm=(c!=ref_codes[start:start+L])
    n=L-15
    cs=np.concatenate([[0],np.cumsum(mm)])
    wcnt=cs[16:16+n]-cs[:n]           # mismatches in window starting at i
    known=wcnt==0
    # U13: a matching 13-mer at window offsets 0..3 unique
    m13=(cs[13:]-cs[:-13])==0       # 13-window starting at i matches
    uf=u13_flag[start:start+len(m13)]&m13
    has_u13=np.zeros(n,bool)
    for o in range(4):
        has_u13|=uf[o:o+n]
    u13ref=~known&has_u13
    # near1: exactly one mismatch at offset j; text window flag clear
    one=wcnt==1
    # offset of the mismatch
    idx_mm=np.where(mm)[0]
    # for each window with one mismatch find offset
    nxt=np.searchsorted(idx_mm,np.arange(n))
    off=np.where(one,idx_mm[np.minimum(nxt,len(idx_mm)-1)]-np.arange(n),-1) if len(idx_mm) else np.full(n,-1)
    fm=near_mid[start:start+n]; fa=near_all[start:start+n]
    n1mid=one&(off>=3)&(off<=12)&~fm&~u13ref
    n1all=one&~fa&~u13ref
    rest=~known&~u13ref&~n1mid
    rest_all=~known&~u13ref&~n1all
    # prefilter pair-slots touched: window at i uses 12-mers starting at i..i+4 -> pair slot (s//2)
    def pairs(unsettled):
        t=np.zeros(n+8,bool)
        for o in range(5):
            t[np.where(unsettled)[0]+o]=True
        return t.reshape(-1,2)[:,:].any(1).sum() if len(t)%2==0 else t[:-1].reshape(-1,2).any(1).sum()
    tot['pos']+=n; tot['known']+=known.sum(); tot['u13']+=u13ref.sum(); tot['n1mid']+=n1mid.sum(); tot['n1all']+=n1all.sum(); tot['rest']+=rest.sum(); tot['rest_all']+=rest_all.sum()
    tot['pairs0']+=pairs(~known); tot['pairs_u13']+=pairs(~known&~u13ref); tot['pairs_mid']+=pairs(rest); tot['pairs_all']+=pairs(rest_all)
for k,v in tot.items(): print(k, v, round(v/tot['pos'],4))
