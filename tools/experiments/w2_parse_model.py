"""dev experiment (CPU only): a model of W2's parse (csrc/bam_write.hip: k_bgzf_deflate, pass 1 -- 64 positions per step, an N-way
hash table of 3-byte prefixes that is looked up before the step's positions move in, the distance-1 candidate, one-step lazy
evaluation, a greedy walk over the step's token lengths) with an entropy estimate of the coded size, to see what a change of the
match finder is worth before it is written as a kernel: more ways, and zlib's TOO_FAR rule for short matches.
usage: w2_parse_model.py [members=6]     (config-2 payload from tools/_build/synth_bam)"""
import sys, struct, zlib, math, os, subprocess, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_tmp = tempfile.mkdtemp()
subprocess.check_output([os.path.join(ROOT, "tools", "_build", "synth_bam"), os.path.join(_tmp, "s.bam"), "64", "42", "4"])
data=open(os.path.join(_tmp, "s.bam"),'rb').read()
members=[];o=0
while o<len(data):
    bs=struct.unpack_from('<H',data,o+16)[0]+1
    raw=zlib.decompress(data[o+18:o+bs-8],-15) if bs>28 else b''
    if raw: members.append(raw)
    o+=bs
members=members[2:2+int(sys.argv[1]) if len(sys.argv)>1 else 10]
LB=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LE=[0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE=[1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DE=[0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
import bisect
def lsym(l):
    i=bisect.bisect_right(LB,l)-1
    if l==258: i=28
    return 257+i, LE[i]
def dsym(d):
    i=bisect.bisect_right(DBASE,d)-1
    return i, DE[i]
def mlen(b,a,p,cap):
    l=0
    while l<cap and b[a+l]==b[p+l]: l+=1
    return l
def ent(c):
    t=sum(c.values())
    return -sum(v*math.log2(v/t) for v in c.values()) if t else 0
def parse(b, ways, hbits, instep, far3=1<<30, far4=1<<30):
    n=len(b); tab=[[None]*(1<<hbits) for _ in range(ways)]
    hl=Counter(); hd=Counter(); eb=0; pos=0
    while pos<n:
        lanes=range(pos,min(pos+64,n))
        info=[]
        cands={}
        for p in lanes:
            if p+3<=n:
                v=b[p]|(b[p+1]<<8)|(b[p+2]<<16)
                h=((v*0x9E3779B1)&0xFFFFFFFF)>>(32-hbits)
                cands[p]=(h,[tab[w][h] for w in range(ways)])
        # insert: newest in, shift ways (all lanes of equal hash: the highest lane wins way 0; way k+1 = old way k)
        byh={}
        for p,(h,c) in cands.items(): byh[h]=p
        for h,p in byh.items():
            old=[tab[w][h] for w in range(ways)]
            tab[0][h]=p
            for w in range(1,ways): tab[w][h]=old[w-1]
        best=[]
        for p in lanes:
            bl=0;bd=0
            if p in cands:
                cap=min(258,n-p)
                cl=list(cands[p][1])
                for c in cl:
                    if c is not None and c<p and p-c<=32768 and bl<cap:
                        l=mlen(b,c,p,cap)
                        if l>=3 and l>bl and not (l==3 and p-c>far3) and not (l==4 and p-c>far4): bl=l;bd=p-c
                for d in ([1]+instep):
                    if p>=d and bl<cap:
                        l=mlen(b,p-d,p,cap)
                        if l>=3 and l>bl: bl=l;bd=d
            best.append((bl,bd))
        # lazy
        bl2=[x[0] for x in best]
        for i in range(len(best)-1):
            if best[i][0]>=3 and bl2[i+1]>best[i][0]: best[i]=(0,0)
        k=0
        while k<len(best):
            l,d=best[k]
            if l>=3:
                s,e=lsym(l); hl[s]+=1; ds,de=dsym(d); hd[ds]+=1; eb+=e+de; k+=l
            else:
                hl[b[pos+k]]+=1; k+=1
        pos+=k
    hl[256]+=1
    bits=ent(hl)+ent(hd)+eb+ 14+19*3+ (len(hl)+len(hd))*4  # rough header
    return bits/8
tot=sum(len(m) for m in members)
for name,(ways,hb,f3,f4) in {"2way":(2,12,1<<30,1<<30),"2way far3=4096":(2,12,4096,1<<30),"2way far3=1024":(2,12,1024,1<<30),"2way far3=256 far4=8192":(2,12,256,8192),"4way far3=1024":(4,12,1024,1<<30),"4way far3=256 far4=8192":(4,12,256,8192),"8way far3=256 far4=8192":(8,12,256,8192)}.items():
    c=sum(parse(m,ways,hb,[],f3,f4) for m in members)
    print(name, round(c/tot,4), flush=True)
