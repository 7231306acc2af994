#!/usr/bin/env python3
"""Fuzz of the parallel gzip reader (krepp_amd/csrc/kr_pgz.inc) against the plain file: random records (lengths, alphabets, quality
entropy, names), random compression level / strategy / memLevel, one or two members, random chunk size, thread count and batch
size.  usage: fuzz_gzip.py <seed> <seconds>   (run it in a directory of its own: it rewrites p.fq / g.fq.gz there)
Round 4: 3,100 iterations, 11,600 chunks handed out with their records parsed, no mismatch."""
import os, sys, gzip, zlib, struct, numpy as np, time
sys.path.insert(0,'/root/repo')
from krepp_amd import capi
os.environ['KR_FASTX_PAR_MIN']='0'
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
def same(a,b): return a[0]==b[0] and np.array_equal(a[1],b[1]) and np.array_equal(a[2],b[2])
t0=time.time(); it=0; parsed_total=0
while time.time()-t0 < float(sys.argv[2] if len(sys.argv)>2 else 120):
    it+=1
    n=int(rng.integers(500,6000))
    qa=int(rng.integers(2,94)); la=[b"ACGT",b"ACGTN",b"ACGTRYKMSWBDHVN"][int(rng.integers(0,3))]
    recs=[]
    fixed=rng.random()<0.4
    for i in range(n):
        L=151 if fixed else int(rng.integers(1,500))
        s=rng.choice(np.frombuffer(la,np.uint8),L).tobytes()
        if rng.random()<0.5: q=bytes((33+rng.integers(0,qa,L)).astype(np.uint8))
        else: q=bytes([33+int(rng.integers(0,qa))])*L
        nm=b"@%s%d %s\n"%([b"r",b"SRR1.",b"M01:2:000-X:1:1101:"][i%3], i, b"c"*int(rng.integers(0,30)))
        recs.append(nm+s+b"\n+\n"+q+b"\n")
    data=b"".join(recs)
    lvl=int(rng.integers(1,10)); strat=[zlib.Z_DEFAULT_STRATEGY,zlib.Z_FILTERED,zlib.Z_RLE,zlib.Z_HUFFMAN_ONLY,zlib.Z_FIXED][int(rng.integers(0,5))] if rng.random()<0.4 else zlib.Z_DEFAULT_STRATEGY
    memlevel=int(rng.integers(1,10))
    co=zlib.compressobj(lvl,zlib.DEFLATED,-15,memlevel,strat)
    cut=int(rng.integers(0,len(data))) if rng.random()<0.3 else None
    def member(d):
        c=zlib.compressobj(lvl,zlib.DEFLATED,-15,memlevel,strat); b=c.compress(d)+c.flush()
        return b"\x1f\x8b\x08\x00"+b"\x00"*6+b+struct.pack("<II",zlib.crc32(d)&0xffffffff,len(d)&0xffffffff)
    blob=member(data) if cut is None else member(data[:cut])+member(data[cut:])
    open('p.fq','wb').write(data); open('g.fq.gz','wb').write(blob)
    os.environ['KR_FASTX_THREADS']='0'
    want=capi.read_fastx('p.fq',min_bases=int(rng.integers(1000,400000)))
    os.environ['KR_FASTX_THREADS']=str(int(rng.integers(1,7)))
    os.environ['KR_PGZ_CHUNK']=str(int(rng.integers(65536,400000)))
    st={}
    got=capi.read_fastx('g.fq.gz',min_bases=int(rng.integers(1000,400000)),stats=st)
    parsed_total+=st['gzip_chunks']['parsed']
    if not same(got,want):
        print("MISMATCH",it,lvl,strat,memlevel,cut,os.environ['KR_PGZ_CHUNK'],st); os.rename('g.fq.gz','bad_%d.fq.gz'%it); os.rename('p.fq','bad_%d.fq'%it); break
print("iterations",it,"chunks parsed",parsed_total)
