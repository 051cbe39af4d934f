#!/usr/bin/env python3
"""Would COHERENT waves make the triclinic distance matrix cheap?  ("Pair distances".)

k_pairdist searches the whole image table for every pair because a wave's pairs point in unrelated directions.  Sorting both
groups in space (Morton order of the fractional coordinates) and working tile by tile -- BI consecutive sorted rows x BJ consecutive
sorted columns -- confines a tile's difference vectors to one box; after one lattice shift only the lattice vectors t with
max over the box of 2 d.t > |t|^2 can still shorten anything.  This script counts them for BASELINE config 3 (1e4 atoms uniform in
the 24 x 23 x 22 nm, 75 / 80 / 70 degree cell), in fractional coordinates (exact for a parallelepiped-shaped box), no GPU needed.

Result (round 3): with the 256-column chunks a wave needs (4 columns per lane) a tile keeps 5.95 signed vectors on average (39 % of the
tiles more than 6), against the 8 +- pairs k_pairdist evaluates with one dot product each -- no saving; 128 columns: 4.3; 64 columns:
3.2.  A chunk of 256 of 1e4 atoms is 1/39 of the cell: too large a box for the short list to be short.  The formulation pays at
>= 1e5 atoms per group (a 40 GB matrix), not at config 3.  A prototype of the whole path (sort, boxes, tile table, LDS-staged
main kernel; parity-green against k_pairdist and an fp64 image search) measured 514 us per config-3 matrix with Cartesian boxes
(98 % of the tiles fell back to the whole-table search) against 174 us for k_pairdist, and was removed."""
import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from groan_rs_amd import workload as W
rng=np.random.default_rng(1)
box=W.c3_box()
L=np.array([[box[0],0,0],[box[5],box[1],0],[box[7],box[8],box[2]]],np.float64)
n=10000
f=rng.random((n,3))
def mkeys(f,G):
    c=np.minimum((f*G).astype(int),G-1)
    def spread(v):
        r=np.zeros_like(v)
        for b in range(4): r|=((v>>b)&1)<<(3*b)
        return r
    return spread(c[:,0])|(spread(c[:,1])<<1)|(spread(c[:,2])<<2)
univ_i=np.array([(i,j,k) for i in range(-2,3) for j in range(-2,3) for k in range(-2,3) if (i,j,k)!=(0,0,0)])
univ=univ_i@L; t2=(univ**2).sum(1)
A=univ@L.T   # A[t,k] = t . a_k
def analyse(k,BI,BJ,label):
    o=np.argsort(k,kind='stable'); p=f[o]
    def boxes(B):
        nb=(n+B-1)//B; lo=np.zeros((nb,3)); hi=np.zeros((nb,3))
        for b in range(nb):
            q=p[b*B:(b+1)*B]; lo[b]=q.min(0); hi[b]=q.max(0)
        return lo,hi
    ilo,ihi=boxes(BI); jlo,jhi=boxes(BJ)
    print(label,'BI',BI,'BJ',BJ,'j frac ext mean',(jhi-jlo).mean(0).round(2),'max',(jhi-jlo).max(0).round(2),'i',(ihi-ilo).mean(0).round(2))
    nbi,nbj=len(ilo),len(jlo)
    sel=[(rng.integers(nbi),rng.integers(nbj)) for _ in range(6000)]
    cnts=[]
    for bi,bj in sel:
        lo=ilo[bi]-jhi[bj]; hi=ihi[bi]-jlo[bj]; c=(lo+hi)/2
        # shift: nearest lattice vector to the Cartesian centre
        cc=c@L
        allv_i=np.array([(i,j,k) for i in range(-2,3) for j in range(-2,3) for k in range(-2,3)])
        Ti=allv_i[((cc-allv_i@L)**2).sum(1).argmin()]
        lo=lo-Ti; hi=hi-Ti
        m=2*(np.maximum(A*lo,A*hi)).sum(1)
        cnts.append((m>t2*(1-1e-5)-1e-3).sum())
    cnts=np.array(cnts); print('   cand count: mean %.2f'%cnts.mean(),'hist',np.bincount(cnts,minlength=14)[:14],' >6: %.3f  >8: %.3f'%((cnts>6).mean(),(cnts>8).mean()))
for BJ in (256,128,64):
    analyse(mkeys(f,16),8,BJ,'morton16')
analyse(mkeys(f,16),4,256,'morton16')
analyse(mkeys(f,16),16,256,'morton16')
