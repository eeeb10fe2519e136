#!/usr/bin/env python3
"""L2 hit rate of the node-major hop's gather stream at config 4, simulated: an LRU cache of 4 MiB / 128 B rows over the sequence of source rows
of one batch entry (destination rows in order).  Natural order: 0.32 (PMC on the production kernel: 0.31); after reverse Cuthill-McKee: 0.39 -- an ER
graph has no locality for a bandwidth-reducing order to find.  What the library does since round 2 (locality_groups in gf_plan.hip: rows only --
label propagation alone 0.38 simulated / 0.37 measured; with the spectral start 0.44 simulated).  CPU only: python tools/l2_lru_sim.py"""
import sys, numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'graph-neural-networks_amd'))
from alegnn_amd import graphgen
from collections import OrderedDict
N=100000
A=graphgen.er(N,seed=0).tocsr()
def hit_rate(A, cap):
    # LRU over the source-row gather sequence of one batch entry, dest rows in order (32 CUs interleave tiles; approximated as in-order)
    seq=A.indices
    cache=OrderedDict(); hits=0
    for j in seq:
        if j in cache:
            cache.move_to_end(j); hits+=1
        else:
            cache[j]=1
            if len(cache)>cap: cache.popitem(last=False)
    return hits/len(seq)
for cap in (24576, 32768):
    print("natural", cap, round(hit_rate(A,cap),3))
p=reverse_cuthill_mckee(A, symmetric_mode=True)
Ap=A[p][:,p].tocsr()
for cap in (24576, 32768):
    print("rcm", cap, round(hit_rate(Ap,cap),3))
# bandwidth stats
r,c=Ap.nonzero(); print("rcm mean |i-j|", np.abs(r-c).mean(), "natural", np.abs(A.nonzero()[0]-A.nonzero()[1]).mean())
