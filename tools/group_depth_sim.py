#!/usr/bin/env python3
"""L2 (LRU, 32768 rows) hit rate of the config-4 gather sequence against the depth of the recursive bisection behind the locality groups
(gf_plan.hip locality_groups restated in numpy): 16 groups 0.443, 64 groups 0.437, 256 groups 0.435 -- CPU only: python tools/group_depth_sim.py 4 6 8"""
import sys, os, numpy as np, scipy.sparse as sp, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'graph-neural-networks_amd'))
from alegnn_amd import graphgen
from collections import OrderedDict
N = 100000
A = graphgen.er(N, seed=0).tocsr()
A.data[:] = 1.0
S = ((A + A.T) > 0).astype(np.float64).tocsr()
S.setdiag(0); S.eliminate_zeros()
deg_rows = np.diff(A.indptr)

def lru(seq, cap=32768):
    cache = OrderedDict(); hits = 0
    for j in seq:
        if j in cache:
            cache.move_to_end(j); hits += 1
        else:
            cache[j] = 1
            if len(cache) > cap: cache.popitem(last=False)
    return hits / len(seq)

def bisect(levels, iters=50):
    label = np.zeros(N, dtype=np.int64)
    r, c = S.nonzero()
    for lev in range(levels):
        parts = 1 << lev
        same = label[r] == label[c]
        M = sp.csr_matrix((np.ones(same.sum()), (r[same], c[same])), shape=(N, N))
        d = np.asarray(M.sum(axis=1)).ravel()
        dmax = np.zeros(parts); np.maximum.at(dmax, label, d)
        cnt = np.bincount(label, minlength=parts).astype(np.float64)
        x = np.sin(0.7 * np.arange(N) + 0.3)
        def deflate(z):
            mean = np.bincount(label, weights=z, minlength=parts) / np.maximum(cnt, 1)
            z = z - mean[label]
            nrm = np.sqrt(np.maximum(np.bincount(label, weights=z * z, minlength=parts), 1e-300))
            return z / nrm[label]
        x = deflate(x)
        for it in range(iters):
            y = (dmax[label] - d) * x + M @ x
            x = deflate(y)
        order = np.lexsort((x, label))
        newlab = np.empty(N, dtype=np.int64)
        pos = 0
        for p in range(parts):
            k = int(cnt[p]); idx = order[pos:pos + k]; pos += k
            newlab[idx[:k // 2]] = 2 * p; newlab[idx[k // 2:]] = 2 * p + 1
        label = newlab
    return label

def lp(label, P, sweeps=6):
    label = label.copy(); size = np.bincount(label, minlength=P)
    cap = (N + P - 1) // P * 103 // 100 + 1
    indptr, indices = A.indptr, A.indices
    for s in range(sweeps):
        moved = 0
        for v in range(N):
            lo, hi = indptr[v], indptr[v + 1]
            if lo == hi: continue
            ls = label[indices[lo:hi]]
            vals, cnts = np.unique(ls, return_counts=True)
            cur = label[v]; ccur = cnts[vals == cur].sum()
            b = np.argmax(cnts)   # lowest label among ties (unique sorts ascending)
            if vals[b] != cur and cnts[b] > ccur and size[vals[b]] < cap:
                size[vals[b]] += 1; size[cur] -= 1; label[v] = vals[b]; moved += 1
        if moved == 0: break
    return label

def schedule_seq(label, degsort=True):
    if degsort:
        order = np.lexsort((-deg_rows, label))
    else:
        order = np.argsort(label, kind='stable')
    seq = np.concatenate([A.indices[A.indptr[v]:A.indptr[v + 1]] for v in order])
    # padding of SELL-8 slices formed along this order
    d = deg_rows[order]; pad = 0
    n8 = (len(d) + 7) // 8 * 8
    dd = np.zeros(n8, dtype=np.int64); dd[:len(d)] = d
    slices = dd.reshape(-1, 8)
    padded = (slices.max(axis=1) * 8).sum()
    return seq, padded / d.sum()

if __name__ == '__main__':
    for levels in [int(a) for a in sys.argv[1:]] or [4]:
        t = time.time(); lab = bisect(levels); P = 1 << levels
        r_, c_ = A.nonzero(); cut = (lab[r_] != lab[c_]).mean()
        seq, padr = schedule_seq(lab)
        h = lru(seq)
        print(f"levels {levels} groups {P} spectral only: cut {cut:.3f} lru-hit {h:.3f} padding x{padr:.3f}  ({time.time()-t:.0f}s)", flush=True)
        lab2 = lp(lab, P)
        cut2 = (lab2[r_] != lab2[c_]).mean()
        seq, padr = schedule_seq(lab2)
        print(f"levels {levels} groups {P} + LP:        cut {cut2:.3f} lru-hit {lru(seq):.3f} padding x{padr:.3f}  ({time.time()-t:.0f}s)", flush=True)
