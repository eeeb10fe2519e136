#!/usr/bin/env python3
"""Bound for the source-blocked LDS hop at N > 10239 (source blocks of <= 10k nodes staged in LDS).

Design family: a workgroup holds one block of SOURCE rows of a c-column panel in LDS (160 KB => 40960 / c nodes per block), its threads own
destination rows (accumulators in registers, so no atomics and a fixed summation order) and walk, block after block, the entries of their
rows whose source lies in the staged block.  A wave executes max-over-its-64-rows entries per (row slot, block): the ELL fill below is
nnz / executed slots with the best row grouping we found (rows with equal per-block entry counts share a wave: lexicographic sort of the count
vectors; plain degree sort for comparison).  CPU only (numpy); run: python tools/blocked_fill.py > profiles/r02_d_blocked/fill.log

Traffic model per hop (config 4: B*G = 4096 columns, 4 bytes per entry of the signal):
  source blocks re-read once per destination range  : ceil(N / rows per workgroup) x the signal
  index stream (2 B per executed slot) per panel     : nnz / fill x 2 B x (4096 / c) panels
against today's 16.4 GB of gathered 128-byte rows.  The column 'ms at 10 TB/s' prices the total at the best rate this chip has
delivered for a stream of this size that does not fit L2 (the production kernel's gathers: 8.6 TB/s, the L2-window prototype: 10.3 TB/s).
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "graph-neural-networks_amd"))
from alegnn_amd import graphgen

N, COLS, L2_RATE = 100_000, 128 * 32, 10e12
A = graphgen.er(N, seed=0)
rows = np.repeat(np.arange(N), np.diff(A.indptr))
signal = N * COLS * 4
print(f"ER N={N} nnz={A.nnz}; signal {signal/1e9:.2f} GB per tap; today: {A.nnz*COLS*4/1e9:.1f} GB of 128-byte gathers, 1.90 ms per hop")
print("cols/panel  nodes/block  blocks  fill(lexsort)  fill(degree)  rows/WG  src re-reads  src GB  index GB  total GB  ms at 10 TB/s")
for c in (1, 2, 4):
    blk = 40960 // c
    nb = -(-N // blk)
    blk = -(-N // nb)                                   # equal blocks
    cnt = np.zeros((N, nb), np.int32)
    np.add.at(cnt, (rows, A.indices // blk), 1)
    pad = (-N) % 64
    def fill(order):
        cc = np.vstack([cnt[order], np.zeros((pad, nb), np.int32)])
        return A.nnz / (cc.reshape(-1, 64, nb).max(1).sum() * 64)
    f_lex = fill(np.lexsort([cnt[:, j] for j in range(nb - 1, -1, -1)] + [cnt.sum(1)]))
    f_deg = fill(np.argsort(cnt.sum(1), kind="stable"))
    rows_wg = 1024 * (64 // c)                         # 1024 threads x 64 accumulator VGPRs (half of the 128 a thread has at that occupancy)
    rereads = -(-N // rows_wg)
    src = rereads * signal
    idx = A.nnz / f_lex * 2 * (COLS // c)
    tot = src + idx + signal                           # + the output once
    print(f"{c:10d}  {blk:11d}  {nb:6d}  {f_lex:13.3f}  {f_deg:12.3f}  {rows_wg:7d}  {rereads:12d}  {src/1e9:6.2f}  {idx/1e9:8.2f}  {tot/1e9:8.2f}  {tot/L2_RATE*1e3:6.2f}")
