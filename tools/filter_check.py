#!/usr/bin/env python3
"""The opt-in wide filter (RIHIP_FILTER_WIDE=1) against the default 256-query filter on one index, with the pattern of any
survivors it loses (experiments on variants of the kernel: RIHIP_LIB=<build> python tools/filter_check.py)."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from recommendit_amd import _lib as L  # noqa: E402

if os.environ.get("RIHIP_LIB"):
    L.LIB_PATH = Path(os.environ["RIHIP_LIB"]).resolve()
from recommendit_amd import FAISSIndex  # noqa: E402

rng = np.random.RandomState(21)
N, d, nq, k = 150_003, 128, 700, 500
X = rng.randn(N, d).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
Q = rng.randn(nq, d).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
idx = FAISSIndex(embed_dim=d, exact=True)
idx.build_ivf_index(X, list(range(N)))
for rep in range(2):
    sn, rn = idx.batch_search(Q, k=k)          # 256-query filter (default)
    os.environ["RIHIP_FILTER_WIDE"] = "1"
    sw, rw = idx.batch_search(Q, k=k)          # scan_bf16_wide_kernel
    del os.environ["RIHIP_FILTER_WIDE"]
    bad = [q for q in range(nq) if set(rw[q]) != set(rn[q])]
    print(f"rep {rep}: queries with different result sets: {len(bad)}", bad[:16], "waves", sorted(set(q // 128 for q in bad)))
from collections import Counter
ns = 256          # splits of this launch: one 1 024-query block -> 256 workgroups
n_seq = (N + 63) // 64
per = (n_seq + ns - 1) // ns
c_stage = Counter(); c_mod = Counter(); c_sub = Counter(); c_wave = Counter()
for q in bad:
    for m in set(rn[q]) - set(rw[q]):
        st = m // 64
        c_stage[st % per] += 1          # stage index inside its split
        c_sub[(m % 64) // 32] += 1
        c_mod[m % 32] += 1
        c_wave[q // 128] += 1
print("per", per, "missing by stage-in-split:", sorted(c_stage.items()))
print("by sub-tile:", sorted(c_sub.items()), "by wave:", sorted(c_wave.items()))
print("by row mod 32:", sorted(c_mod.items()))
ranks = []
for q in bad:
    for m in set(rn[q]) - set(rw[q]):
        ranks.append(int(np.where(rn[q] == m)[0][0]))
print("ranks of the missing rows in the reference result:", sorted(ranks)[:60])
