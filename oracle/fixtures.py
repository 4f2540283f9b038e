"""Seeded synthetic parameters / inputs shared by the golden generator, the tests
and the bench (TEST INFRASTRUCTURE + synthetic-data helpers; no reference code).

np.random.RandomState (legacy MT19937) streams are stable across NumPy versions,
so fixtures can store a seed + checksum instead of megabytes of weights.
"""
from __future__ import annotations

import hashlib
import math
from typing import Dict

import numpy as np

F32 = np.float32
N_GENRES = 18


def xavier_uniform(rng, rows, cols):
    a = math.sqrt(6.0 / (rows + cols))
    return rng.uniform(-a, a, size=(rows, cols)).astype(F32)


def linear_init(rng, out_f, in_f):
    # torch nn.Linear default: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(in), 1/sqrt(in)) for W and b
    b = 1.0 / math.sqrt(in_f)
    return (rng.uniform(-b, b, size=(out_f, in_f)).astype(F32), rng.uniform(-b, b, size=(out_f,)).astype(F32))


def make_state(n_users: int, n_items: int, d: int, H: int, seed: int) -> Dict[str, np.ndarray]:
    """state_dict-keyed parameters (keys = two_tower.py:28-33,57-62 nn.Sequential names)."""
    rng = np.random.RandomState(seed)
    sd = {}
    sd["user_tower.embedding.weight"] = xavier_uniform(rng, n_users + 1, d)
    sd["user_tower.mlp.0.weight"], sd["user_tower.mlp.0.bias"] = linear_init(rng, H, d)
    sd["user_tower.mlp.3.weight"], sd["user_tower.mlp.3.bias"] = linear_init(rng, d, H)
    sd["item_tower.embedding.weight"] = xavier_uniform(rng, n_items + 1, d)
    sd["item_tower.mlp.0.weight"], sd["item_tower.mlp.0.bias"] = linear_init(rng, H, d + N_GENRES)
    sd["item_tower.mlp.3.weight"], sd["item_tower.mlp.3.bias"] = linear_init(rng, d, H)
    return sd


PARAM_ORDER = [
    "user_tower.embedding.weight", "user_tower.mlp.0.weight", "user_tower.mlp.0.bias",
    "user_tower.mlp.3.weight", "user_tower.mlp.3.bias",
    "item_tower.embedding.weight", "item_tower.mlp.0.weight", "item_tower.mlp.0.bias",
    "item_tower.mlp.3.weight", "item_tower.mlp.3.bias",
]


def state_checksum(sd: Dict[str, np.ndarray]) -> str:
    h = hashlib.sha256()
    for k in PARAM_ORDER:
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()


def make_batch(n_users: int, n_items: int, B: int, seed: int, boundary: bool = True):
    rng = np.random.RandomState(seed)
    u = rng.randint(1, n_users + 1, size=B).astype(np.int64)
    p = rng.randint(1, n_items + 1, size=B).astype(np.int64)
    n = rng.randint(1, n_items + 1, size=B).astype(np.int64)
    if boundary and B >= 2:   # boundary ids 1 and n
        u[0], u[-1] = 1, n_users
        p[0], p[-1] = 1, n_items
        n[0], n[-1] = n_items, 1
    gp = (rng.rand(B, N_GENRES) < 0.12).astype(F32)
    gn = (rng.rand(B, N_GENRES) < 0.12).astype(F32)
    return u, p, gp, n, gn


def unit_rows(rng, n, d):
    x = rng.randn(n, d).astype(F32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(F32)
