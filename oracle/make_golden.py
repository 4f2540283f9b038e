#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference (THIS CONTAINER ONLY).

Run:  python oracle/make_golden.py            (needs /root/reference)

Only src/models/two_tower.py of the reference is imported (torch/numpy only);
it is the oracle-of-the-oracle for towers, both losses and -- through autograd
-- every gradient and the clip+Adam+cosine loop (stock torch.optim).  The
outputs written here are DATA (inputs/expected outputs); no reference source
travels.  Vectors: G1..G6 of SURVEY.md §8c.
"""
import importlib.util
import math
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import fixtures as fx  # noqa: E402

REF = Path(os.environ.get("RECOMMENDIT_REFERENCE", "/root/reference"))
OUT = ROOT / "tests" / "golden"


def load_ref_two_tower():
    spec = importlib.util.spec_from_file_location("ref_two_tower", REF / "src/models/two_tower.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build(ref, n_users, n_items, d, H, seed, dropout=0.0):
    m = ref.TwoTowerModel(n_users, n_items, embed_dim=d, hidden_dim=H, dropout=dropout)
    sd = fx.make_state(n_users, n_items, d, H, seed)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return m, sd


def t(x):
    return torch.from_numpy(np.asarray(x))


def main():
    torch.set_num_threads(1)
    torch.manual_seed(0)
    ref = load_ref_two_tower()
    OUT.mkdir(parents=True, exist_ok=True)

    # ---------------- G1: tower forward (eval) --------------------------------
    g1 = {}
    for tag, (nu, ni, d, H, seed) in {"small": (100, 200, 32, 64, 11), "ml1m": (6040, 3952, 64, 128, 12),
                                      "d128": (500, 700, 128, 128, 13)}.items():
        m, sd = build(ref, nu, ni, d, H, seed)
        m.eval()
        g1[f"{tag}_cfg"] = np.array([nu, ni, d, H, seed], dtype=np.int64)
        g1[f"{tag}_sha"] = np.frombuffer(bytes.fromhex(fx.state_checksum(sd)), dtype=np.uint8)
        for B in (1, 16, 256):
            u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=100 + B)
            with torch.no_grad():
                U = m.user_tower(t(u)).numpy()
                P = m.item_tower(t(p), t(gp)).numpy()
            g1[f"{tag}_B{B}_U"] = U
            g1[f"{tag}_B{B}_P"] = P
    np.savez_compressed(OUT / "g1_tower_forward.npz", **g1)

    # ---------------- G2: bpr_loss + grads of all 10 params --------------------
    g2 = {}
    for tag, (nu, ni, d, H, seed, B) in {"small": (100, 200, 32, 64, 21, 16), "mid": (300, 400, 64, 128, 22, 64)}.items():
        m, sd = build(ref, nu, ni, d, H, seed)
        m.train()  # dropout p=0 -> identity
        u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=200 + B)
        U = m.user_tower(t(u)); P = m.item_tower(t(p), t(gp)); N = m.item_tower(t(n), t(gn))
        U.retain_grad(); P.retain_grad(); N.retain_grad()
        loss = m.bpr_loss(U, P, N)
        loss.backward()
        g2[f"{tag}_cfg"] = np.array([nu, ni, d, H, seed, B], dtype=np.int64)
        g2[f"{tag}_loss"] = np.array(loss.item(), dtype=np.float64)
        g2[f"{tag}_dU"] = U.grad.numpy(); g2[f"{tag}_dP"] = P.grad.numpy(); g2[f"{tag}_dN"] = N.grad.numpy()
        for k, prm in m.named_parameters():
            g2[f"{tag}_grad_{k}"] = prm.grad.numpy()
    np.savez_compressed(OUT / "g2_bpr_grads.npz", **g2)

    # ---------------- G3: in_batch_bpr_loss value + grads ---------------------
    g3 = {}
    m, _ = build(ref, 10, 10, 32, 64, 31)
    for B, d in ((2, 32), (16, 32), (256, 64), (96, 128)):
        rng = np.random.RandomState(300 + B)
        U0 = fx.unit_rows(rng, B, d); I0 = fx.unit_rows(rng, B, d)
        U = t(U0).clone().requires_grad_(True); I = t(I0).clone().requires_grad_(True)
        loss = m.in_batch_bpr_loss(U, I)
        loss.backward()
        g3[f"B{B}_U"] = U0; g3[f"B{B}_I"] = I0
        g3[f"B{B}_loss"] = np.array(loss.item(), dtype=np.float64)
        g3[f"B{B}_dU"] = U.grad.numpy(); g3[f"B{B}_dI"] = I.grad.numpy()
    np.savez_compressed(OUT / "g3_inbatch.npz", **g3)

    # ---------------- G4: 50 optimiser steps (clip + Adam+L2 + cosine) --------
    nu, ni, d, H, seed, B = 100, 200, 32, 64, 41, 16
    m, sd = build(ref, nu, ni, d, H, seed)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=2)
    losses, lrs = [], []
    for step in range(50):
        u, p, gp, n, gn = fx.make_batch(nu, ni, B, seed=4000 + step, boundary=False)
        U = m.user_tower(t(u)); P = m.item_tower(t(p), t(gp)); N = m.item_tower(t(n), t(gn))
        loss = m.bpr_loss(U, P, N)
        opt.zero_grad(); loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)
        opt.step()
        losses.append(loss.item()); lrs.append(opt.param_groups[0]["lr"])
        if step == 24:
            sched.step()  # "epoch" boundary
    g4 = {"cfg": np.array([nu, ni, d, H, seed, B], dtype=np.int64), "losses": np.array(losses), "lrs": np.array(lrs)}
    for k, prm in m.named_parameters():
        g4[f"final_{k}"] = prm.detach().numpy()
    np.savez_compressed(OUT / "g4_train50.npz", **g4)

    # ---------------- G5: inference helpers -----------------------------------
    nu, ni, d, H, seed = 100, 1200, 32, 64, 51
    m, sd = build(ref, nu, ni, d, H, seed)
    item_ids = list(range(1, 1001))
    genres = (np.random.RandomState(52).rand(1000, fx.N_GENRES) < 0.15).astype(np.float32)
    g5 = {"cfg": np.array([nu, ni, d, H, seed], dtype=np.int64), "genres": genres,
          "item_embs": m.get_item_embeddings(item_ids, genres), "user7": m.get_user_embedding(7),
          "user100": m.get_user_embedding(100)}
    np.savez_compressed(OUT / "g5_inference.npz", **g5)

    # ---------------- G6: a reference save() checkpoint -----------------------
    m, sd = build(ref, 20, 30, 32, 128, 61)  # hidden=128: the only size reference load() can read back
    m.precompute_item_embeddings([1, 2, 3], np.zeros((3, fx.N_GENRES), dtype=np.float32))
    m.save(str(OUT / "g6_reference_checkpoint.pt"))
    with torch.no_grad():
        np.savez_compressed(OUT / "g6_expected.npz", U=m.user_tower(t(np.array([1, 2, 20]))).numpy())
    print("golden vectors written to", OUT)
    for f in sorted(OUT.iterdir()):
        print(f"  {f.name}: {f.stat().st_size} B")


if __name__ == "__main__":
    main()
