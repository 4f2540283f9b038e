"""GPU, 2 ranks sharing one card (gloo staging of the collectives): the sharded HipBPRTrainer
reproduces the single-process step -- same loss, same MLPs, user shards equal to the corresponding rows of the
single-process table, and the item table either replicated (identical on both ranks) or row-sharded
(rank r holds global rows r+1, r+1+W, ...: ids/rows/grads all-to-alls).  A third test runs the distributed code path
through RCCL itself with a world of one rank (the only RCCL run a one-GPU box allows)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fixtures as fx

pytestmark = pytest.mark.gpu

NU, NI, D, H, B = 64, 90, 64, 128, 40   # NU users split in two shards of 32 rows (+ padding row 0 each)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _state():
    return fx.make_state(NU, NI, D, H, seed=5)


def _batches(mode):
    out = []
    for step in range(3):
        u, p, gp, n, gn = fx.make_batch(NU, NI, 2 * B, seed=50 + step, boundary=False)
        out.append((u, p, gp, n, gn))
    return out


def _run_steps(model, trainer, batches, mode, rank, world, shard):
    losses = []
    for (u, p, gp, n, gn) in batches:
        sel = np.nonzero((u - 1) // shard == rank)[0] if world > 1 else np.arange(u.size)
        # sharded runs need exactly B local pairs: batches are built so that each half owns B pairs
        ul = u[sel] - rank * shard if world > 1 else u[sel]
        if mode == "sampled":
            items = np.concatenate([p[sel], n[sel]]); g = np.concatenate([gp[sel], gn[sel]])
        else:
            items = p[sel]; g = gp[sel]
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        losses.append(float(trainer.step(t(ul), t(items), t(g)).item()))
    return losses


def _fix_batches(batches, shard):
    """re-draw user ids so that the first B pairs belong to shard 0 and the last B to shard 1"""
    out = []
    for (u, p, gp, n, gn) in batches:
        u = u.copy()
        u[:B] = (u[:B] - 1) % shard + 1
        u[B:] = (u[B:] - 1) % shard + 1 + shard
        out.append((u, p, gp, n, gn))
    return out


def _worker(rank, world, port, mode, out_dir, item_shard="replicate"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from recommendit_amd import TwoTowerModel
    from recommendit_amd.trainer import HipBPRTrainer
    shard = NU // world
    sd = _state()
    m = TwoTowerModel(shard, NI, embed_dim=D, hidden_dim=H, dropout=0.0)
    loc = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    ut = sd["user_tower.embedding.weight"]
    loc["user_tower.embedding.weight"] = torch.from_numpy(
        np.concatenate([ut[:1], ut[1 + rank * shard:1 + (rank + 1) * shard]]).copy())
    if item_shard == "rows":
        from recommendit_amd.dist_utils import shard_rows
        it_local = shard_rows(sd["item_tower.embedding.weight"], rank, world)
        m = TwoTowerModel(shard, it_local.shape[0] - 1, embed_dim=D, hidden_dim=H, dropout=0.0)
        loc["item_tower.embedding.weight"] = torch.from_numpy(it_local.copy())
    m.load_state_dict(loc)
    m.train()
    tr = HipBPRTrainer(m, B, lr=5e-3, loss_mode=mode, table_opt="sparse", distributed=True, item_shard=item_shard)
    losses = _run_steps(m, tr, _fix_batches(_batches(mode), shard), mode, rank, world, shard)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), losses=np.array(losses),
             **{k: v.detach().cpu().numpy() for k, v in m.named_parameters()})
    dist.destroy_process_group()


@pytest.mark.parametrize("item_shard", ["replicate", "rows"])
@pytest.mark.parametrize("mode", ["inbatch", "sampled"])
def test_two_rank_trainer_matches_single_process(tmp_path, mode, item_shard):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), item_shard), nprocs=world, join=True)
    from recommendit_amd import TwoTowerModel
    from recommendit_amd.trainer import HipBPRTrainer
    sd = _state()
    m = TwoTowerModel(NU, NI, embed_dim=D, hidden_dim=H, dropout=0.0)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.train()
    tr = HipBPRTrainer(m, 2 * B, lr=5e-3, loss_mode=mode, table_opt="sparse")
    ref_losses = _run_steps(m, tr, _fix_batches(_batches(mode), NU // world), mode, 0, 1, NU)
    ref = {k: v.detach().cpu().numpy() for k, v in m.named_parameters()}
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    shard = NU // world
    for r, p in enumerate(parts):
        np.testing.assert_allclose(p["losses"], ref_losses, atol=2e-6)
        for k in ref:
            if k == "user_tower.embedding.weight":
                np.testing.assert_allclose(p[k][1:], ref[k][1 + r * shard:1 + (r + 1) * shard], atol=3e-5, err_msg=k)
            elif k == "item_tower.embedding.weight" and item_shard == "rows":
                np.testing.assert_allclose(p[k][1:], ref[k][1 + r::world], atol=3e-5, err_msg=k)
            else:
                np.testing.assert_allclose(p[k], ref[k], atol=3e-5, err_msg=k)  # Adam amplifies ulp-level grad differences of near-zero grads


def _nccl_world1_worker(rank, port, out_dir):
    """FRESH process: the RCCL group is created before any other GPU work, then the distributed code path (async
    all-gather / reduce-scatter handles, all-to-alls, stream ordering against the ctypes kernels) runs with W = 1."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from recommendit_amd import TwoTowerModel
    from recommendit_amd.trainer import HipBPRTrainer
    out = {}
    for mode in ("inbatch", "sampled"):
        for tag, kw in (("plain", {}), ("dist", dict(distributed=True)), ("rows", dict(distributed=True, item_shard="rows"))):
            sd = _state()
            m = TwoTowerModel(NU, NI, embed_dim=D, hidden_dim=H, dropout=0.1)
            m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
            m.train()
            tr = HipBPRTrainer(m, 2 * B, lr=5e-3, loss_mode=mode, table_opt="sparse", seed=3, **kw)
            losses = _run_steps(m, tr, _batches(mode), mode, 0, 1, NU)
            torch.cuda.synchronize()
            out[f"{mode}_{tag}_loss"] = np.array(losses)
            for k, v in m.named_parameters():
                out[f"{mode}_{tag}_{k}"] = v.detach().cpu().numpy()
    np.savez(os.path.join(out_dir, "w1.npz"), **out)
    dist.destroy_process_group()


def test_rccl_world1_distributed_path_is_bitwise_the_plain_step(tmp_path):
    """RCCL (backend "nccl") with one rank: all_gather_into_tensor / reduce_scatter_tensor async handles, the
    all_to_all_single exchange of the row-sharded item table and their ordering against kernels launched through
    ctypes on torch's current and side streams.  With W = 1 every collective is the identity, so the distributed step
    must equal the non-distributed one bit for bit (dropout on: same counter-based masks)."""
    mp.spawn(_nccl_world1_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "w1.npz")
    for mode in ("inbatch", "sampled"):
        keys = [k[len(mode) + 7:] for k in r.files if k.startswith(f"{mode}_plain_")]
        assert len(keys) == 11
        for tag in ("dist", "rows"):
            for k in keys:
                np.testing.assert_array_equal(r[f"{mode}_{tag}_{k}"], r[f"{mode}_plain_{k}"], err_msg=f"{mode} {tag} {k}")


def test_route_rows_and_gather_kernels_vs_numpy():
    """csrc/shard.hip against the NumPy restatement used by tests/test_dist_cpu.py (stable sort by owner rank)"""
    from recommendit_amd import _lib as L
    lib, dev, st = L.lib(), L.device(), L.stream_ptr()
    rng = np.random.RandomState(3)
    for B, W, n in ((40, 2, 90), (8192, 8, 10_000_000), (1000, 5, 7), (16384, 1, 1000)):
        ids = rng.zipf(1.2, B).clip(1, n).astype(np.int64)
        i64 = dict(dtype=torch.int64, device=dev)
        idd = torch.from_numpy(ids).to(dev)
        loc, perm, pos, cnt = (torch.empty(B, **i64), torch.empty(B, **i64), torch.empty(B, **i64), torch.empty(W, **i64))
        ws = torch.empty(lib.rihip_route_workspace_bytes(B), dtype=torch.uint8, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        L.check(lib.rihip_route_rows(idd.data_ptr(), B, W, loc.data_ptr(), perm.data_ptr(), pos.data_ptr(), cnt.data_ptr(),
                                     err.data_ptr(), ws.data_ptr(), ws.numel(), st), "route_rows")
        owner = (ids - 1) % W
        p = np.argsort(owner, kind="stable")
        np.testing.assert_array_equal(perm.cpu().numpy(), p)
        np.testing.assert_array_equal(loc.cpu().numpy(), (ids[p] - 1) // W + 1)
        np.testing.assert_array_equal(pos.cpu().numpy()[p], np.arange(B))
        np.testing.assert_array_equal(cnt.cpu().numpy(), np.bincount(owner, minlength=W))
        assert err.item() == 0
        tab = torch.randn(257, 64, device=dev)
        gi = torch.from_numpy(rng.randint(0, 257, B)).to(dev)
        out = torch.empty(B, 64, device=dev)
        L.check(lib.rihip_gather_rows(tab.data_ptr(), 257, gi.data_ptr(), B, 64, out.data_ptr(), err.data_ptr(), st), "gather")
        assert torch.equal(out, tab[gi]) and err.item() == 0


def test_route_rows_fixed_and_scatter_kernels_vs_numpy():
    """the fixed-capacity routing of the host-sync-free exchange against tests/test_dist_cpu.py::_route_fixed_np: slots,
    padding id 0 -> rank 0's row 0, unused slots 0, counts on the device, overflow -> error bit 2, scatter into slots"""
    from recommendit_amd import _lib as L
    from test_dist_cpu import _route_fixed_np      # (tests/ is on sys.path: rootdir conftest, no package)
    lib, dev, st = L.lib(), L.device(), L.stream_ptr()
    rng = np.random.RandomState(4)
    for B, W, n, cap in ((40, 2, 90, 40), (8192, 8, 10_000_000, 8192), (1000, 5, 7, 1000), (256, 256, 5000, 256),
                         (4096, 8, 100000, 1024)):
        ids = rng.zipf(1.2, B).clip(1, n).astype(np.int64)
        ids[rng.randint(0, B, 3)] = 0
        if cap < B:     # a capacity with slack: uniform ids stay below it
            ids = rng.randint(1, n + 1, B).astype(np.int64)
        i64 = dict(dtype=torch.int64, device=dev)
        idd = torch.from_numpy(ids).to(dev)
        slot_ids, slot, cnt = torch.full((W * cap,), -5, **i64), torch.empty(B, **i64), torch.empty(W, **i64)
        ws = torch.empty(lib.rihip_route_workspace_bytes(B), dtype=torch.uint8, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        L.check(lib.rihip_route_rows_fixed(idd.data_ptr(), B, W, cap, slot_ids.data_ptr(), slot.data_ptr(), cnt.data_ptr(),
                                           err.data_ptr(), ws.data_ptr(), ws.numel(), st), "route_rows_fixed")
        e_ids, e_slot, e_cnt = _route_fixed_np(ids, W, cap)
        np.testing.assert_array_equal(slot_ids.cpu().numpy(), e_ids)
        np.testing.assert_array_equal(slot.cpu().numpy(), e_slot)
        np.testing.assert_array_equal(cnt.cpu().numpy(), e_cnt)
        assert err.item() == 0
        src = torch.randn(B, 64, device=dev)
        out = torch.full((W * cap, 64), 3.0, device=dev)
        L.check(lib.rihip_scatter_rows(src.data_ptr(), slot.data_ptr(), B, W * cap, 64, out.data_ptr(), err.data_ptr(), st),
                "scatter_rows")
        exp = torch.full((W * cap, 64), 3.0, device=dev)
        exp[slot] = src
        assert torch.equal(out, exp) and err.item() == 0
    # overflow: every id owned by rank 1, capacity 16
    B, W, cap = 64, 4, 16
    idd = torch.full((B,), 2, dtype=torch.int64, device=dev)
    slot_ids, slot, cnt = torch.empty(W * cap, dtype=torch.int64, device=dev), torch.empty(B, dtype=torch.int64, device=dev), \
        torch.empty(W, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.rihip_route_workspace_bytes(B), dtype=torch.uint8, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(lib.rihip_route_rows_fixed(idd.data_ptr(), B, W, cap, slot_ids.data_ptr(), slot.data_ptr(), cnt.data_ptr(),
                                       err.data_ptr(), ws.data_ptr(), ws.numel(), st), "route_rows_fixed")
    assert err.item() & 2 and cnt.cpu().tolist() == [0, 16, 0, 0]
    assert int(slot.max()) < W * cap and int(slot.min()) >= cap          # aliased, never out of range


def test_trainer_check_errors_raises_on_bad_ids():
    from recommendit_amd import TwoTowerModel
    from recommendit_amd.trainer import HipBPRTrainer
    m = TwoTowerModel(50, 60, embed_dim=64, hidden_dim=128, dropout=0.0)
    m.train()
    tr = HipBPRTrainer(m, 32, loss_mode="sampled", table_opt="sparse")
    g = torch.zeros(64, 18, device="cuda")
    tr.step(torch.randint(1, 51, (32,), device="cuda"), torch.randint(1, 61, (64,), device="cuda"), g)
    tr.check_errors()                                                   # clean step: nothing raised
    bad = torch.randint(1, 61, (64,), device="cuda"); bad[5] = 10_000
    tr.step(torch.randint(1, 51, (32,), device="cuda"), bad, g)
    with pytest.raises(RuntimeError, match="outside its embedding table"):
        tr.check_errors()
    tr.check_errors()                                                   # the word was cleared
