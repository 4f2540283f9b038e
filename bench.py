#!/usr/bin/env python3
"""bench.py -- BPR pairs/sec (+ top-500 IP queries/sec, end-to-end serve) of the MI355X hot path.

    python bench.py --gpus N --steps K --warmup W           (any N: for N > 1 without a launcher this process starts the
                                                             N ranks itself as fresh children, BEFORE any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (RANK/WORLD_SIZE from env)

Headline (BASELINE.json north_star): one step = one pass of the Two-Tower BPR training path (towers fwd ->
in-batch-negative BPR -> towers bwd -> global clip-norm -> Adam on MLPs + row-sparse Adam on touched embedding rows)
over a GLOBAL batch of 65 536 synthetic pairs with global in-batch negatives.  STRONG scaling: tables and the global
batch are fixed, N ranks shard the user rows and the batch (8 192 pairs per rank at N=8), in-batch negatives stay global
through an RCCL all-gather of the item-tower outputs; per-pair work (6 x 65 536 x 128 FLOP of score matrix) is the same
at every N.

  --config cfg3 (default for N < 8): BASELINE configs[2] tables, 10M users x 1M items; item table replicated.
  --config cfg4 (default for N = 8): BASELINE configs[3] tables, 100M users x 10M items, BOTH tables row-sharded (item
            rows travel by ids/rows/grads all-to-alls), B_local = 8 192.  Fits one GPU too (169 GB of tables+moments).

Inputs (ids, genres, tables) are resident in HBM before the timed region.  Embedding rows are drawn U(-2, 2)
(trained-scale rows; Xavier over 10M rows gives +-0.0008, every tower output collapses onto the bias direction and
the loss sits at ln 2 -- degenerate operands for a power-limited chip).

Extra objects on the JSON line: `roofline` (dominant kernel = in-batch user pass, exact-f32 MFMA bound), `cpu_baseline`
(stock torch-CPU f32 restatement of the same step on the host cores, bounded sample), `secondary` (each leg with its
own `roofline` and `cpu_baseline`): split-bf16 modes of the same step, the sampled-negative step (the mode the
reference trains in), the ML-1M-shaped configs[0]/[1] steps, top-500 brute-force IP retrieval on two query
distributions, and the cfg5 serve chain.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_BYTES = 8.0e12
D, H = 128, 128
GLOBAL_BATCH = 65536
K_TOP = 500
CONFIGS = {
    "cfg3": dict(users=10_000_000, items=1_000_000, item_shard="replicate",
                 name="BASELINE configs[2] tables: synthetic 10M users x 1M items"),
    "cfg4": dict(users=100_000_000, items=10_000_000, item_shard="rows",
                 name="BASELINE configs[3] tables: synthetic 100M users x 10M items"),
}
INIT_HALF_WIDTH = 2.0


def config_for(world, choice="auto"):
    return choice if choice != "auto" else ("cfg4" if world == 8 else "cfg3")


def workload_name(world, choice="auto"):
    return CONFIGS[config_for(world, choice)]["name"]


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def make_model(n_users_local, n_items_local, d, hidden, seed, user_seed, item_seed):
    """MLPs from `seed`; tables re-drawn U(-w, w) (rank-specific seeds for shards) so that tower outputs are not
    degenerate.  Replicas are made identical across ranks by an explicit broadcast in build_trainer."""
    from recommendit_amd import TwoTowerModel
    torch.manual_seed(seed)
    m = TwoTowerModel(n_users_local, n_items_local, embed_dim=d, hidden_dim=hidden, dropout=0.1)
    dev = m.user_tower.embedding.weight.device
    g = torch.Generator(device=dev)
    g.manual_seed(item_seed)
    m.item_tower.embedding.weight.data.uniform_(-INIT_HALF_WIDTH, INIT_HALF_WIDTH, generator=g)
    g.manual_seed(user_seed)
    m.user_tower.embedding.weight.data.uniform_(-INIT_HALF_WIDTH, INIT_HALF_WIDTH, generator=g)
    m.train()
    return m


def zipf_ids(n, n_items, a, gen, device):
    """item ids ~ Zipf(a) over [1, n_items] by inverse-CDF of the continuous approximation (popularity skew
    stresses the row-gradient grouping)."""
    u = torch.rand((n,), device=device, generator=gen, dtype=torch.float64)
    x = (1.0 + u * (float(n_items) ** (1.0 - a) - 1.0)) ** (1.0 / (1.0 - a))
    return x.floor().clamp_(1, n_items).to(torch.int64)


def make_batches(n, B, n_users_local, n_items, device, seed, sampled):
    n = min(n, 64)   # distinct synthetic batches kept in HBM (callers index modulo len): bounds memory for large --steps
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out = []
    for _ in range(n):
        u = torch.randint(1, n_users_local + 1, (B,), device=device, generator=gen)
        items = zipf_ids(B * (2 if sampled else 1), n_items, 1.05, gen, device)
        genres = (torch.rand((items.numel(), 18), device=device, generator=gen) < 0.1).float()
        out.append((u, items, genres))
    return out


def timed(fn, n, world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def timed_blocks(step, steps_per_block, world=1, blocks=5, warm_s=0.2):
    """Short legs (a few ms of GPU work per timed window) are exposed to one host stall or a clock ramp after a CPU-only
    phase: warm up BY TIME with the leg's own work (>= warm_s), then time `blocks` windows of `steps_per_block` steps
    and report the median window (value) next to the best one.  Returns (median_s_per_step, min_s_per_step, all)."""
    i, t_end = 0, time.perf_counter() + warm_s
    while True:
        step(i); i += 1
        if i % 4 == 0:
            torch.cuda.synchronize()
            stop = time.perf_counter() > t_end
            if world > 1:     # every rank must leave the warm-up after the same number of (collective) steps
                f = torch.tensor([1.0 if stop else 0.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(f, op=dist.ReduceOp.MAX)
                stop = bool(f.item() > 0)
            if stop:
                break
    per = []
    for _ in range(blocks):
        base = i
        per.append(timed(lambda j: step(base + j), steps_per_block, world) / steps_per_block)
        i += steps_per_block
    return statistics.median(per), min(per), per


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh children of THIS process, which has made
    no GPU call (device_count() does not initialise HIP on this image) and never execs.  Rank 0's stdout is ours (the
    one JSON line); a failing rank takes the others down by PID."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("MASTER_PORT", str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + list(argv), env=e,
                                      stdout=None if r == 0 else sys.stderr))
    rc, alive = 0, set(range(n))
    while alive:
        for r in sorted(alive):
            c = procs[r].poll()
            if c is None:
                continue
            alive.discard(r)
            if c != 0 and rc == 0:
                rc = c
                print(f"[bench] rank {r} exited with {c}: stopping the other ranks", file=sys.stderr, flush=True)
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def git_head():
    try:
        return subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=5).stdout.strip() or None
    except Exception:
        return None


def load_traffic():
    """HBM bytes per launch from the rocprofv3 --pmc passes of tools/profile_round.sh (FETCH_SIZE doubled as the guide
    prescribes for wide coalesced reads + WRITE_SIZE).  A static table measured on an earlier commit, NOT in this run:
    the source commit travels with the number."""
    tf = ROOT / "profiles" / "traffic.json"
    if tf.exists():
        try:
            return json.loads(tf.read_text())
        except Exception:
            pass
    return {}


def event_means(ev):
    by = {}
    for what, e0, e1 in ev:
        by.setdefault(what, []).append(e0.elapsed_time(e1))
    return {k: sum(v) / len(v) for k, v in by.items()}


def run_inbatch(tr, batches, W, K, world, precision):
    tr.inbatch_precision = precision

    def step(i):
        u, it, g = batches[i % len(batches)]
        tr.step(u, it, g)

    for i in range(W):
        step(i)
    ev = []
    tr.sweep_events = ev          # per-launch HIP events on the launch stream, inside the timed region
    dt = timed(lambda i: step(W + i), K, world)
    tr.sweep_events = None
    return dt, event_means(ev), float(tr.loss.item())


def build_trainer(cfg, world, rank, B, loss_mode, seed):
    from recommendit_amd.dist_utils import broadcast_, n_local_rows
    from recommendit_amd.trainer import HipBPRTrainer
    n_users_local = cfg["users"] // world
    rows = cfg["item_shard"] == "rows" and world > 1
    n_items_local = n_local_rows(cfg["items"], rank, world) if rows else cfg["items"]
    model = make_model(n_users_local, n_items_local, D, H, seed=1234, user_seed=1000 + rank,
                       item_seed=2000 + (rank if rows else 0))
    tr = HipBPRTrainer(model, B, lr=1e-3, weight_decay=1e-5, loss_mode=loss_mode, table_opt="sparse", seed=seed,
                       distributed=world > 1, item_shard="rows" if rows else "replicate")
    if world > 1:   # replicated state starts bit-identical on every rank: MLPs always, the item table when replicated
        broadcast_(tr.flat_p, 0)
        if not rows:
            broadcast_(tr.itab, 0)
    return model, tr, n_users_local


def headline(cfg_name, args, world, rank, dev, want_modes):
    cfg = CONFIGS[cfg_name]
    G = args.global_batch
    B = G // world
    K, W = args.steps, args.warmup
    model, tr, n_users_local = build_trainer(cfg, world, rank, B, "inbatch", seed=rank)
    batches = make_batches(W + K, B, n_users_local, cfg["items"], dev, seed=7 + rank, sampled=False)
    dt, mean_ms, loss = run_inbatch(tr, batches, W, K, world, 0)
    pairs_per_s = G * K / dt
    traffic = load_traffic()
    bgd = float(B) * G * D
    if "inbatch_user_pass" in mean_ms:
        # stored-G form: the user pass computes the scores once (2BGd) and dU (2BGd) and writes G; the item pass is
        # dI = G^T.U (2BGd).  Executed = algorithmic = 6*B_neg*d per pair.
        t_launch, t_item = mean_ms["inbatch_user_pass"] / 1e3, mean_ms["inbatch_item_pass"] / 1e3
        flop_launch = 4.0 * bgd
        achieved = flop_launch / t_launch / 1e12
        roofline = {"bound": "mfma", "kernel": "inbatch_sweep_kernel<128,user,store-G>", "achieved": achieved,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                    "traffic": traffic.get(f"inbatch_user_pass_n{world}"),
                    "traffic_source": traffic.get("_source", "profiles/traffic.json (static PMC table, not this run)"),
                    "launch_ms": t_launch * 1e3, "algorithmic_flop_per_launch": flop_launch,
                    "executed_flop_per_launch": flop_launch,
                    "second_kernel": {"kernel": "inbatch_gt_kernel<128>", "launch_ms": t_item * 1e3,
                                      "algorithmic_flop_per_launch": 2.0 * bgd, "achieved": 2.0 * bgd / t_item / 1e12,
                                      "frac": 2.0 * bgd / t_item / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                      "traffic": traffic.get(f"inbatch_item_pass_n{world}")},
                    "loss_stage_algorithmic_tflops": 6.0 * bgd / (t_launch + t_item) / 1e12}
        log(f"[bench] {cfg_name} in-batch: {pairs_per_s:,.0f} pairs/s, {dt / K * 1e3:.2f} ms/step, loss {loss:.4f}, "
            f"user pass {t_launch * 1e3:.3f} ms = {achieved:.1f} TFLOP/s, item pass {t_item * 1e3:.3f} ms = "
            f"{2.0 * bgd / t_item / 1e12:.1f} TFLOP/s")
    else:
        t_launch = mean_ms.get("inbatch_sweep", 0.0) / 1e3
        flop_launch = 3.0 * bgd                       # algorithmic: 6*B_neg*d per pair / 2 launches
        achieved = flop_launch / t_launch / 1e12 if t_launch > 0 else 0.0
        roofline = {"bound": "mfma", "kernel": "inbatch_sweep_kernel<128>", "achieved": achieved,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                    "traffic": traffic.get(f"inbatch_sweep_n{world}"), "launch_ms": t_launch * 1e3,
                    "algorithmic_flop_per_launch": flop_launch, "executed_flop_per_launch": 4.0 * bgd}
        log(f"[bench] {cfg_name} in-batch: {pairs_per_s:,.0f} pairs/s, {dt / K * 1e3:.2f} ms/step, loss {loss:.4f}, "
            f"sweep {t_launch * 1e3:.3f} ms/launch = {achieved:.1f} TFLOP/s algorithmic")
    modes = {}
    if want_modes:
        # same step, stored-G passes on split-bf16 MFMA with fp32-level accuracy ("bf16x6": exact 3-way operand split, 6
        # partial products; same test tolerances) and the two-piece "bf16x3" sweep
        dt6, m6, l6 = run_inbatch(tr, batches, W, K, world, 2)
        tu6, ti6 = m6.get("inbatch_user_pass", 0.0) / 1e3, m6.get("inbatch_item_pass", 0.0) / 1e3
        modes["inbatch_bf16x6"] = {
            "metric": "bpr_pairs_per_sec", "value": G * K / dt6, "unit": "pairs/s", "ms_per_step": dt6 / K * 1e3,
            "dtype": "bf16x6: fp32 operands split exactly into 3 bf16 pieces, 6 of 9 partial products on bf16 MFMA, "
                     "f32 accumulate (dropped terms <= 2^-23 |a||b|)",
            "user_pass_ms": tu6 * 1e3, "item_pass_ms": ti6 * 1e3, "final_loss": l6,
            "roofline": {"bound": "mfma", "kernel": "inbatch_x6_user_kernel", "unit": "TFLOP/s",
                         "achieved": 24.0 * bgd / tu6 / 1e12 if tu6 > 0 else 0.0, "peak": PEAK_BF16_MFMA_TFLOPS,
                         "frac": 24.0 * bgd / tu6 / 1e12 / PEAK_BF16_MFMA_TFLOPS if tu6 > 0 else 0.0,
                         "note": "executed bf16 MFMA FLOP (6 partial products); algorithmic = 1/6 of it"},
            "note": "optional precision mode (HipBPRTrainer(inbatch_precision=2)); held to the SAME tolerances as the "
                    "f32-MFMA path in tests/test_gpu_towers.py; the chip runs it power-limited at ~1.8 GHz"}
        log(f"[bench] in-batch bf16x6: {G * K / dt6:,.0f} pairs/s, {dt6 / K * 1e3:.2f} ms/step, user pass "
            f"{tu6 * 1e3:.3f} ms, item pass {ti6 * 1e3:.3f} ms")
        dtb, mb, lb = run_inbatch(tr, batches, W, K, world, 1)
        tl2 = mb.get("inbatch_sweep", 0.0) / 1e3
        modes["inbatch_bf16x3"] = {
            "metric": "bpr_pairs_per_sec", "value": G * K / dtb, "unit": "pairs/s", "ms_per_step": dtb / K * 1e3,
            "dtype": "bf16x3 split products (hi.hi+hi.lo+lo.hi), f32 accumulate", "sweep_launch_ms": tl2 * 1e3,
            "final_loss": lb,
            "roofline": {"bound": "mfma", "kernel": "inbatch_bf16_sweep_kernel", "unit": "TFLOP/s",
                         "achieved": 12.0 * bgd / tl2 / 1e12 if tl2 > 0 else 0.0, "peak": PEAK_BF16_MFMA_TFLOPS,
                         "frac": 12.0 * bgd / tl2 / 1e12 / PEAK_BF16_MFMA_TFLOPS if tl2 > 0 else 0.0,
                         "note": "executed bf16 MFMA FLOP per sweep launch (3 partial products x 4BGd)"},
            "note": "optional precision mode of the dominant kernel; relative product error ~2^-16; same tests, "
                    "looser tolerance (tests/test_gpu_towers.py::test_inbatch_bf16x3_precision_mode)"}
        log(f"[bench] in-batch bf16x3: {G * K / dtb:,.0f} pairs/s, {dtb / K * 1e3:.2f} ms/step, sweep {tl2 * 1e3:.3f} ms/launch")
    rows = cfg["item_shard"] == "rows" and world > 1
    per_rank_gb = (cfg["users"] // world + (cfg["items"] // world if rows else cfg["items"])) * D * 4 * 3 / 1e9
    config = {"workload": f"{cfg['name']}, d={D}, hidden={H}: Two-Tower BPR step, global batch {G} with global "
                          f"in-batch negatives, row-sparse Adam, random-init MLPs, embedding rows U(-{INIT_HALF_WIDTH},"
                          f"{INIT_HALF_WIDTH}); per-rank tables+moments {per_rank_gb:.1f} GB",
              "config": cfg_name, "global_batch": G, "per_gpu_batch": B, "embed_dim": D, "loss_mode": "inbatch",
              "parallelism": f"user rows sharded x{world}, item table "
                             + (f"row-sharded x{world} (ids/rows/grads all-to-all)" if rows else
                                ("replicated" if world > 1 else "on the one GPU"))}
    res = dict(value=pairs_per_s, ms_per_step=dt / K * 1e3, loss=loss, roofline=roofline, config=config, modes=modes)
    del tr, model, batches
    torch.cuda.empty_cache()
    return res


def leg_sampled(cfg, args, world, rank, dev, cpu):
    K, W = args.steps, args.warmup
    Bs = 65536 // world
    model, tr, n_users_local = build_trainer(cfg, world, rank, Bs, "sampled", seed=rank)
    batches = make_batches(W + K, Bs, n_users_local, cfg["items"], dev, seed=9 + rank, sampled=True)
    med, best, _ = timed_blocks(lambda i: tr.step(*batches[i % len(batches)]), max(K, 10), world)
    dts = med * K
    sp = Bs * world / med
    # SURVEY §8d per pair at d=128: 617 472 FLOP (fwd+bwd of three tower passes), 9 384 B of row traffic
    f_mfma, f_hbm = sp * 617472 / (PEAK_F32_MFMA_TFLOPS * 1e12 * world), sp * 9384 / (PEAK_HBM_BYTES * world)
    out = {"metric": "bpr_pairs_per_sec_sampled_negative", "value": sp, "unit": "pairs/s", "ms_per_step": med * 1e3,
           "ms_per_step_min": best * 1e3, "timing": "median of 5 windows after >= 0.2 s of warm-up steps",
           "global_batch": Bs * world, "final_loss": float(tr.loss.item()),
           "roofline": {"bound": "mfma", "kernel": "tower_fwd2 / tower_bwd_data / tower_wgrad (exact f32)",
                        "achieved": sp * 617472 / 1e12 / world, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": f_mfma, "frac_of_hbm_roofline": f_hbm,
                        "note": "whole step (towers + loss + row-sparse optimiser) priced at the towers' algorithmic FLOP"}}
    if cpu:
        from oracle import torch_cpu_baseline as T
        out["cpu_baseline"] = T.time_sampled_step(B=4096, n_users=65536, n_items=65536, d=D, hidden=H, budget_s=4.0)
        out["cpu_baseline"]["note"] = ("dense Adam on 65 536-row tables stands in for the 10M-row ones (dense Adam over "
                                       "10M rows would add 3 x 5 GB of traffic per step on the CPU)")
    log(f"[bench] sampled: {sp:,.0f} pairs/s, {dts / K * 1e3:.2f} ms/step")
    del tr, batches
    return out, model


def leg_ml1m(dev, cpu):
    """BASELINE.json configs[0]/[1] shapes: 6041 x 64 + 3953 x 64 tables, dense Adam + L2 (the reference's optimiser)."""
    from recommendit_amd.trainer import HipBPRTrainer
    from recommendit_amd import TwoTowerModel
    out = {}
    for tag, (bb, mode) in {"cfg1_ml1m_d64_b256_sampled": (256, "sampled"),
                            "cfg2_ml1m_d64_b8192_inbatch": (8192, "inbatch")}.items():
        torch.manual_seed(5)
        m2 = TwoTowerModel(6040, 3952, embed_dim=64, hidden_dim=128, dropout=0.1)
        m2.train()
        b2 = make_batches(8, bb, 6040, 3952, dev, seed=11, sampled=(mode == "sampled"))
        flop_pair = 322560.0 + (6.0 * bb * 64 if mode == "inbatch" else 0.0)   # SURVEY §8d, d=64
        n2 = 400 if bb <= 1024 else 100
        res = {}
        for how in ("eager", "hipgraph"):
            t2 = HipBPRTrainer(m2, bb, loss_mode=mode, table_opt="dense", seed=1, use_graph=(how == "hipgraph"))
            med, best, per = timed_blocks(lambda i: t2.step(*b2[i % 8]), n2)
            res[how] = {"ms_per_step": med * 1e3, "ms_per_step_min": best * 1e3, "pairs_per_s": bb / med,
                        "windows_ms": [x * 1e3 for x in per]}
            del t2
        if mode == "sampled":   # the same step as ONE persistent launch (three grid barriers): reported beside, opt-in
            try:
                t2 = HipBPRTrainer(m2, bb, loss_mode=mode, table_opt="dense", seed=1, persistent=True)
                med, best, per = timed_blocks(lambda i: t2.step(*b2[i % 8]), n2)
                t2.check_errors()
                res["persistent"] = {"ms_per_step": med * 1e3, "ms_per_step_min": best * 1e3, "pairs_per_s": bb / med,
                                     "launches_per_step": 1, "grid_barriers": 3}
                del t2
            except Exception as e:
                res["persistent"] = {"error": repr(e)}
        best_how = min(("eager", "hipgraph"), key=lambda h: res[h]["ms_per_step"])
        med = res[best_how]["ms_per_step"] / 1e3
        launches = 7 if mode == "sampled" else 13
        out[tag] = {"metric": "bpr_pairs_per_sec", "value": bb / med, "unit": "pairs/s",
                    "ms_per_step": med * 1e3, "ms_per_step_min": res[best_how]["ms_per_step_min"], "submission": best_how,
                    "eager": res["eager"], "hipgraph": res["hipgraph"], "launches_per_step": launches,
                    "persistent_one_launch": res.get("persistent"),
                    "timing": f"median of 5 windows of {n2} steps after >= 0.2 s of warm-up steps",
                    "batch": bb, "loss_mode": mode,
                    "tables": "6041x64 + 3953x64 (MovieLens-1M shape), dense Adam+L2 (exact reference optimiser)",
                    "roofline": {"bound": "mfma", "achieved": bb / med * flop_pair / 1e12,
                                 "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": bb / med * flop_pair / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                 "note": "launch-bound regime: %d dependent launches per step (user + item tower share one launch each way, slab reduction rides in the scatter launch)" % launches}}
        log(f"[bench] {tag}: {bb / med:,.0f} pairs/s, {med * 1e3:.3f} ms/step ({best_how}; eager "
            f"{res['eager']['ms_per_step']:.3f}, hipGraph {res['hipgraph']['ms_per_step']:.3f}, best window "
            f"{res[best_how]['ms_per_step_min']:.3f})")
        del m2, b2
    if cpu:
        from oracle import torch_cpu_baseline as T
        c1 = T.time_sampled_step(B=256, budget_s=4.0)
        smp = T.time_reference_sampler(budget_s=2.0)
        c1["sampler_fed"] = {"value": min(c1["value"], 2 * smp["value"]), "unit": "pairs/s",
                             "note": "reference DataLoader with 2 workers (train_embeddings.py:144-151): min(model-bound, "
                                     "2 x per-worker sampler rate)", "sampler": smp}
        out["cfg1_ml1m_d64_b256_sampled"]["cpu_baseline"] = c1
        out["cfg2_ml1m_d64_b8192_inbatch"]["cpu_baseline"] = T.time_inbatch_block(G=8192, blk=8192, d=64, hidden=128,
                                                                                   budget_s=3.0)
    return out


def leg_retrieval(model, cfg, n_users_local, args, world, rank, dev, cpu):
    from recommendit_amd import FAISSIndex
    K = args.steps
    N = cfg["items"] if cfg["items"] <= 1_000_000 else 1_000_000     # BASELINE configs[2]: 1M-item corpus
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.randn((N, D), device=dev, generator=g)
    X = (X / X.norm(dim=1, keepdim=True)).contiguous()
    idx = FAISSIndex(embed_dim=D, exact=True)
    idx.build_from_device(X, np.arange(1, N + 1))
    nq = 4096
    g3 = torch.Generator(device=dev); g3.manual_seed(3 + rank)
    qs = []
    for i in range(2):   # pure-retrieval queries: L2-normalised N(0,1) (seed 3)
        qq = torch.randn((nq, D), device=dev, generator=g3)
        qs.append((qq / qq.norm(dim=1, keepdim=True)).contiguous())
    Kq = max(4, K)
    medq, bestq, _ = timed_blocks(lambda i: idx.batch_search_device(qs[i % 2], k=K_TOP, normalized=True), Kq, world)
    dtq = medq * Kq
    qps = nq * world / medq
    flop_q = 2.0 * N * D

    def roof(q):
        return {"bound": "mfma", "dtype": "bf16 filter pass", "kernel": "scan_bf16_kernel<128>",
                "achieved": q * flop_q / 1e12 / world, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": q * flop_q / 1e12 / world / PEAK_BF16_MFMA_TFLOPS,
                "vs_f32_mfma_peak": q * flop_q / 1e12 / world / PEAK_F32_MFMA_TFLOPS}

    out = {"metric": "top500_ip_queries_per_sec", "value": qps, "unit": "queries/s", "ms_per_batch": medq * 1e3,
           "ms_per_batch_min": bestq * 1e3, "timing": f"median of 5 windows of {Kq} batches after >= 0.2 s of warm-up",
           "queries_per_batch": nq * world, "k": K_TOP, "queries": "L2-normalised N(0,1)",
           "corpus": f"{N}x{D} f32 L2-normalised N(0,1), exact brute force (bf16-MFMA filter with proven completeness "
                     f"+ exact f32 re-score; results identical to all-f32)", "roofline": roof(qps)}
    log(f"[bench] retrieval: {qps:,.0f} q/s ({dtq / Kq * 1e3:.2f} ms per {nq} queries)")
    # second distribution: user-tower outputs against an item-tower-output corpus (clustered scores)
    try:
        model.eval()
        with torch.no_grad():
            ids = torch.arange(1, N + 1, device=dev)
            gen = torch.Generator(device=dev); gen.manual_seed(17)
            chunks = []
            for s in range(0, N, 131072):
                ii = ids[s:s + 131072]
                gg = (torch.rand((ii.numel(), 18), device=dev, generator=gen) < 0.1).float()
                chunks.append(model.item_tower(ii, gg))
            Xt = torch.cat(chunks, 0).contiguous()
            idx2 = FAISSIndex(embed_dim=D, exact=True)
            idx2.build_from_device(Xt, np.arange(1, N + 1))
            qt = [model.user_tower(torch.randint(1, n_users_local + 1, (nq,), device=dev, generator=g3)).contiguous()
                  for _ in range(2)]
        med2, best2, _ = timed_blocks(lambda i: idx2.batch_search_device(qt[i % 2], k=K_TOP, normalized=True), Kq, world)
        q2 = nq * world / med2
        out["tower_outputs"] = {"value": q2, "unit": "queries/s", "ms_per_batch": med2 * 1e3,
                                "ms_per_batch_min": best2 * 1e3,
                                "queries": "user-tower outputs", "corpus": f"item-tower outputs of {N} items",
                                "roofline": roof(q2)}
        # PCIe-inclusive: the reference API hands over / returns host NumPy arrays (faiss_index.py:126-153)
        qh = qt[0].cpu().numpy()
        idx2.batch_search(qh, k=K_TOP)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            idx2.batch_search(qh, k=K_TOP)
        out["tower_outputs"]["pcie_inclusive_qps"] = 3 * nq / (time.perf_counter() - t0)
        log(f"[bench] retrieval (tower outputs): {q2:,.0f} q/s, host-array API {out['tower_outputs']['pcie_inclusive_qps']:,.0f} q/s")
        del idx2, Xt, qt
    except Exception as e:  # never take the headline down
        out["tower_outputs"] = {"error": repr(e)}
        log(f"[bench] retrieval tower-output leg failed: {e!r}")
    if cpu:
        from oracle import torch_cpu_baseline as T
        out["cpu_baseline"] = T.time_retrieval(N=N, d=D, k=K_TOP, budget_s=6.0)
    return out, idx, X


def leg_lambdamart(dev):
    """§8f-4: LambdaMART training on the GPU at the ML-1M ranker shape (the reference trains LightGBM on the CPU,
    ranker.py:52-155): 6 040 queries, ~2.4 M rows x 50 features, 63 leaves, a few trees."""
    import pandas as pd
    from recommendit_amd import LightGBMRanker
    rng = np.random.RandomState(0)
    nq, F = 6040, 50
    sizes = np.clip(rng.lognormal(5.6, 0.9, nq).astype(int), 20, 9000)
    n = int(sizes.sum())
    Xr = rng.randn(n, F).astype(np.float32)
    y = ((Xr @ rng.randn(F) + 2.0 * rng.randn(n)) > 5.0).astype(np.float32)
    cols = [f"f{i}" for i in range(F)]
    df = pd.DataFrame(Xr, columns=cols)
    df["label"] = y
    df["query_id"] = np.repeat(np.arange(nq), sizes)
    times, res = {}, None
    for nt in (5, 25):       # two runs: the difference isolates the per-tree time from binning + upload
        rk = LightGBMRanker(num_leaves=63, n_estimators=nt, learning_rate=0.05)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = rk.train(df, cols, backend="hip")
        times[nt] = time.perf_counter() - t0
    per_tree = (times[25] - times[5]) / 20.0
    out = {"metric": "lambdamart_trees_per_sec", "value": 1.0 / per_tree, "unit": "trees/s", "ms_per_tree": per_tree * 1e3,
           "setup_s": times[5] - 5 * per_tree, "rows": n, "features": F, "queries": nq, "num_leaves": 63,
           "train_ndcg10_first_last": [res["train"]["ndcg@10"][0], res["train"]["ndcg@10"][-1]],
           "note": "rihip_lambdamart_train; per-tree time = (25-tree run - 5-tree run) / 20, setup = binning + upload; "
                   "trees bit-identical to oracle/lambdamart_np (tests/test_gpu_lambdamart.py); lightgbm itself is not "
                   "installed (parity unpinned)"}
    log(f"[bench] LambdaMART training: {out['ms_per_tree']:.1f} ms/tree on {n} rows x {F} features (+ {out['setup_s']:.2f} s set-up)")
    return out


def leg_serve(model, X, n_users_local, dev, cpu):
    """cfg5: user tower -> IVF-IP (100 lists, nprobe 10, 500 candidates) -> feature assembly -> LambdaMART -> top-20"""
    import tempfile
    from recommendit_amd import synthetic as GB
    from recommendit_amd import FAISSIndex, LightGBMRanker
    from recommendit_amd.recommender import GpuFeatureStore, GpuRecommendationPipeline, feature_columns
    N = X.shape[0]
    ivf = FAISSIndex(embed_dim=D, n_lists=100, n_probe=10)
    t0 = time.perf_counter()
    ivf.build_from_device(X, np.arange(1, N + 1))
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    forest = GB.random_forest_model(500, 63, 50, seed=4, names=feature_columns())
    text = GB.write_text_model(forest)
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "f.lgbm")
        open(pth, "w").write(text)
        ranker = LightGBMRanker.load(pth)
    store = GpuFeatureStore(8, 8)          # tiny host tables; device tables built directly below
    gg = torch.Generator(device=dev); gg.manual_seed(5)
    store._dev = (torch.rand((n_users_local + 1, 24), device=dev, generator=gg, dtype=torch.float64),
                  torch.rand((N + 1, 23), device=dev, generator=gg, dtype=torch.float64))
    pipe = GpuRecommendationPipeline(model, ivf, ranker, store, top_k_candidates=K_TOP, top_k_results=20)
    nqs = int(os.environ.get("RIHIP_SERVE_BATCH", "256"))
    uids = [torch.randint(1, n_users_local + 1, (nqs,), device=dev, generator=gg) for _ in range(3)]
    meds, bests, _ = timed_blocks(lambda i: pipe.recommend_batch(uids[i % 3]), 12)
    one = uids[0][:1]
    pipe.recommend_batch(one)
    lat = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe.recommend_batch(one)
        torch.cuda.synchronize(); lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    one_list = one.tolist()
    pipe.recommend_batch(one_list, graph=True)          # capture
    latg = []
    for _ in range(50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe.recommend_batch(one_list, graph=True)
        torch.cuda.synchronize(); latg.append((time.perf_counter() - t0) * 1e3)
    latg.sort()
    rps = nqs / meds
    # ranker alone on the candidates of one batch: node visits / s (neither HBM nor MFMA bound: dependent LDS reads)
    Xf = torch.rand((nqs * K_TOP, 50), device=dev, generator=gg)
    ranker.predict_device(Xf)
    medr, _, _ = timed_blocks(lambda i: ranker.predict_device(Xf), 10)
    dtr = medr * 5
    out = {"metric": "end_to_end_recommendations_per_sec", "value": rps, "unit": "requests/s", "batch": nqs,
           "ms_per_batch": meds * 1e3, "ms_per_batch_min": bests * 1e3,
           "timing": "median of 5 windows of 12 batches after >= 0.2 s of warm-up",
           "single_request_ms_p50": lat[len(lat) // 2], "single_request_ms_max": lat[-1],
           "single_request_graph_ms_p50": latg[len(latg) // 2], "single_request_graph_ms_p99": latg[-1], "ivf_build_s": build_s,
           "pipeline": "user tower -> IVF-IP(100 lists, nprobe 10, 500 cands) -> feature assembly -> LambdaMART 500 "
                       "trees x 63 leaves x 50 features -> top-20",
           "ranker": {"candidates_per_s": nqs * K_TOP * 5 / dtr, "tree_walks_per_s": nqs * K_TOP * 5 / dtr * 500,
                      "ms_per_128k_candidates": dtr / 5 * 1e3,
                      "roofline": {"bound": "neither", "note": "forest (1 MB) is LDS-resident; chains of dependent LDS "
                                   "reads bound the walk (SURVEY §8d): HBM/MFMA fractions are not meaningful",
                                   "hbm_frac": nqs * K_TOP * 5 / dtr * 208 / PEAK_HBM_BYTES}}}
    log(f"[bench] serve: {rps:,.0f} req/s batched, {lat[len(lat) // 2]:.2f} ms p50 single ({latg[len(latg) // 2]:.3f} ms as a "
        f"hipGraph); ranker "
        f"{nqs * K_TOP * 5 / dtr / 1e6:.1f} M candidates/s")
    if cpu:
        from oracle import gbdt_np as G
        from oracle import torch_cpu_baseline as T
        out["ranker"]["cpu_baseline"] = T.time_tree_walk(G.parse_text_model(text), 50, n=8192, budget_s=4.0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=GLOBAL_BATCH)
    ap.add_argument("--config", choices=["auto", "cfg3", "cfg4"], default="auto")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, rendezvous (gloo) and all-reduce a one per rank; no GPU work, no number")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become the launcher (no GPU call has been made in this process, and none will be)
        backend = os.environ.get("RIHIP_DIST_BACKEND", "nccl")
        n_dev = torch.cuda.device_count()
        if backend == "nccl" and n_dev < args.gpus and not args.launch_check:
            print(f"[bench] --gpus {args.gpus} but only {n_dev} HIP device(s) visible: RCCL needs one device per rank "
                  f"(RIHIP_DIST_BACKEND=gloo rehearses the N>1 path with ranks sharing a card)", file=sys.stderr)
            sys.exit(2)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size wins", file=sys.stderr)
        args.gpus = world
    backend = os.environ.get("RIHIP_DIST_BACKEND", "nccl") if world > 1 else None   # "gloo": CPU-staged rehearsal
    if args.launch_check:
        # launcher / rendezvous check only (runs on a GPU-less host too): no kernel runs and no number is reported
        ranks = 1
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            ranks = int(t.item())
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "dist_backend": "gloo" if world > 1 else None,
                              "dist_ranks": ranks, "rccl_ranks": None, "value": None,
                              "config": {"workload": workload_name(world, args.config)}}), flush=True)
        return
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP device: the hot path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)   # several ranks may share a card only in the gloo rehearsal below
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # how many ranks the collective backend really joins: an all-reduce of ones through it
        t = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        dist_ranks = int(t.item())
        assert dist_ranks == dist.get_world_size() == world, (dist_ranks, dist.get_world_size(), world)
    assert args.global_batch % world == 0
    cfg_name = config_for(world, args.config)
    cpu_ok = rank == 0 and world == 1 and not args.no_cpu_baseline
    log(f"[bench] host threads: torch.get_num_threads()={torch.get_num_threads()}, os.cpu_count()={os.cpu_count()}")

    head = headline(cfg_name, args, world, rank, dev, want_modes=not args.no_secondary)
    secondary = dict(head.pop("modes"))
    if not args.no_secondary:
        if world == 8 and cfg_name != "cfg3":
            # the strong-scaling curve's own point: the N<8 runs use the cfg3 tables
            alt = headline("cfg3", args, world, rank, dev, want_modes=False)
            secondary["inbatch_cfg3_tables"] = {"metric": "bpr_pairs_per_sec", "value": alt["value"], "unit": "pairs/s",
                                                "ms_per_step": alt["ms_per_step"], "roofline": alt["roofline"],
                                                "config": alt["config"]}
        cfg = CONFIGS["cfg3"]        # secondary legs run on the cfg3 tables
        n_users_local = cfg["users"] // world
        model = idx = X = None
        if world == 1:
            # (a secondary leg must never take the headline down; with world > 1 every rank takes part in the legs'
            # barriers, so there a failure is left to surface instead of hanging the other ranks)
            def guarded(name, fn):
                try:
                    return fn()
                except Exception as e:
                    secondary[name] = {"error": repr(e)}
                    log(f"[bench] {name} leg failed: {e!r}")
                    return None
            r = guarded("sampled_bpr", lambda: leg_sampled(cfg, args, world, rank, dev, cpu_ok))
            if r is not None:
                secondary["sampled_bpr"], model = r
            r = guarded("ml1m", lambda: leg_ml1m(dev, cpu_ok))
            if r is not None:
                secondary.update(r)
            if model is not None:
                r = guarded("retrieval", lambda: leg_retrieval(model, cfg, n_users_local, args, world, rank, dev, cpu_ok))
                if r is not None:
                    secondary["retrieval"], idx, X = r
                    r = guarded("serve", lambda: leg_serve(model, X, n_users_local, dev, cpu_ok))
                    if r is not None:
                        secondary["serve"] = r
            model = idx = X = None
            torch.cuda.empty_cache()
            r = guarded("lambdamart_train", lambda: leg_lambdamart(dev))
            if r is not None:
                secondary["lambdamart_train"] = r
        else:
            sampled, model = leg_sampled(cfg, args, world, rank, dev, cpu_ok)
            secondary["sampled_bpr"] = sampled
            retr, idx, X = leg_retrieval(model, cfg, n_users_local, args, world, rank, dev, cpu_ok)
            secondary["retrieval"] = retr
        del model, idx, X
    cpu = None
    if cpu_ok:
        from oracle import torch_cpu_baseline as T
        cpu = T.time_inbatch_block(G=args.global_batch, blk=2048, d=D, hidden=H, budget_s=8.0)
        log(f"[bench] cpu (torch f32, {cpu['cores']} threads): {cpu['value']:.1f} pairs/s")

    if rank == 0:
        line = {
            "metric": "bpr_pairs_per_sec", "value": head["value"], "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": head["config"],
            "rccl_ranks": dist_ranks if (world == 1 or backend == "nccl") else None, "dist_backend": backend,
            "dist_ranks": dist_ranks,
            "final_loss": head["loss"], "git_head": git_head(), "roofline": head["roofline"], "cpu_baseline": cpu,
            "secondary": secondary,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
