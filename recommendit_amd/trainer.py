"""Fused BPR training step over the C ABI -- the body of the reference's hot loop
(src/training/train_embeddings.py:183-192: towers -> bpr_loss -> backward -> clip_grad_norm_
-> Adam.step) without autograd, without host syncs, with preallocated buffers.

Two optimiser modes for the embedding tables:
  * ``dense``  -- the reference's exact semantics: dense gradient, every row decays through the
    coupled L2 term and moves every step.  Used for the ML-1M parity configs.
  * ``sparse`` -- row-wise Adam on touched rows only (sort ids -> segment-reduce -> fused row
    update).  The only feasible mode at 10M/100M rows; deviation from the reference: untouched rows
    neither decay nor move.  Throughput configs run this.
MLP parameters are always dense (one flat buffer: one grad-norm pass + one Adam launch).

Two loss modes: ``sampled`` (one explicit negative per positive -- what the reference trains
with) and ``inbatch`` (TwoTowerModel.in_batch_bpr_loss, never called by the reference trainer).

Multi-GPU (one process per GPU, torch.distributed/RCCL; opt-in with ``distributed=True``): user rows are sharded
(each rank trains pairs whose user row it owns: zero communication for user rows).  The item table is either
  * ``item_shard="replicate"`` -- a full copy per rank, kept identical by all-gathering the per-rank (item id,
    row-gradient) lists; or
  * ``item_shard="rows"`` -- cut by rows (global row g on rank (g-1) % W, BASELINE cfg4): per step the batch's item ids
    are routed to their owners (ids all-to-all), the owners gather the rows and send them back (rows all-to-all), the
    item tower runs on the received rows, and the row gradients return to the owners (grads all-to-all), who apply the
    row-sparse Adam to the rows they own.  The three all-to-alls use EQUAL splits of `exchange_cap` slots per peer
    (default = the worst case nI: can never overflow), so no split size crosses to the host: the step has no host sync.
In-batch negatives are global through an all-gather of the item-tower outputs; MLP grads are all-reduced; the clip norm
is an all-reduced scalar.  See DESIGN.md "Multi-GPU".
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import _lib as L
from .dist_utils import all_gather_into, all_reduce_sum_, all_to_all_rows, reduce_scatter_sum
from .two_tower import TwoTowerModel

_MLP_KEYS = ["user_tower.mlp.0.weight", "user_tower.mlp.0.bias", "user_tower.mlp.3.weight", "user_tower.mlp.3.bias",
             "item_tower.mlp.0.weight", "item_tower.mlp.0.bias", "item_tower.mlp.3.weight", "item_tower.mlp.3.bias"]


class _RowsOpt:
    """Row-sparse Adam state for one table (ids may repeat; up to max_ids per step)."""

    def __init__(self, table: torch.Tensor, max_ids: int):
        lib = L.lib()
        dev = table.device
        self.table = table
        self.d = table.shape[1]
        self.m = torch.zeros_like(table)
        self.v = torch.zeros_like(table)
        self.max_ids = max_ids
        nbytes = lib.rihip_rows_workspace_bytes(max_ids, self.d)
        if nbytes < 0:
            raise RuntimeError("rihip_rows_workspace_bytes failed")
        self.ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        self.uniq = torch.empty((max_ids,), dtype=torch.int64, device=dev)
        self.Gc = torch.empty((max_ids, self.d), dtype=torch.float32, device=dev)

    def group(self, ids: torch.Tensor, st: int) -> None:
        """sort the step's ids and find the unique rows: needs only the ids, so it can run beside the towers"""
        lib = L.lib()
        B = ids.numel()
        assert B <= self.max_ids
        L.check(lib.rihip_rows_group(ids.data_ptr(), B, self.d, self.table.shape[0], self.uniq.data_ptr(), self.ws.data_ptr(), self.ws.numel(),
                                     st),
                "rows_group")
        self._B = B

    def reduce(self, dX: torch.Tensor, part_ptr: int, st: int) -> None:
        """per-unique-row gradient sums (+ their squared-norm partials) of the ids grouped by group()"""
        L.check(L.lib().rihip_rows_reduce(dX.data_ptr(), self._B, self.d, self.uniq.data_ptr(), self.ws.data_ptr(),
                                          self.Gc.data_ptr(), part_ptr, st), "rows_reduce")

    def group_reduce(self, ids: torch.Tensor, dX: torch.Tensor, part_ptr: int, st: int) -> None:
        self.group(ids, st)
        self.reduce(dX, part_ptr, st)

    def adam(self, lr, b1, b2, eps, wd, step, coef_ptr, st, hyper_ptr=None) -> None:
        L.check(L.lib().rihip_adam_rows(self.table.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                        self.uniq.data_ptr(), self.Gc.data_ptr(), self._B, self.d, self.ws.data_ptr(),
                                        lr, b1, b2, eps, wd, step, coef_ptr, hyper_ptr, st), "adam_rows")


class _DenseOpt:
    def __init__(self, table: torch.Tensor):
        self.table = table
        self.d = table.shape[1]
        self.m = torch.zeros_like(table)
        self.v = torch.zeros_like(table)
        self.grad = torch.zeros_like(table)

    def scatter(self, ids: torch.Tensor, dX: torch.Tensor, st: int, zero: bool = True) -> None:
        if zero:
            self.grad.zero_()
        L.check(L.lib().rihip_embedding_scatter_add(self.grad.data_ptr(), self.table.shape[0], ids.data_ptr(),
                                                    dX.data_ptr(), ids.numel(), self.d, st), "embedding_scatter_add")

    def sumsq(self, part_ptr: int, st: int) -> None:
        L.check(L.lib().rihip_sumsq(self.grad.data_ptr(), self.grad.numel(), part_ptr, st), "sumsq")

    def adam(self, lr, b1, b2, eps, wd, step, coef_ptr, st, hyper_ptr=None) -> None:
        L.check(L.lib().rihip_adam_dense(self.table.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(),
                                         self.v.data_ptr(), self.table.numel(), lr, b1, b2, eps, wd, step, coef_ptr,
                                         hyper_ptr, st), "adam_dense")


class HipBPRTrainer:
    def __init__(self, model: TwoTowerModel, batch_size: int, lr: float = 1e-3, weight_decay: float = 1e-5,
                 betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0, loss_mode: str = "sampled",
                 table_opt: str = "dense", seed: int = 0, process_group=None, user_row_offset: int = 0,
                 inbatch_precision: int = 0, use_graph: bool = False, inbatch_store_g=None,
                 distributed: bool = False, item_shard: str = "replicate", exchange_cap: Optional[int] = None,
                 persistent: Optional[bool] = None):
        """distributed=True (or a process_group): this trainer is one rank of a collective job -- EVERY rank of the
        group must construct it and call step() in lock-step.  Default False even when torch.distributed is
        initialised, so that a single-rank trainer inside a distributed program never issues collectives.
        item_shard="rows": model.item_tower.embedding holds only this rank's rows of the item table (see module doc).
        exchange_cap: send slots per peer of the row exchange (None = nI, the worst case).  A smaller capacity saves
        wire volume (W*cap*d*4 bytes per all-to-all) but a step that routes more than `cap` rows to one owner sets
        error bit 2 (check_errors() raises).
        persistent: run the sampled-negative step with the dense (reference) optimiser as ONE persistent launch
        (csrc/step_persistent.hip: three grid barriers instead of seven dependent launches; single GPU, B <= 2048).
        Opt-in: measured on MI355X it is SLOWER than the seven launches today (0.124 vs 0.074 ms at the ML-1M
        B = 256 shape: each of the three grid barriers needs an L2 write-back + invalidate across the 8 XCDs, and the
        runtime-shape tile code it is built from is latency-bound) -- see DESIGN.md §9.  None / False = the multi-launch
        path."""
        assert loss_mode in ("sampled", "inbatch") and table_opt in ("dense", "sparse")
        assert item_shard in ("replicate", "rows")
        self.lib = L.lib()
        self.model = model
        self.dev = L.device()
        self.B = int(batch_size)
        self.lr, self.wd, self.b1, self.b2, self.eps, self.max_norm = lr, weight_decay, betas[0], betas[1], eps, max_norm
        self.loss_mode, self.table_opt = loss_mode, table_opt
        self.step_count = 0
        self.seed = seed
        self.inbatch_precision = int(inbatch_precision)  # 0 = f32 MFMA, 1 = bf16x3, 2 = bf16x6 (fp32-level accuracy)
        self.inbatch_store_g = inbatch_store_g  # None = auto: store G (no second score sweep) when it fits in HBM
        self.sweep_events = None
        self.use_graph = bool(use_graph)
        self._graph = None
        self._dI_work = None
        self._side_stream = torch.cuda.Stream(device=self.dev)
        self._side_stream2 = torch.cuda.Stream(device=self.dev)
        self._ev_dxu, self._ev_dxi = torch.cuda.Event(), torch.cuda.Event()   # "dX complete" of the two tower backwards
        self._eager_steps = 0  # bench hook: list collecting (start, end) events around every sweep launch
        self.pg = process_group
        self.dist = bool(distributed) or process_group is not None   # collectives are issued iff this is set
        if self.dist and not dist.is_initialized():
            raise RuntimeError("HipBPRTrainer(distributed=True) needs torch.distributed.init_process_group first")
        self.world = dist.get_world_size(process_group) if self.dist else 1
        self.rank = dist.get_rank(process_group) if self.dist else 0
        self.item_rows = self.dist and item_shard == "rows"
        self.user_row_offset = int(user_row_offset)  # global id of local user-table row 0 (sharded tables)
        d, H = model.embed_dim, model.hidden_dim
        self.d, self.H = d, H
        self.p_drop = float(model.user_tower.mlp[2].p)

        # ---- flat MLP parameter / grad / moment buffers; module parameters become views
        sd = dict(model.named_parameters())
        sizes = [sd[k].numel() for k in _MLP_KEYS]
        offs, tot = [], 0
        for s in sizes:
            offs.append(tot)
            tot += (s + 3) // 4 * 4  # keep every tensor 16-byte aligned inside the flat buffer
        self.flat_p = torch.zeros((tot,), dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.pv: Dict[str, torch.Tensor] = {}
        self.gv: Dict[str, torch.Tensor] = {}
        for k, o, s in zip(_MLP_KEYS, offs, sizes):
            prm = sd[k]
            view = self.flat_p[o:o + s].view_as(prm)
            view.copy_(prm.data)
            prm.data = view
            self.pv[k] = view
            self.gv[k] = self.flat_g[o:o + s].view_as(prm)
        self.utab = model.user_tower.embedding.weight.data
        self.itab = model.item_tower.embedding.weight.data
        nI = self.B * (2 if loss_mode == "sampled" else 1)
        self.nI = nI
        # replicated item table: applies every rank's rows; row-sharded: up to every rank's requests can hit one owner
        cap = self.exchange_cap = (int(exchange_cap) if exchange_cap else nI) if self.item_rows else 0
        assert not self.item_rows or 1 <= cap <= nI
        nI_all = self.world * cap if self.item_rows else nI * self.world
        if table_opt == "sparse":
            self.uopt = _RowsOpt(self.utab, self.B)
            self.iopt = _RowsOpt(self.itab, nI_all)
        else:
            assert not self.dist, "dense table optimiser is single-GPU (parity mode)"
            self.uopt = _DenseOpt(self.utab)
            self.iopt = _DenseOpt(self.itab)
            # host arrays of device pointers for the multi-tensor launches: [MLP flat buffer, user table, item table]
            PA, NA = C.c_void_p * 3, C.c_int64 * 3
            ts = [(self.flat_p, self.flat_g, self.flat_m, self.flat_v),
                  (self.utab, self.uopt.grad, self.uopt.m, self.uopt.v),
                  (self.itab, self.iopt.grad, self.iopt.m, self.iopt.v)]
            self._mt_p = PA(*[x[0].data_ptr() for x in ts]); self._mt_g = PA(*[x[1].data_ptr() for x in ts])
            self._mt_m = PA(*[x[2].data_ptr() for x in ts]); self._mt_v = PA(*[x[3].data_ptr() for x in ts])
            self._mt_n = NA(*[x[0].numel() for x in ts])

        can_persist = (loss_mode == "sampled" and table_opt == "dense" and not self.dist
                       and bool(self.lib.rihip_bpr_step_persistent_supported(self.B, d, H)))
        if persistent and not can_persist:
            raise ValueError("persistent=True needs loss_mode='sampled', table_opt='dense', one GPU and B <= 2048")
        self.persistent = bool(persistent)
        self._pargs = None

        # ---- per-step buffers
        f32 = dict(dtype=torch.float32, device=self.dev)
        B = self.B
        self.U = torch.empty((B, d), **f32); self.hidU = torch.empty((B, H), **f32); self.denU = torch.empty((B,), **f32)
        self.I = torch.empty((nI, d), **f32); self.hidI = torch.empty((nI, H), **f32); self.denI = torch.empty((nI,), **f32)
        self.dU = torch.empty((B, d), **f32); self.dI = torch.empty((nI, d), **f32)
        self.dXu = torch.empty((B, d), **f32); self.dXi = torch.empty((nI, d), **f32)
        self.loss = torch.zeros((), **f32)
        # device-resident step clock (graph replay must not bake host constants): step counter, lr, {lr/bc1, sqrt(bc2)}
        # (the clock is advanced by the clip-coefficient launch of each step: *step_dev = the step that is running)
        self.step_dev = torch.ones((1,), dtype=torch.int64, device=self.dev)
        self.lr_dev = torch.full((1,), float(lr), **f32)
        self.hyper_dev = torch.zeros((2,), **f32)
        self._lr_host = float(lr)
        self.coef = torch.ones((1,), **f32); self.gnorm = torch.zeros((1,), **f32)
        self.err = torch.zeros((1,), dtype=torch.int32, device=self.dev)
        self.bws_u = torch.empty((self.lib.rihip_tower_backward_workspace_floats(B, d, H, 0),), **f32)
        self.bws_i = torch.empty((self.lib.rihip_tower_backward_workspace_floats(nI, d, H, 1),), **f32)
        self._nslab = [(0, 0), (0, 0)]
        self._tio = None
        self.fws_u = torch.empty((self.lib.rihip_tower_forward_workspace_floats(d, H, 0),), **f32)
        self.fws_i = torch.empty((self.lib.rihip_tower_forward_workspace_floats(d, H, 1),), **f32)
        self.np_mlp = self.lib.rihip_sumsq_nparts()
        self.np_rows = self.lib.rihip_rows_nparts() if table_opt == "sparse" else self.np_mlp
        self.part = torch.zeros((self.np_mlp + 2 * self.np_rows + 8,), dtype=torch.float64, device=self.dev)
        self.lpart = torch.zeros((max(1024, self.lib.rihip_inbatch_workspace_doubles(B)),), dtype=torch.float64,
                                 device=self.dev)
        Gall = B * self.world
        self.sws = torch.empty((max(self.lib.rihip_inbatch_workspace_floats(B, Gall, d),
                                    self.lib.rihip_inbatch_workspace_floats(Gall, B, d)),), **f32)
        self.n_lparts = self.lib.rihip_inbatch_loss_parts(B, Gall)
        self.gmat = None
        if loss_mode == "inbatch":
            self.pos = torch.empty((B,), **f32); self.r = torch.empty((B,), **f32)
            # stored-G form (exact f32 only): G^T of the local users x all items stays in HBM between the two passes
            # (17 GB at B = 65536 on one GPU, 2 GB per rank on eight) and removes the second score sweep
            ng = self.lib.rihip_inbatch_gmat_floats(B, Gall)
            store = self.inbatch_store_g
            if store is None:
                free_b = torch.cuda.mem_get_info(self.dev)[0]
                store = self.inbatch_precision in (0, 2) and 4 * ng <= 0.5 * free_b
            # (the stored-G passes exist for embed_dim 32/64/128; other widths run the runtime-width two-sweep kernel)
            self.inbatch_store_g = bool(store) and self.inbatch_precision in (0, 2) and d in (32, 64, 128)
            if self.inbatch_store_g:
                self.gmat = torch.empty((ng,), **f32)
            if self.dist:
                W = self.world
                self.I_all = torch.empty((W * B, d), **f32)
                if self.inbatch_store_g:
                    self.dI_all = torch.empty((W * B, d), **f32)
                self.U_all = None  # recompute form only: allocated on first use
        if self.dist and not self.item_rows:
            W = self.world
            self.iid_all = torch.empty((W * nI,), dtype=torch.int64, device=self.dev)
            self.dXi_all = torch.empty((W * nI, d), **f32)
        if self.item_rows:
            # routing state of one step (ids all-to-all -> rows all-to-all -> grads all-to-all): `cap` slots per peer
            W = self.world
            i64 = dict(dtype=torch.int64, device=self.dev)
            nS = self.n_slots = W * cap
            self.rt_ws = torch.empty((self.lib.rihip_route_workspace_bytes(nI),), dtype=torch.uint8, device=self.dev)
            self.rt_slot_ids = torch.empty((nS,), **i64)            # requester: owner-local row per send slot (0 = unused)
            self.rt_slot = torch.empty((nI,), **i64)                # requester: slot of each pair (= its row in rows_in)
            self.rt_counts = torch.empty((W,), **i64)               # requester: requests per owner (device only)
            self.req_ids = torch.empty((nS,), **i64)                # owner: local rows requested by every rank's slots
            self.rows_out = torch.empty((nS, d), **f32)             # owner: those rows, later their gradients
            self.rows_in = torch.empty((nS, d), **f32)              # requester: staging table the item tower reads
            self.dX_slots = torch.zeros((nS, d), **f32)             # requester: row gradients in their send slots

    # ------------------------------------------------------------------------------------------
    def _fwd(self, table, ids, genres, keys, out, hid, den, seed):
        training = self.model.training and self.p_drop > 0
        L.check(self.lib.rihip_tower_forward(table.data_ptr(), table.shape[0], ids.data_ptr(), L.ptr(genres),
                                             ids.numel(), self.d, self.H, self.pv[keys[0]].data_ptr(),
                                             self.pv[keys[1]].data_ptr(), self.pv[keys[2]].data_ptr(),
                                             self.pv[keys[3]].data_ptr(), 1 if training else 0, self.p_drop, seed, 0,
                                             out.data_ptr(), hid.data_ptr(), den.data_ptr(), self.err.data_ptr(),
                                             (self.fws_u if genres is None else self.fws_i).data_ptr(),
                                             self.step_dev.data_ptr(), self._st), "tower_forward")

    def _tower_io(self, ukeys, ikeys, user_ids, item_ids, item_genres, s0):
        """the two rihip_tower_io blocks of this step (single GPU, item table on this GPU)"""
        if self._tio is None:
            self._tio = (L.TowerIO(), L.TowerIO())
        for io, (tab, keys, ids, gen, out, hid, den, fws, gout, dX, bws, seed) in zip(self._tio, (
                (self.utab, ukeys, user_ids, None, self.U, self.hidU, self.denU, self.fws_u, self.dU, self.dXu, self.bws_u, s0),
                (self.itab, ikeys, item_ids, item_genres, self.I, self.hidI, self.denI, self.fws_i, self.dI, self.dXi,
                 self.bws_i, s0 + 1))):
            io.table, io.n_rows, io.ids, io.genres, io.B = tab.data_ptr(), tab.shape[0], ids.data_ptr(), L.ptr(gen), ids.numel()
            io.W1, io.b1, io.W2, io.b2 = (self.pv[k].data_ptr() for k in keys)
            io.seed, io.row0 = seed, 0
            io.out, io.hid, io.denom, io.fwd_workspace = out.data_ptr(), hid.data_ptr(), den.data_ptr(), fws.data_ptr()
            io.grad_out, io.dX, io.bwd_workspace = gout.data_ptr(), dX.data_ptr(), bws.data_ptr()
        return self._tio

    @staticmethod
    def _ev_handle(ev):
        if ev is None:
            return None
        if not ev.cuda_event:      # the handle only exists after a first record
            ev.record(torch.cuda.current_stream())
        return ev.cuda_event

    def _bwd(self, table, ids, genres, keys, gout, out, den, hid, dX, dx_event=None):
        """dx_event: a torch.cuda.Event that is recorded on the launch stream as soon as dX is complete (before the
        weight-gradient kernels)."""
        scale = 1.0 / (1.0 - self.p_drop) if (self.model.training and self.p_drop > 0) else 1.0
        ev = 0
        if dx_event is not None:
            if not dx_event.cuda_event:      # the handle only exists after a first record
                dx_event.record(torch.cuda.current_stream(self.dev))
            ev = dx_event.cuda_event
        # gradient kernels only: the weight-gradient slabs of the two towers are summed together by _bwd_reduce (one
        # pair of launches for both, after the item tower -- the user tower's reduce is off the critical path)
        ws = self.bws_u if genres is None else self.bws_i
        n = C.c_int(0)
        L.check(self.lib.rihip_tower_backward_partial(table.data_ptr(), table.shape[0], ids.data_ptr(), L.ptr(genres),
                                                      ids.numel(), self.d, self.H, self.pv[keys[0]].data_ptr(),
                                                      self.pv[keys[2]].data_ptr(), gout.data_ptr(), out.data_ptr(),
                                                      den.data_ptr(), hid.data_ptr(), scale, dX.data_ptr(),
                                                      ws.data_ptr(), self._st, ev, C.byref(n)), "tower_backward")
        self._nslab[0 if genres is None else 1] = (n.value, ids.numel())

    def _bwd_reduce(self, ukeys, ikeys):
        (nu, bu), (ni, bi) = self._nslab
        g = self.gv
        L.check(self.lib.rihip_tower_backward_reduce2(
            self.d, self.H, self.bws_u.data_ptr(), bu, 0, nu, g[ukeys[0]].data_ptr(), g[ukeys[1]].data_ptr(),
            g[ukeys[2]].data_ptr(), g[ukeys[3]].data_ptr(), self.bws_i.data_ptr(), bi, 1, ni, g[ikeys[0]].data_ptr(),
            g[ikeys[1]].data_ptr(), g[ikeys[2]].data_ptr(), g[ikeys[3]].data_ptr(), 0, self._st), "tower_backward_reduce2")

    def step(self, user_ids: torch.Tensor, item_ids: torch.Tensor, item_genres: torch.Tensor,
             lr: Optional[float] = None) -> torch.Tensor:
        """One optimisation step.  user_ids int64 [B] (LOCAL row ids when tables are sharded);
        sampled mode: item_ids [2B] = pos || neg, item_genres [2B,18]; inbatch: [B] / [B,18].
        Returns the loss as a 0-dim device tensor (no host sync).

        use_graph=True (single GPU): the second call captures the whole step (~45 launches) into a hipGraph;
        later calls copy the ids into static buffers and replay it.  The Adam step clock and the dropout seed
        offset live in device memory (rihip_adam_hyper_step), so replays advance them."""
        assert user_ids.numel() == self.B and item_ids.numel() == self.I.shape[0]
        lr = self.lr if lr is None else float(lr)
        if lr != self._lr_host:
            self.lr_dev.fill_(lr)
            self._lr_host = lr
        self.step_count += 1
        if not (self.use_graph and not self.dist and self.sweep_events is None):
            return self._step_impl(user_ids, item_ids, item_genres)
        if self._graph is None:
            if self._eager_steps < 1:      # first call eager: warms the library (rocPRIM temp queries, lazy module load)
                self._eager_steps += 1
                return self._step_impl(user_ids, item_ids, item_genres)
            self._s_u = user_ids.clone(); self._s_i = item_ids.clone(); self._s_g = item_genres.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step_impl(self._s_u, self._s_i, self._s_g)
            self._graph = g
        self._s_u.copy_(user_ids); self._s_i.copy_(item_ids); self._s_g.copy_(item_genres)
        self._graph.replay()
        return self.loss

    def _step_persistent(self, user_ids: torch.Tensor, item_ids: torch.Tensor, item_genres: torch.Tensor) -> torch.Tensor:
        """the whole step in one launch (rihip_bpr_step_persistent); the argument block is built once, only the three
        input pointers change from step to step"""
        a = self._pargs
        if a is None:
            ukeys, ikeys = _MLP_KEYS[:4], _MLP_KEYS[4:]
            a = self._pargs = L.StepArgs()
            s0 = (self.seed * 1000003 + self.rank * 7919) & ((1 << 62) - 1)
            for io, (tab, keys, out, hid, den, gout, dX, bws, seed, n) in zip((a.user, a.item), (
                    (self.utab, ukeys, self.U, self.hidU, self.denU, self.dU, self.dXu, self.bws_u, s0, self.B),
                    (self.itab, ikeys, self.I, self.hidI, self.denI, self.dI, self.dXi, self.bws_i, s0 + 1, self.nI))):
                io.table, io.n_rows, io.B = tab.data_ptr(), tab.shape[0], n
                io.W1, io.b1, io.W2, io.b2 = (self.pv[k].data_ptr() for k in keys)
                io.seed, io.row0 = seed, 0
                io.out, io.hid, io.denom, io.fwd_workspace = out.data_ptr(), hid.data_ptr(), den.data_ptr(), None
                io.grad_out, io.dX, io.bwd_workspace = gout.data_ptr(), dX.data_ptr(), bws.data_ptr()
            g = self.gv
            (a.dW1_u, a.db1_u, a.dW2_u, a.db2_u) = (g[k].data_ptr() for k in ukeys)
            (a.dW1_i, a.db1_i, a.dW2_i, a.db2_i) = (g[k].data_ptr() for k in ikeys)
            a.flat_p, a.flat_g, a.flat_m, a.flat_v = (x.data_ptr() for x in (self.flat_p, self.flat_g, self.flat_m, self.flat_v))
            a.n_flat = self.flat_p.numel()
            a.utab_g, a.utab_m, a.utab_v = self.uopt.grad.data_ptr(), self.uopt.m.data_ptr(), self.uopt.v.data_ptr()
            a.itab_g, a.itab_m, a.itab_v = self.iopt.grad.data_ptr(), self.iopt.m.data_ptr(), self.iopt.v.data_ptr()
            a.d, a.hidden = self.d, self.H
            a.beta1, a.beta2, a.eps, a.weight_decay, a.max_norm = self.b1, self.b2, self.eps, self.wd, self.max_norm
            a.lr_dev, a.step_dev, a.hyper_dev = self.lr_dev.data_ptr(), self.step_dev.data_ptr(), self.hyper_dev.data_ptr()
            a.coef, a.gnorm, a.loss, a.err_flag = (x.data_ptr() for x in (self.coef, self.gnorm, self.loss, self.err))
            nsd = int(self.lib.rihip_bpr_step_scratch_doubles(self.B))
            self._pscratch = torch.zeros((nsd,), dtype=torch.float64, device=self.dev)
            self._pbar = torch.zeros((4,), dtype=torch.int32, device=self.dev)
            a.scratch_doubles, a.n_scratch_doubles, a.barrier = self._pscratch.data_ptr(), nsd, self._pbar.data_ptr()
        a.user.ids, a.item.ids, a.item.genres = user_ids.data_ptr(), item_ids.data_ptr(), item_genres.data_ptr()
        a.user.genres = None
        a.training = 1 if (self.model.training and self.p_drop > 0) else 0
        a.dropout_p = self.p_drop
        L.check(self.lib.rihip_bpr_step_persistent(C.byref(a), L.stream_ptr()), "bpr_step_persistent")
        return self.loss

    def _step_impl(self, user_ids: torch.Tensor, item_ids: torch.Tensor, item_genres: torch.Tensor) -> torch.Tensor:
        if self.persistent:
            return self._step_persistent(user_ids, item_ids, item_genres)
        lib, B, d = self.lib, self.B, self.d
        self._st = st = L.stream_ptr()
        t, lr = 0, 0.0   # the device clock (hyper_dev) overrides the host-side step / lr arguments below
        ukeys, ikeys = _MLP_KEYS[:4], _MLP_KEYS[4:]
        s0 = (self.seed * 1000003 + self.rank * 7919) & ((1 << 62) - 1)   # + device step counter inside the kernel
        self._I_work = None
        itab, iids = self.itab, item_ids
        sparse = self.table_opt == "sparse"
        cur = torch.cuda.current_stream(self.dev)
        sideA, sideB = self._side_stream, self._side_stream2
        early_item_group = sparse and not self.dist       # distributed: the owner's id list only exists after the exchange
        ev_in = None
        if sparse:
            ev_in = torch.cuda.Event()
            ev_in.record(cur)      # inputs + everything of the previous step; recorded BEFORE this step's first launch

        # The id sort / unique-row search (~12 small dependent kernels per table) needs only the ids: it runs on side
        # streams beside the towers instead of after them (150 us of launch-bound tail per step).  HOST order matters as
        # much as stream order: each group is ~12 launches = ~70 us of host time, so they are enqueued only once the
        # host is well ahead of the GPU (after the forward + loss launches / after the user tower's backward).
        def early_group_user():
            if sparse:
                sideA.wait_event(ev_in)
                with torch.cuda.stream(sideA):
                    self.uopt.group(user_ids, sideA.cuda_stream)

        def early_group_item():
            if early_item_group:
                sideB.wait_event(ev_in)
                with torch.cuda.stream(sideB):
                    self.iopt.group(item_ids, sideB.cuda_stream)
        if self.item_rows:
            # row-sharded item table: the rows travel (all-to-all) under the user tower, then the item tower reads the
            # received rows as a [nI,d] staging table indexed by each pair's send slot
            w_rows = self._fetch_item_rows(item_ids, st)
            if sparse:
                # the owner's id list of this step is known as soon as the ids all-to-all lands: its sort / unique-row
                # search (~12 small dependent kernels) runs beside the towers, not after the gradient exchange
                ev_ids = torch.cuda.Event()
                ev_ids.record(cur)
                sideB.wait_event(ev_ids)
                with torch.cuda.stream(sideB):
                    self.iopt.group(self.req_ids, sideB.cuda_stream)
            self._fwd(self.utab, user_ids, None, ukeys, self.U, self.hidU, self.denU, s0)
            w_rows.wait()
            itab, iids = self.rows_in, self.rt_slot
            self._fwd(itab, iids, item_genres, ikeys, self.I, self.hidI, self.denI, s0 + 1)
            if self.loss_mode == "inbatch":
                self._I_work = all_gather_into(self.I_all, self.I, self.pg, async_op=True)
        else:
            # item tower first: in the multi-GPU in-batch mode its outputs travel (all-gather) under the user tower
            if not self.dist:   # both towers in one launch (small batches; large ones take the two chip-filling kernels)
                tio = self._tower_io(ukeys, ikeys, user_ids, item_ids, item_genres, s0)
                training = self.model.training and self.p_drop > 0
                L.check(lib.rihip_tower_forward_pair(C.byref(tio[0]), C.byref(tio[1]), d, self.H, 1 if training else 0,
                                                     self.p_drop, self.err.data_ptr(), self.step_dev.data_ptr(), st),
                        "tower_forward_pair")
            else:
                self._fwd(self.itab, item_ids, item_genres, ikeys, self.I, self.hidI, self.denI, s0 + 1)
                if self.loss_mode == "inbatch":
                    self._I_work = all_gather_into(self.I_all, self.I, self.pg, async_op=True)
                self._fwd(self.utab, user_ids, None, ukeys, self.U, self.hidU, self.denU, s0)

        if self.loss_mode == "sampled":
            # single GPU: the loss partials are summed by the clip-coefficient launch (one dependent launch less)
            L.check(lib.rihip_bpr_pair_loss(self.U.data_ptr(), self.I.data_ptr(), self.I[B:].data_ptr(), B, d,
                                            self.loss.data_ptr() if self.dist else None, self.dU.data_ptr(),
                                            self.dI.data_ptr(), self.dI[B:].data_ptr(), self.lpart.data_ptr(), st),
                    "bpr_pair_loss")
            self._loss_fold = None if self.dist else (self.lpart.data_ptr(), lib.rihip_bpr_pair_nparts(B), 1.0 / B)
            if self.dist:  # mean over the global batch
                self.dU.div_(self.world); self.dI.div_(self.world)
                self.loss.div_(self.world)   # (summed over the ranks together with the squared norms below)
        else:
            self._inbatch(st)

        pp = self.part.data_ptr()
        o1 = self.np_mlp
        o2 = o1 + self.np_rows
        pair_bwd = not self.dist and not sparse   # dense tables = the small-batch regime: both towers in one launch
        if pair_bwd:
            tio = self._tower_io(ukeys, ikeys, user_ids, item_ids, item_genres, s0)
            scale = 1.0 / (1.0 - self.p_drop) if (self.model.training and self.p_drop > 0) else 1.0
            nu, ni = C.c_int(0), C.c_int(0)
            L.check(lib.rihip_tower_backward_partial_pair(C.byref(tio[0]), C.byref(tio[1]), d, self.H, scale, st, None, None,
                                                          C.byref(nu), C.byref(ni)), "tower_backward_partial_pair")
            self._nslab = [(nu.value, user_ids.numel()), (ni.value, item_ids.numel())]
        early_group_user()
        # user tower first: in the multi-GPU stored-G form dI is still being reduce-scattered
        if not pair_bwd:
            self._bwd(self.utab, user_ids, None, ukeys, self.dU, self.U, self.denU, self.hidU, self.dXu,
                      dx_event=self._ev_dxu if sparse else None)
        if sparse:   # the user rows' segment sums start as soon as dXu exists: beside the user tower's weight gradients
            sideA.wait_event(self._ev_dxu)
            with torch.cuda.stream(sideA):
                self.uopt.reduce(self.dXu, pp + 8 * o1, sideA.cuda_stream)
        early_group_item()
        if self._dI_work is not None:
            self._dI_work.wait(); self._dI_work = None
        if not pair_bwd:
            self._bwd(itab, iids, item_genres, ikeys, self.dI, self.I, self.denI, self.hidI, self.dXi,
                      dx_event=self._ev_dxi if early_item_group else None)
        if early_item_group:   # ... and the item rows' beside the item tower's weight gradients
            sideB.wait_event(self._ev_dxi)
            with torch.cuda.stream(sideB):
                self.iopt.reduce(self.dXi, pp + 8 * o2, sideB.cuda_stream)
        fuse_scatter = self.table_opt == "dense" and not self.dist   # (dense tables are single-GPU)
        if not fuse_scatter:
            self._bwd_reduce(ukeys, ikeys)

        # ---- gradient exchange (multi-GPU) + global grad norm -> clip coef (device scalar)
        iid, dXi = item_ids, self.dXi
        w_i = w_x = None
        if self.item_rows:
            # row gradients return to the owners of the rows in the slots the ids went out in (unused slots carry
            # whatever they held: the owner sees id 0 there and drops them)
            nS = self.n_slots
            L.check(lib.rihip_scatter_rows(self.dXi.data_ptr(), self.rt_slot.data_ptr(), self.nI, nS, d,
                                           self.dX_slots.data_ptr(), self.err.data_ptr(), st), "scatter_rows")
            eq = [self.exchange_cap] * self.world
            w_x = all_to_all_rows(self.rows_out, self.dX_slots, eq, eq, self.pg, async_op=True)
            all_reduce_sum_(self.flat_g, self.pg)
            iid, dXi = self.req_ids, self.rows_out
        elif self.dist:
            # the item row gradients travel while the MLP all-reduce and the user-row grouping run
            w_i = all_gather_into(self.iid_all, item_ids, self.pg, async_op=True)
            w_x = all_gather_into(self.dXi_all, self.dXi, self.pg, async_op=True)
            all_reduce_sum_(self.flat_g, self.pg)
            iid, dXi = self.iid_all, self.dXi_all
        dense = self.table_opt == "dense"
        if not dense:
            L.check(lib.rihip_sumsq(self.flat_g.data_ptr(), self.flat_g.numel(), pp, st), "sumsq")
        if sparse:
            if w_i is not None:
                w_i.wait()
            if w_x is not None:
                w_x.wait()
            if iid.numel() > 0:
                if early_item_group:
                    cur.wait_stream(sideB)
                elif self.item_rows:      # grouped beside the towers (above); the sums need the gradients that just landed
                    cur.wait_stream(sideB)
                    self.iopt.reduce(dXi, pp + 8 * o2, st)
                else:
                    self.iopt.group_reduce(iid, dXi, pp + 8 * o2, st)
            else:   # this rank owns none of the rows of the step
                self.part[o2:o2 + self.np_rows].zero_()
                self.iopt._B = 0
            cur.wait_stream(sideA)
        else:
            # dense tables (ML-1M scale): the table gradients were zeroed by the previous step's Adam launch
            # both tables in one launch, which also carries the weight-gradient slab reduction of both towers (both wait
            # only for the tower backward)
            uo, io = self.uopt, self.iopt
            (nu, bu), (ni, bi) = self._nslab
            g = self.gv
            L.check(lib.rihip_backward_reduce2_scatter2(
                d, self.H, self.bws_u.data_ptr(), bu, 0, nu, g[ukeys[0]].data_ptr(), g[ukeys[1]].data_ptr(),
                g[ukeys[2]].data_ptr(), g[ukeys[3]].data_ptr(), self.bws_i.data_ptr(), bi, 1, ni, g[ikeys[0]].data_ptr(),
                g[ikeys[1]].data_ptr(), g[ikeys[2]].data_ptr(), g[ikeys[3]].data_ptr(), 0,
                uo.grad.data_ptr(), uo.table.shape[0], user_ids.data_ptr(), self.dXu.data_ptr(), user_ids.numel(),
                io.grad.data_ptr(), io.table.shape[0], iid.data_ptr(), dXi.data_ptr(), iid.numel(), st),
                "backward_reduce2_scatter2")
        n_part = o2 + self.np_rows
        if self.dist:
            # user rows are disjoint across ranks: their squared norms add; the MLP part is already global; the item
            # part is global when the table is replicated and disjoint (-> added up) when it is row-sharded
            # the loss value (only reported, never consumed by the step) rides in the same small all-reduce
            sq = torch.stack([self.part[o1:o2].sum(), self.part[o2:n_part].sum(), self.loss.double()])
            all_reduce_sum_(sq, self.pg)
            self.loss.copy_(sq[2])
            self.part[o1:o2].zero_()
            self.part[o1] = sq[0]
            if self.item_rows:
                self.part[o2:n_part].zero_()
                self.part[o2] = sq[1]
        clock = (self.step_dev.data_ptr(), self.lr_dev.data_ptr(), self.b1, self.b2, self.hyper_dev.data_ptr())
        if dense:   # one launch for the three squared norms (MLP, user table, item table gradients)
            L.check(lib.rihip_sumsq_multi(3, self._mt_g, self._mt_n, pp, st), "sumsq_multi")
        # clip coefficient + the step clock (Adam's bias-corrected step size of this step; counter advanced for the next)
        lf = self._loss_fold
        L.check(lib.rihip_clip_coef_step(pp, n_part, self.max_norm, self.coef.data_ptr(), self.gnorm.data_ptr(),
                                         *clock, lf[0] if lf else None, lf[1] if lf else 0, lf[2] if lf else 0.0,
                                         self.loss.data_ptr(), st), "clip_coef_step")
        cp = self.coef.data_ptr()
        hp = self.hyper_dev.data_ptr()
        if dense:   # one Adam launch for MLP + both tables; it leaves the table gradients zeroed for the next scatter
            L.check(lib.rihip_adam_dense_multi(3, self._mt_p, self._mt_g, self._mt_m, self._mt_v, self._mt_n, 0b110, lr,
                                               self.b1, self.b2, self.eps, self.wd, t, cp, hp, st), "adam_dense_multi")
            return self.loss
        L.check(lib.rihip_adam_dense(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                     self.flat_v.data_ptr(), self.flat_p.numel(), lr, self.b1, self.b2, self.eps,
                                     self.wd, t, cp, hp, st), "adam_dense")
        # both row-sparse Adam launches on the main stream: they are HBM-bound, side by side they take as long as one
        # after the other, and the two stream joins cost more than the one kernel boundary
        self.uopt.adam(lr, self.b1, self.b2, self.eps, self.wd, t, cp, st, hp)
        if self.iopt._B > 0:
            self.iopt.adam(lr, self.b1, self.b2, self.eps, self.wd, t, cp, st, hp)
        return self.loss

    def _fetch_item_rows(self, item_ids: torch.Tensor, st: int):
        """Row-sharded item table, forward half of the exchange: route the step's global item ids into `cap` send slots
        per owner, all-to-all the owner-local row numbers (equal splits: nothing crosses to the host), gather the
        requested rows here, all-to-all them back (async handle).  Leaves rt_slot (slot of each pair = its row in
        rows_in) and req_ids (0 = the padding row in unused slots)."""
        lib, d, nI, nS = self.lib, self.d, self.nI, self.n_slots
        L.check(lib.rihip_route_rows_fixed(item_ids.data_ptr(), nI, self.world, self.exchange_cap,
                                           self.rt_slot_ids.data_ptr(), self.rt_slot.data_ptr(), self.rt_counts.data_ptr(),
                                           self.err.data_ptr(), self.rt_ws.data_ptr(), self.rt_ws.numel(), st),
                "route_rows_fixed")
        eq = [self.exchange_cap] * self.world
        all_to_all_rows(self.req_ids, self.rt_slot_ids, eq, eq, self.pg)
        L.check(lib.rihip_gather_rows(self.itab.data_ptr(), self.itab.shape[0], self.req_ids.data_ptr(), nS, d,
                                      self.rows_out.data_ptr(), self.err.data_ptr(), st), "gather_rows")
        return all_to_all_rows(self.rows_in, self.rows_out, eq, eq, self.pg, async_op=True)

    def check_errors(self) -> None:
        """Read the device error word (ONE host sync: call it per epoch / at the end of a run, not per step) and raise:
        bit 1 = an id outside its table (negative, or beyond the rows of the shard), bit 2 = more than exchange_cap rows
        routed to one owner in a step."""
        e = int(self.err.item())
        if e:
            self.err.zero_()
            what = []
            if e & 1:
                what.append("an id outside its embedding table")
            if e & 2:
                what.append(f"more than exchange_cap={getattr(self, 'exchange_cap', 0)} rows routed to one owner")
            raise RuntimeError("HipBPRTrainer: " + " and ".join(what or [f"device error word {e}"]))

    def _timed(self, fn, what, *args) -> None:
        ev = self.sweep_events
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(fn(*args), what)
        if ev is not None:
            e1.record()
            ev.append((what, e0, e1))

    def _sweep(self, *args) -> None:
        self._timed(self.lib.rihip_inbatch_sweep, "inbatch_sweep", *args)

    def _inbatch(self, st: int) -> None:
        lib, B, d, W = self.lib, self.B, self.d, self.world
        G, off = W * B, self.rank * B
        L.check(lib.rihip_rowdot(self.U.data_ptr(), self.I.data_ptr(), B, 0, d, self.pos.data_ptr(), st), "rowdot")
        I_all = self.I
        if self.dist:
            # global in-batch negatives: every rank scores its users against ALL items (4 MiB/rank at B=8192, d=128);
            # the all-gather was started right after the item tower's forward
            self._I_work.wait()
            I_all = self.I_all
        if self.inbatch_store_g and self.inbatch_precision in (0, 2):
            # user pass keeps G in HBM; item pass = G^T.U over the local users for all items, reduce-scattered to
            # the rank that owns each item's tower backward (no second score sweep, no U/pos/r gathers)
            self._timed(lib.rihip_inbatch_user_pass, "inbatch_user_pass", self.U.data_ptr(), B, off, I_all.data_ptr(),
                        G, 0, d, self.pos.data_ptr(), G, self.dU.data_ptr(), self.r.data_ptr(), self.lpart.data_ptr(),
                        self.sws.data_ptr(), self.gmat.data_ptr(), self.inbatch_precision, st)
            dI_all = self.dI if not self.dist else self.dI_all
            self._timed(lib.rihip_inbatch_item_pass, "inbatch_item_pass", self.gmat.data_ptr(), self.U.data_ptr(), B,
                        off, G, 0, d, self.r.data_ptr(), G, dI_all.data_ptr(), self.sws.data_ptr(), self.inbatch_precision, st)
            if self.dist:
                self._dI_work = reduce_scatter_sum(self.dI, self.dI_all, self.pg, async_op=True)
        elif not self.dist:
            self._sweep(1, self.U.data_ptr(), B, 0, self.I.data_ptr(), B, 0, d,
                                            self.pos.data_ptr(), None, B, self.dU.data_ptr(), self.r.data_ptr(),
                                            self.lpart.data_ptr(), self.sws.data_ptr(), self.inbatch_precision, st)
            self._sweep(0, self.I.data_ptr(), B, 0, self.U.data_ptr(), B, 0, d,
                                            self.pos.data_ptr(), self.r.data_ptr(), B, self.dI.data_ptr(), None,
                                            None, self.sws.data_ptr(), self.inbatch_precision, st)
        else:
            # recompute form: each rank sweeps its own items against ALL users
            if self.U_all is None:
                f32 = dict(dtype=torch.float32, device=self.dev)
                self.U_all = torch.empty((G, d), **f32)
                self.pos_all = torch.empty((G,), **f32); self.r_all = torch.empty((G,), **f32)
            # U_all / pos_all are only read by the item-mode sweep: they travel under the user-mode sweep
            w_u = all_gather_into(self.U_all, self.U, self.pg, async_op=True)
            w_p = all_gather_into(self.pos_all, self.pos, self.pg, async_op=True)
            self._sweep(1, self.U.data_ptr(), B, off, self.I_all.data_ptr(), G, 0, d,
                                            self.pos.data_ptr(), None, G, self.dU.data_ptr(), self.r.data_ptr(),
                                            self.lpart.data_ptr(), self.sws.data_ptr(), self.inbatch_precision, st)
            all_gather_into(self.r_all, self.r, self.pg)
            w_u.wait(); w_p.wait()
            self._sweep(0, self.I.data_ptr(), B, off, self.U_all.data_ptr(), G, 0, d,
                                            self.pos_all.data_ptr(), self.r_all.data_ptr(), G, self.dI.data_ptr(), None,
                                            None, self.sws.data_ptr(), self.inbatch_precision, st)
        if self.dist:
            L.check(lib.rihip_sum_partials(self.lpart.data_ptr(), self.n_lparts, 1.0 / (G * (G - 1.0)),
                                           self.loss.data_ptr(), st), "sum_partials")
            self._loss_fold = None   # (summed over the ranks together with the squared norms in step())
        else:   # summed by the clip-coefficient launch
            self._loss_fold = (self.lpart.data_ptr(), self.n_lparts, 1.0 / (G * (G - 1.0)))


def cosine_lr(lr0: float, epoch: int, t_max: int) -> float:
    """CosineAnnealingLR(T_max, eta_min=0) closed form; `epoch` = scheduler.step() calls so far
    (reference steps it once per epoch: train_embeddings.py:161,197)."""
    return 0.5 * lr0 * (1.0 + math.cos(math.pi * epoch / t_max))
