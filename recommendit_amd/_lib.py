"""ctypes binding of librecommendit_hip.so (the C ABI declared in include/recommendit_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (or ``make -C recommendit_amd/csrc``).
There is NO CPU fallback: if the library is missing, ``lib()`` raises ImportError; if no HIP
device is visible, every compute entry raises RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "librecommendit_hip.so"
_lib: Optional[C.CDLL] = None

c_f32p = C.c_void_p
c_i64 = C.c_int64
vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/recommendit_hip.h one-to-one
SIGNATURES = {
    "rihip_abi_version": (C.c_int, []),
    "rihip_target_arch": (C.c_char_p, []),
    "rihip_last_error": (C.c_char_p, []),
    "rihip_scratch_generation": (C.c_uint64, []),
    "rihip_device_arch": (C.c_int, [C.c_char_p, C.c_int]),
    "rihip_tower_supported": (C.c_int, [C.c_int, C.c_int]),
    "rihip_tower_shape_ok": (C.c_int, [C.c_int, C.c_int]),
    "rihip_tower_forward_workspace_floats": (c_i64, [C.c_int, C.c_int, C.c_int]),
    "rihip_tower_forward": (C.c_int, [vp, c_i64, vp, vp, c_i64, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_float,
                                      C.c_uint64, c_i64, vp, vp, vp, vp, vp, vp, vp]),
    "rihip_tower_backward_workspace_floats": (c_i64, [c_i64, C.c_int, C.c_int, C.c_int]),
    "rihip_tower_backward": (C.c_int, [vp, c_i64, vp, vp, c_i64, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_float,
                                       vp, vp, vp, vp, vp, C.c_int, vp, vp]),
    "rihip_tower_backward_ev": (C.c_int, [vp, c_i64, vp, vp, c_i64, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_float,
                                          vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]),
    "rihip_embedding_scatter_add": (C.c_int, [vp, c_i64, vp, vp, c_i64, C.c_int, vp]),
    "rihip_embedding_scatter_add2": (C.c_int, [vp, c_i64, vp, vp, c_i64, vp, c_i64, vp, vp, c_i64, C.c_int, vp]),
    "rihip_bpr_pair_loss": (C.c_int, [vp, vp, vp, c_i64, C.c_int, vp, vp, vp, vp, vp, vp]),
    "rihip_rowdot": (C.c_int, [vp, vp, c_i64, c_i64, C.c_int, vp, vp]),
    "rihip_inbatch_workspace_doubles": (c_i64, [c_i64]),
    "rihip_inbatch_loss_parts": (c_i64, [c_i64, c_i64]),
    "rihip_inbatch_workspace_floats": (c_i64, [c_i64, c_i64, C.c_int]),
    "rihip_inbatch_sweep": (C.c_int, [C.c_int, vp, c_i64, c_i64, vp, c_i64, c_i64, C.c_int, vp, vp, c_i64, vp, vp, vp,
                                      vp, C.c_int, vp]),
    "rihip_sum_partials": (C.c_int, [vp, c_i64, C.c_double, vp, vp]),
    "rihip_inbatch_gmat_floats": (c_i64, [c_i64, c_i64]),
    "rihip_inbatch_user_pass": (C.c_int, [vp, c_i64, c_i64, vp, c_i64, c_i64, C.c_int, vp, c_i64, vp, vp, vp, vp, vp,
                                          C.c_int, vp]),
    "rihip_inbatch_item_pass": (C.c_int, [vp, vp, c_i64, c_i64, c_i64, c_i64, C.c_int, vp, c_i64, vp, vp, C.c_int,
                                          vp]),
    "rihip_sumsq_nparts": (C.c_int, []),
    "rihip_sumsq_multi": (C.c_int, [C.c_int, vp, vp, vp, vp]),
    "rihip_adam_dense_multi": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, C.c_float,
                                         C.c_float, C.c_float, c_i64, vp, vp, vp]),
    "rihip_sumsq": (C.c_int, [vp, c_i64, vp, vp]),
    "rihip_clip_coef": (C.c_int, [vp, c_i64, C.c_float, vp, vp, vp]),
    "rihip_adam_dense": (C.c_int, [vp, vp, vp, vp, c_i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, c_i64,
                                   vp, vp, vp]),
    "rihip_adam_hyper_step": (C.c_int, [vp, vp, C.c_float, C.c_float, vp, vp]),
    "rihip_clip_coef_step": (C.c_int, [vp, c_i64, C.c_float, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, c_i64,
                                       C.c_double, vp, vp]),
    "rihip_bpr_pair_nparts": (c_i64, [c_i64]),
    "rihip_tower_backward_partial": (C.c_int, [vp, c_i64, vp, vp, c_i64, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp,
                                               C.c_float, vp, vp, vp, vp, vp]),
    "rihip_backward_reduce2_scatter2": (C.c_int, [C.c_int, C.c_int, vp, c_i64, C.c_int, C.c_int, vp, vp, vp, vp, vp, c_i64,
                                                  C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, c_i64, vp, vp, c_i64, vp,
                                                  c_i64, vp, vp, c_i64, vp]),
    "rihip_tower_forward_pair": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp]),
    "rihip_tower_backward_partial_pair": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp, vp]),
    "rihip_tower_backward_reduce2": (C.c_int, [C.c_int, C.c_int, vp, c_i64, C.c_int, C.c_int, vp, vp, vp, vp, vp, c_i64,
                                               C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp]),
    "rihip_rows_workspace_bytes": (c_i64, [c_i64, C.c_int]),
    "rihip_rows_nparts": (C.c_int, []),
    "rihip_rows_group": (C.c_int, [vp, c_i64, C.c_int, c_i64, vp, vp, c_i64, vp]),
    "rihip_rows_n_unique_ptr": (C.c_int, [vp, c_i64, C.c_int, C.POINTER(vp)]),
    "rihip_rows_reduce": (C.c_int, [vp, c_i64, C.c_int, vp, vp, vp, vp, vp]),
    "rihip_adam_rows": (C.c_int, [vp, vp, vp, vp, vp, c_i64, C.c_int, vp, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_float, c_i64, vp, vp, vp]),
    "rihip_route_workspace_bytes": (c_i64, [c_i64]),
    "rihip_route_rows": (C.c_int, [vp, c_i64, C.c_int, vp, vp, vp, vp, vp, vp, c_i64, vp]),
    "rihip_gather_rows": (C.c_int, [vp, c_i64, vp, c_i64, C.c_int, vp, vp, vp]),
    "rihip_route_rows_fixed": (C.c_int, [vp, c_i64, C.c_int, c_i64, vp, vp, vp, vp, vp, c_i64, vp]),
    "rihip_scatter_rows": (C.c_int, [vp, vp, c_i64, c_i64, C.c_int, vp, vp, vp]),
    "rihip_ip_index_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "rihip_ip_index_destroy": (C.c_int, [vp]),
    "rihip_ip_index_set_vectors": (C.c_int, [vp, vp, c_i64, C.c_int, vp]),
    "rihip_ip_index_ntotal": (c_i64, [vp]),
    "rihip_ip_index_is_ivf": (C.c_int, [vp]),
    "rihip_ip_index_max_k": (C.c_int, []),
    "rihip_ip_index_train_ivf": (C.c_int, [vp, C.c_int, C.c_int, C.c_uint64, vp]),
    "rihip_ip_index_nlist": (C.c_int, [vp]),
    "rihip_ip_index_train_ivf_from": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "rihip_ip_index_set_ivf": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "rihip_ip_index_get_ivf": (C.c_int, [vp, vp, vp]),
    "rihip_ip_index_reconstruct": (C.c_int, [vp, vp]),
    "rihip_ip_index_assign": (C.c_int, [vp, vp, c_i64, vp, vp]),
    "rihip_ip_index_set_nprobe": (C.c_int, [vp, C.c_int]),
    "rihip_ip_index_set_two_precision": (C.c_int, [vp, C.c_int]),
    "rihip_ip_index_search": (C.c_int, [vp, vp, c_i64, C.c_int, vp, vp, vp]),
    "rihip_ip_index_set_deferred_check": (C.c_int, [vp, C.c_int]),
    "rihip_ip_index_search_finish": (C.c_int, [vp, vp, vp]),
    "rihip_ip_index_search_pending": (C.c_int, [vp]),
    "rihip_ip_index_last_fail_count": (C.c_int, [vp, vp, vp]),
    "rihip_ip_index_save": (C.c_int, [vp, C.c_char_p]),
    "rihip_ip_index_load": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "rihip_map_rows_to_ids": (C.c_int, [vp, c_i64, vp, vp]),
    "rihip_ip_index_set_id_map": (C.c_int, [vp, vp]),
    "rihip_gbdt_load_text": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "rihip_gbdt_create_from_text": (C.c_int, [C.c_char_p, c_i64, C.POINTER(vp)]),
    "rihip_gbdt_destroy": (C.c_int, [vp]),
    "rihip_gbdt_num_trees": (C.c_int, [vp]),
    "rihip_gbdt_num_features": (C.c_int, [vp]),
    "rihip_gbdt_feature_names": (c_i64, [vp, C.c_char_p, c_i64]),
    "rihip_gbdt_feature_importance": (C.c_int, [vp, C.c_int, vp]),
    "rihip_gbdt_predict": (C.c_int, [vp, vp, c_i64, C.c_int, vp, vp]),
    "rihip_lambdamart_train": (C.c_int, [vp, vp, vp, c_i64, C.c_int, C.c_int, vp, vp, vp, c_i64, C.c_int, vp, C.c_char_p,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), vp, vp]),
    "rihip_free": (None, [vp]),
    "rihip_sample_negatives": (C.c_int, [vp, c_i64, vp, c_i64, vp, c_i64, c_i64, C.c_uint64, C.c_int, vp, vp, vp]),
    "rihip_bpr_step_persistent_supported": (C.c_int, [c_i64, C.c_int, C.c_int]),
    "rihip_bpr_step_scratch_doubles": (c_i64, [c_i64]),
    "rihip_bpr_step_persistent": (C.c_int, [vp, vp]),
    "rihip_rank_features_widths": (C.c_int, [vp, vp, vp]),
    "rihip_rank_topk": (C.c_int, [vp, vp, vp, c_i64, C.c_int, C.c_int, vp, vp, vp, vp]),
    "rihip_rank_features_build": (C.c_int, [vp, c_i64, vp, c_i64, vp, vp, c_i64, C.c_int, vp, C.c_int, vp, vp]),
}


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; ImportError if it was never built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C recommendit_amd/csrc`. recommendit_amd has no CPU fallback."
            )
        l = C.CDLL(str(LIB_PATH), mode=os.RTLD_GLOBAL if hasattr(os, "RTLD_GLOBAL") else C.DEFAULT_MODE)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError here == header/library drift
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class LambdamartParams(C.Structure):
    """mirror of rihip_lambdamart_params (include/recommendit_hip.h)"""
    _fields_ = [("num_leaves", C.c_int), ("n_estimators", C.c_int), ("min_child_samples", C.c_int), ("max_bin", C.c_int),
                ("truncation_level", C.c_int), ("early_stopping_rounds", C.c_int), ("lambdarank_norm", C.c_int),
                ("bin_sample", C.c_int), ("n_eval_at", C.c_int), ("eval_at", C.c_int * 8), ("n_label_gain", C.c_int),
                ("label_gain", C.c_double * 32), ("learning_rate", C.c_double), ("reg_alpha", C.c_double),
                ("reg_lambda", C.c_double), ("feature_fraction", C.c_double), ("min_sum_hessian", C.c_double),
                ("sigmoid", C.c_double), ("seed", C.c_uint64), ("hist_bits", C.c_int), ("use_missing", C.c_int),
                ("split_order", C.c_int), ("reserved", C.c_int)]


class TowerIO(C.Structure):
    """mirror of rihip_tower_io (include/recommendit_hip.h): one tower of a step for the *_pair entry points"""
    _fields_ = [("table", C.c_void_p), ("n_rows", C.c_int64), ("ids", C.c_void_p), ("genres", C.c_void_p), ("B", C.c_int64),
                ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p), ("seed", C.c_uint64),
                ("row0", C.c_int64), ("out", C.c_void_p), ("hid", C.c_void_p), ("denom", C.c_void_p),
                ("fwd_workspace", C.c_void_p), ("grad_out", C.c_void_p), ("dX", C.c_void_p), ("bwd_workspace", C.c_void_p)]


class StepArgs(C.Structure):
    """mirror of rihip_step_args (include/recommendit_hip.h): the one-launch sampled-negative training step"""
    _fields_ = ([("user", TowerIO), ("item", TowerIO)]
                + [(n, C.c_void_p) for n in ("dW1_u", "db1_u", "dW2_u", "db2_u", "dW1_i", "db1_i", "dW2_i", "db2_i",
                                             "flat_p", "flat_g", "flat_m", "flat_v")]
                + [("n_flat", C.c_int64)]
                + [(n, C.c_void_p) for n in ("utab_g", "utab_m", "utab_v", "itab_g", "itab_m", "itab_v")]
                + [("d", C.c_int), ("hidden", C.c_int), ("training", C.c_int), ("dropout_p", C.c_float),
                   ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
                   ("max_norm", C.c_float)]
                + [(n, C.c_void_p) for n in ("lr_dev", "step_dev", "hyper_dev", "coef", "gnorm", "loss", "err_flag",
                                             "scratch_doubles")]
                + [("n_scratch_doubles", C.c_int64), ("barrier", C.c_void_p)])


class RihipError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().rihip_last_error().decode("utf-8", "replace")
        raise RihipError(f"{what or 'recommendit_hip'} failed (status {rc}): {msg}")


def have_gpu() -> bool:
    return torch.cuda.is_available()


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("recommendit_amd: no HIP device visible (the product path has no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def f32c(t: torch.Tensor) -> torch.Tensor:
    """contiguous float32 view/copy on the HIP device"""
    return t.to(device=device(), dtype=torch.float32).contiguous()


def i64c(t: torch.Tensor) -> torch.Tensor:
    return t.to(device=device(), dtype=torch.int64).contiguous()
