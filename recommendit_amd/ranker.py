"""Drop-in LightGBMRanker whose load/predict run in the gfx950 HIP library (no lightgbm).

Mirrors the reference's src/models/ranker.py (:23-249).  ``load`` parses the LightGBM *text*
model in C++ (rihip_gbdt_load_text) and ``predict`` walks the forest on the GPU, returning the
raw lambdarank score as float64 like ``Booster.predict`` (ranker.py:174).  ``train`` runs the
library's own LambdaMART trainer on the GPU (csrc/gbdt_train.hip: LightGBM's published histogram /
leaf-wise / lambdarank algorithm restated) with the reference's parameters, or --
``backend="lightgbm"``, the default when the real package is importable -- delegates to lightgbm
like the reference does.  Either way the trained forest is served by the HIP predictor.
"""
from __future__ import annotations

import ctypes as C
import logging
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import pandas as pd
import torch

from . import _lib as L

logger = logging.getLogger(__name__)

try:  # only needed for train()
    import lightgbm as lgb  # type: ignore
    LGB_AVAILABLE = True
except ImportError:
    lgb = None
    LGB_AVAILABLE = False


class _Forest:
    """Owns one rihip gbdt handle; method names follow lgb.Booster where callers use them."""

    def __init__(self, handle: int, path: Optional[str] = None):
        self._h = C.c_void_p(handle)
        self.path = path
        self.best_iteration = -1  # a Booster loaded from a model file carries no best_iteration

    def num_trees(self) -> int:
        return int(L.lib().rihip_gbdt_num_trees(self._h))

    def num_feature(self) -> int:
        return int(L.lib().rihip_gbdt_num_features(self._h))

    def feature_name(self) -> List[str]:
        n = int(L.lib().rihip_gbdt_feature_names(self._h, None, 0))
        buf = C.create_string_buffer(n)
        L.lib().rihip_gbdt_feature_names(self._h, buf, n)
        s = buf.value.decode()
        return s.split("\n") if s else []

    def feature_importance(self, importance_type: str = "split") -> np.ndarray:
        out = np.zeros(self.num_feature(), dtype=np.float64)
        L.check(L.lib().rihip_gbdt_feature_importance(self._h, 0 if importance_type == "split" else 1,
                                                      out.ctypes.data_as(C.c_void_p)), "gbdt_feature_importance")
        return out.astype(np.int64) if importance_type == "split" else out

    def predict_device(self, X: torch.Tensor) -> torch.Tensor:
        """X f32 [n, >=n_features] on device -> f64 [n] on device."""
        X = X.to(dtype=torch.float32).contiguous()
        out = torch.empty((X.shape[0],), dtype=torch.float64, device=X.device)
        L.check(L.lib().rihip_gbdt_predict(self._h, X.data_ptr(), X.shape[0], X.shape[1], out.data_ptr(),
                                           L.stream_ptr()), "gbdt_predict")
        return out

    def predict(self, X: np.ndarray) -> np.ndarray:
        Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(L.device())
        return self.predict_device(Xd).cpu().numpy()

    def __del__(self):
        try:
            if self._h:
                L.lib().rihip_gbdt_destroy(self._h)
                self._h = None
        except Exception:
            pass


class LightGBMRanker:
    def __init__(self, num_leaves: int = 63, n_estimators: int = 500, learning_rate: float = 0.05,
                 eval_at: List[int] = None):
        self.num_leaves = num_leaves
        self.n_estimators = n_estimators
        self.learning_rate = learning_rate
        self.eval_at = eval_at or [5, 10, 20]
        self.model: Optional[_Forest] = None
        self.feature_names: Optional[List[str]] = None
        self._trained = False
        self._text: Optional[str] = None

    # -- training (ranker.py:52-155) -----------------------------------------------------------
    def train(self, train_df: pd.DataFrame, feature_cols: List[str], label_col: str = "label",
              query_col: str = "query_id", valid_df: Optional[pd.DataFrame] = None, verbose_eval: int = 50,
              backend: Optional[str] = None, seed: int = 2, hist_dtype: str = "int20", use_missing: bool = False,
              split_order: str = "low"):
        """backend: "hip" = the library's GPU trainer, "lightgbm" = the real package (as the reference), None = lightgbm
        when importable, else hip.  Returns the evals_result dict of lgb.record_evaluation:
        {"train": {"ndcg@5": [...], ...}, "valid": {...}}.

        Fidelity switches of the hip backend towards LightGBM's defaults (parity with the real package stays unpinned --
        it is not importable here; each switch is pinned to its own branch of the NumPy restatement the tests use):
          hist_dtype  "int20": gradients quantised to 2^20 levels (LightGBM's use_quantized_grad idea; the default here);
                      "float": float-histogram fidelity -- 2^-40 fixed point, finer than the float32 rounding of a
                      gradient by 2^16, and still independent of summation order;
          use_missing True: NaN is a value of its own (missing bin, learned default direction per node, the text model
                      carries missing type NaN) instead of being read as 0.0;
          split_order "lightgbm": equal-gain thresholds resolved in FeatureHistogram::FindBestThreshold's scan order (the
                      highest threshold of a run of empty bins) instead of the lowest."""
        backend = backend or ("lightgbm" if LGB_AVAILABLE else "hip")
        if backend == "hip":
            if hist_dtype not in ("int20", "float", "int40") or split_order not in ("low", "lightgbm"):
                raise ValueError("hist_dtype in {'int20', 'float'}, split_order in {'low', 'lightgbm'}")
            return self._train_hip(train_df, feature_cols, label_col, query_col, valid_df, verbose_eval, seed,
                                   40 if hist_dtype in ("float", "int40") else 20, bool(use_missing),
                                   1 if split_order == "lightgbm" else 0)
        if not LGB_AVAILABLE:
            raise ImportError("lightgbm is required for training. Install with: pip install lightgbm "
                              "(load()/predict() do not need it)")
        import tempfile
        self.feature_names = feature_cols
        X = train_df[feature_cols].values.astype(np.float32)
        y = train_df[label_col].values.astype(np.float32)
        groups = train_df.groupby(query_col, sort=False).size().values
        dtrain = lgb.Dataset(X, label=y, group=groups, feature_name=feature_cols, free_raw_data=False)
        valid_sets, valid_names = [dtrain], ["train"]
        if valid_df is not None:
            Xv = valid_df[feature_cols].values.astype(np.float32)
            yv = valid_df[label_col].values.astype(np.float32)
            gv = valid_df.groupby(query_col, sort=False).size().values
            valid_sets.append(lgb.Dataset(Xv, label=yv, group=gv, feature_name=feature_cols, reference=dtrain,
                                          free_raw_data=False))
            valid_names.append("valid")
        params = {"objective": "lambdarank", "metric": "ndcg", "eval_at": self.eval_at, "num_leaves": self.num_leaves,
                  "learning_rate": self.learning_rate, "min_child_samples": 20, "subsample": 0.8,
                  "colsample_bytree": 0.8, "reg_alpha": 0.1, "reg_lambda": 0.1, "label_gain": [0, 1, 3, 7, 15],
                  "verbose": -1, "n_jobs": -1}
        evals_result: Dict = {}
        callbacks = [lgb.log_evaluation(period=verbose_eval), lgb.record_evaluation(evals_result)]
        if valid_df is not None:
            callbacks.append(lgb.early_stopping(stopping_rounds=30, verbose=True))
        booster = lgb.train(params, dtrain, num_boost_round=self.n_estimators, valid_sets=valid_sets,
                            valid_names=valid_names, callbacks=callbacks)
        with tempfile.TemporaryDirectory() as td:  # hand the trained forest to the HIP predictor
            p = str(Path(td) / "m.lgbm")
            booster.save_model(p)
            self._text = Path(p).read_text()
        self._load_text(self._text)
        self.model.best_iteration = booster.best_iteration
        self._trained = True
        return evals_result

    def _train_hip(self, train_df, feature_cols, label_col, query_col, valid_df, verbose_eval, seed, hist_bits=20,
                   use_missing=False, split_order=0):
        lib = L.lib()
        self.feature_names = list(feature_cols)
        dev = L.device()

        def pack(df):
            # (copy=False: a float32 frame -- what feature_engineering.py writes -- is not copied a second time)
            X = torch.from_numpy(np.ascontiguousarray(df[feature_cols].to_numpy(dtype=np.float32, copy=False))).to(dev)
            y = torch.from_numpy(np.ascontiguousarray(df[label_col].to_numpy(dtype=np.float32, copy=False))).to(dev)
            g = np.ascontiguousarray(df.groupby(query_col, sort=False).size().values.astype(np.int32))
            return X, y, g

        X, y, g = pack(train_df)
        Xv = yv = gv = None
        if valid_df is not None:
            Xv, yv, gv = pack(valid_df)
        prm = L.LambdamartParams()
        prm.num_leaves, prm.n_estimators, prm.learning_rate = self.num_leaves, self.n_estimators, self.learning_rate
        prm.min_child_samples, prm.max_bin, prm.truncation_level, prm.early_stopping_rounds = 20, 255, 30, 30
        prm.lambdarank_norm, prm.bin_sample = 1, 200000
        prm.reg_alpha, prm.reg_lambda, prm.feature_fraction, prm.min_sum_hessian, prm.sigmoid = 0.1, 0.1, 0.8, 1e-3, 1.0
        prm.seed = seed
        prm.hist_bits, prm.use_missing, prm.split_order = int(hist_bits), 1 if use_missing else 0, int(split_order)
        prm.n_eval_at = len(self.eval_at)
        for i, k in enumerate(self.eval_at):
            prm.eval_at[i] = int(k)
        gains = [0, 1, 3, 7, 15]                                                     # ranker.py:118
        prm.n_label_gain = len(gains)
        for i, v in enumerate(gains):
            prm.label_gain[i] = float(v)
        nk = len(self.eval_at)
        hist = np.full((self.n_estimators, 2, nk), np.nan, dtype=np.float64)
        text_p, best_it, rounds = C.c_void_p(), C.c_int(0), C.c_int(0)
        logger.info("Training LambdaMART ranker: %d samples, %d features, %d queries", len(train_df), len(feature_cols), len(g))
        L.check(lib.rihip_lambdamart_train(X.data_ptr(), y.data_ptr(), g.ctypes.data, X.shape[0], X.shape[1], len(g),
                                           None if Xv is None else Xv.data_ptr(), None if yv is None else yv.data_ptr(),
                                           None if gv is None else gv.ctypes.data, 0 if Xv is None else Xv.shape[0],
                                           0 if gv is None else len(gv), C.byref(prm), " ".join(feature_cols).encode(),
                                           C.byref(text_p), C.byref(best_it), C.byref(rounds), hist.ctypes.data,
                                           L.stream_ptr()), "lambdamart_train")
        try:
            self._text = C.string_at(text_p.value).decode()
        finally:
            lib.rihip_free(text_p)
        self._load_text(self._text)
        self.model.best_iteration = int(best_it.value)
        self._trained = True
        r = int(rounds.value)
        res: Dict = {"train": {f"ndcg@{k}": hist[:r, 0, t].tolist() for t, k in enumerate(self.eval_at)}}
        if valid_df is not None:
            res["valid"] = {f"ndcg@{k}": hist[:r, 1, t].tolist() for t, k in enumerate(self.eval_at)}
        logger.info("Training complete. Best iteration: %d", self.model.best_iteration)
        return res

    def _load_text(self, text: str) -> None:
        h = C.c_void_p()
        b = text.encode()
        L.check(L.lib().rihip_gbdt_create_from_text(b, len(b), C.byref(h)), "gbdt_create_from_text")
        self.model = _Forest(h.value)

    # -- inference (ranker.py:161-174) --------------------------------------------------------
    def predict(self, features_df: pd.DataFrame) -> np.ndarray:
        if not self._trained or self.model is None:
            raise RuntimeError("Model not trained. Call train() first.")
        X = features_df[self.feature_names].values.astype(np.float32)
        return self.model.predict(X)

    def predict_device(self, X: torch.Tensor) -> torch.Tensor:
        """Device-resident scoring (not in the reference): X f32 [n, n_features] in feature_names order."""
        if not self._trained or self.model is None:
            raise RuntimeError("Model not trained. Call train() first.")
        return self.model.predict_device(X)

    # -- analysis (ranker.py:180-197) ---------------------------------------------------------
    def feature_importance(self, importance_type: str = "gain") -> Dict[str, float]:
        if not self._trained or self.model is None:
            raise RuntimeError("Model not trained.")
        importances = self.model.feature_importance(importance_type=importance_type)
        names = self.model.feature_name()
        result = dict(zip(names, importances.tolist()))
        return dict(sorted(result.items(), key=lambda x: x[1], reverse=True))

    def top_features(self, n: int = 10, importance_type: str = "gain") -> Dict[str, float]:
        imp = self.feature_importance(importance_type)
        return dict(list(imp.items())[:n])

    # -- persistence (ranker.py:203-226) ------------------------------------------------------
    def save(self, path: str) -> None:
        if not self._trained or self.model is None:
            raise RuntimeError("Model not trained.")
        save_path = Path(path)
        save_path.parent.mkdir(parents=True, exist_ok=True)
        save_path.write_text(self._text)
        logger.info("Saved ranker to %s", save_path)

    @classmethod
    def load(cls, path: str) -> "LightGBMRanker":
        load_path = Path(path)
        if not load_path.exists():
            raise FileNotFoundError(f"Ranker model not found at {load_path}")
        obj = cls()
        obj._text = load_path.read_text()
        obj._load_text(obj._text)
        obj.model.path = str(load_path)
        obj.feature_names = obj.model.feature_name()
        obj._trained = True
        logger.info("Loaded ranker from %s (%d features, %d trees)", load_path, len(obj.feature_names),
                    obj.model.num_trees())
        return obj

    @property
    def n_features(self) -> int:
        return len(self.feature_names) if self.feature_names else 0

    @property
    def best_iteration(self) -> int:
        if self.model is not None:
            return self.model.best_iteration
        return 0

    def model_info(self) -> Dict:
        if not self._trained:
            return {"status": "not trained"}
        return {
            "n_features": self.n_features,
            "best_iteration": self.best_iteration,
            "num_leaves": self.num_leaves,
            "learning_rate": self.learning_rate,
            "eval_at": self.eval_at,
            "top_10_features": self.top_features(10) if self._trained else {},
        }
