"""recommendit_amd -- MI355X-native (gfx950) hot path of sarihammad/recommendit.

Same names as the reference's ``src/models/__init__.py:1-3`` so callers switch with one import:
``from recommendit_amd import TwoTowerModel, FAISSIndex, LightGBMRanker``.
"""
from .two_tower import ItemTower, TwoTowerModel, UserTower, N_GENRES  # noqa: F401
from .faiss_index import FAISSIndex  # noqa: F401
from .ranker import LightGBMRanker  # noqa: F401
from ._lib import have_gpu  # noqa: F401  (the switch a caller guards the swap with: INTEGRATION.md §1)

__all__ = ["TwoTowerModel", "UserTower", "ItemTower", "FAISSIndex", "LightGBMRanker", "N_GENRES", "have_gpu"]
