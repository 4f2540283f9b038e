"""CPU oracle for the recommendit hot path -- TEST INFRASTRUCTURE ONLY.

A plain NumPy restatement of the reference's algorithms (file:line cited per
function).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import anything from here, and only
as the checker.  The product path (``recommendit_amd``) never imports this
package and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * two_tower_np  -- pinned by golden vectors generated from the importable
    reference (``oracle/make_golden.py`` -> ``tests/golden/*.npz``).
  * metrics_np    -- pinned by the reference's known-answer tests
    (tests/test_models.py:372-426).
  * retrieval_np / gbdt_np -- faiss / lightgbm are third-party, absent here:
    **parity unpinned**; restated from the wrapper semantics
    (src/models/faiss_index.py, src/models/ranker.py) and property-tested.
"""
