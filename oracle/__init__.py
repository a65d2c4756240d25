"""CPU oracle for the EAV per-modality training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``eav_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may use it, and only as the checker.

Each module restates, in plain fp32 torch/numpy CPU arithmetic, the algorithm
the reference executes on this path, citing the reference file:line it follows.
Parity pin: the restatements are checked against golden vectors captured from
the *imported reference itself* (shimmed as SURVEY.md section 8c describes) by
``tests/golden/make_goldens.py``; see tests/test_oracle_*.py.
"""
