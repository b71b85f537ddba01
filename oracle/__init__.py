"""CPU restatement of the reference hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``sam6d_amd/`` imports this package.  Importers allowed by the
build contract: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` (always as the checker / baseline, never
as the thing measured or shipped).
"""
