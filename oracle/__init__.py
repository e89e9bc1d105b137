"""CPU oracle for the JIRF network_g inference forward passes.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the HIP library, the
drop-in ``basicsr.models.archs`` classes) may import this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker.

PARITY UNPINNED: the reference ships no tests, golden vectors or checkpoints
for these forwards, and its arithmetic lives in Jittor 1.3.10.0 (pinned in
``requirements.txt:16`` of the reference), which is neither vendored nor
installed here.  Every function below is a restatement, in torch-CPU fp32 (plus
float64 where a discrete decision is taken), of the cited lines of the
reference's ``basicsr/models/archs/*.py``.
"""
