"""CPU oracle for the GraphSAGE sample -> gather -> mean -> linear -> act path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker.  The product path
(``graphsage-simple_amd/``) never imports this package and raises when its HIP
library is missing.

Contents
--------
ref_dense   -- reference-faithful restatement (Python set sampling, dense
               [B,U] mask, div, mm) of graphsage/aggregators.py:34-76 and
               graphsage/encoders.py:40-62.  Pinned against golden vectors
               generated from the imported reference (tests/golden/).
ref_sparse  -- same arithmetic on padded neighbour lists without the dense
               mask (fp64 accumulation), for BASELINE-size batches where the
               dense mask would be 10 GB.  Pinned against ref_dense.
sampler_ref -- bit-exact C restatement of the device sampler's integer work
               (Philox4x32-10 + Floyd subset selection); there is no reference
               counterpart for its random stream (the reference consumes
               Python's global ``random``), so its *distributional* contract
               (aggregators.py:42-46) is what the tests pin.
"""
