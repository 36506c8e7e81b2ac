"""CPU oracle for the vision-representation scoring path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the reported CPU baseline.  The
product package (``law_of_vision_representation_in_mllms_amd``) never imports
this module and fails loudly when its HIP library is missing.

Every function restates, in plain torch-CPU / numpy fp32, the arithmetic of a
reference function and cites the reference file:line it follows.  Pinning:
the reference ships no tests or golden vectors for this path,
so the oracle is pinned against outputs of the reference code itself, imported
in the build container by ``tests/golden/make_golden.py``; the resulting
fixtures live in ``tests/golden/*.npz`` and are checked by
``tests/test_oracle_golden.py``.
"""
