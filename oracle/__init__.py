"""TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the reference hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and only as the checker.  The product package ``dsen2_amd`` never imports this package.
"""
