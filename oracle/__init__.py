"""CPU oracle for the MultINN LSTM-NADE / LSTM-RBM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is shipped or measured as the
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  ``multinn_amd`` never imports ``oracle``.

PARITY UNPINNED: the reference (ilya16/MultINN) holds no tests, golden vectors or
fixtures for this path and its TF 1.13.1 / TFP 0.6.0 dependencies cannot be
installed here, so this oracle is a restatement of the
reference files plus the TF op semantics recorded in ``oracle/tf_semantics.py``.
It is pinned by analytical known-answer tests (tests/test_oracle_kats.py) and by
agreement between two independent restatements (NumPy float64 loops here,
torch-CPU float32 autograd in ``oracle/torch_ref.py``).

Exception -- PINNED: ``oracle/musical.py`` (the musical sample metrics) is checked against outputs of
the reference's own NumPy-only module, captured in this container as ``tests/golden/musical_metrics.npz`` by
``tests/golden/make_musical_golden.py``.
"""
