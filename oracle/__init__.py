"""CPU oracle for the captioning hot path -- TEST INFRASTRUCTURE ONLY.

A NumPy restatement of the reference graph (`/root/reference/ImageCaptioning/model/*`,
`train.py`'s optimizer wiring), written from the reference's Python sources and the
PaddlePaddle-1.8 op semantics they call.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this package; the product package
`myimagecaptioningmodel_amd` never does.

PARITY UNPINNED: the reference holds no tests, golden vectors or recorded losses, and
PaddlePaddle 1.8 (which owns every FLOP of the reference path) is not installable here, so
this oracle is pinned only by (1) agreement with an independent `torch.autograd` build of
the same graph (tests/test_oracle_vs_torch.py), (2) analytic known-answer tests and
(3) finite differences.  Paddle op semantics taken from memory are flagged "unverified
against Paddle" where they are used.
"""
