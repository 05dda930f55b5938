"""CPU oracle for the X-as-Supervision training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / the timed CPU
baseline.  The shipped path (``x-as-supervision_amd/``) never imports this
package and raises when its HIP library is missing.

Every function is an independent restatement (stock PyTorch CPU fp32 ops) of
the reference algorithm and cites the reference file:line it follows
(paths relative to the upstream repository root).  The restatement is pinned
by golden vectors produced by importing the reference's own Python modules
(``tests/golden/make_golden.py``); see ``tests/test_oracle_golden.py``.

Pinning status per piece (details in DESIGN.md):
  * head, geometry, mask renderer, losses, physique net, SMPL LBS, model
    step wiring: pinned by reference-import goldens.
  * ResNet ``Bottleneck`` block: lives in torchvision 0.17.2 (absent here,
    un-vendored) -> restated from its published definition; PARITY UNPINNED.
  * GCN discriminator (SAGEConv / graph LayerNorm): lives in
    torch_geometric 2.5.3 (absent here) -> restated from its published
    definition; PARITY UNPINNED (only ``my_batched_dense_to_sparse`` has a
    known-answer vector, modules/gcn.py:112-116).
"""
