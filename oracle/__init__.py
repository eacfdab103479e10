"""CPU oracle for the ImmunoStruct hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain un-fused PyTorch (fp32 or fp64, CPU), the
arithmetic of the reference's multimodal forward/backward path so that the
HIP product path in ``immunostruct_amd`` can be checked against it.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from here.  Nothing under
``immunostruct_amd/`` imports it: the product path fails loudly when the HIP
extension is missing, it never falls back to this code.

Pinning status
--------------
* Fusion head, attention, VAE, losses, contrastive loss: PINNED.  In the build
  container the reference's own ``models/*.py``, ``utils/loss.py`` and
  ``utils/contrastive.py`` are imported unchanged from ``/root/reference``
  (``oracle/shims.py``) and their outputs are (i) asserted equal to this
  restatement and (ii) committed as golden vectors under ``tests/golden/`` by
  ``oracle/make_golden.py``.
* ``dgl.nn.EGNNConv``, ``dgl.batch`` and ``torch_geometric`` pooling are
  third-party code that is NOT vendored in the reference (DGL is unpinned,
  README.md:107,137; torch_geometric==2.5.3, README.md:143) and not
  installable here.  For those three operators the oracle follows the
  published algorithm (Satorras et al. 2021 eqs. 3-6 with DGL 2.x's concrete
  choices, see ``oracle/graph_ref.py``) and is checked by E(3)-equivariance,
  permutation and degree-0 property tests and by state-dict shape agreement
  with the reference call sites: **parity unpinned** for these operators.
"""
