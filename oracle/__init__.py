"""CPU oracle for the helping-hands hot path.  TEST INFRASTRUCTURE ONLY.

A plain fp32 PyTorch/NumPy/C restatement of the reference algorithm, each function citing the
reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it; the product package (helping_hand_for_egocentric_videos_amd) never does.

Pinning status: the reference ships no tests or golden vectors for this path,
so the oracle is pinned against the reference ITSELF, imported in the development container:
  * tests/test_oracle_vs_reference.py  (runs where /root/reference exists) proves every oracle
    function equal to the imported reference module on seeded inputs, incl. full-width models;
  * tests/golden/*.npz were emitted by tests/golden/make_golden.py from the imported reference and
    are checked against the oracle everywhere (tests/test_oracle_golden.py).
Third-party arithmetic restated here: scipy.optimize.linear_sum_assignment (scipy 1.15.3 in this
image; unpinned by the reference) -- oracle/lsap.c restates its published shortest-augmenting-path
algorithm (Crouse 2016) and is checked against scipy itself on tie-heavy and random matrices.
"""
