"""CPU oracle for the IIC training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``iic_amd/`` imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may use it, and there only as the checker / the timed CPU baseline.

Parity pinning: the reference (xu-ji/IIC) ships no tests or golden vectors
(SURVEY.md §4), so the oracle is pinned against *outputs of the reference itself
run in the build container*: ``oracle/gen_golden.py`` imports the reference's own
``IID_losses.py`` / ``archs`` read-only from /root/reference, evaluates them on
seeded inputs and commits the results under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks every oracle function against those.
"""
