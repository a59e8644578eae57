"""TEST INFRASTRUCTURE ONLY.

CPU restatements ("oracles") of the MoSh++ hot path: `stageii_oracle` (the per-frame Stage-II chain) and `stagei_oracle` (the
joint Stage-I solve).  Nothing in the product package (`moshpp_amd/`) may import from here; only `tests/`,
`__graft_entry__.smoke()` and the CPU-baseline legs of `bench.py` do.

PARITY PARTLY PINNED.  The reference's core arithmetic lives in chumpy / psbody.smpl, neither of which is present in
/root/reference nor installable here, and the reference ships no tests or golden vectors (SURVEY.md sections 4, 8c).  What CAN be
executed of the reference is used to pin the restatements (tests/test_ref_golden.py, fixtures + generating scripts under
tests/golden/): its pure-Python classes and functions run with stand-in modules, and its C++ distance header compiled in place into
oracle/_ref/ (recipe: oracle/ref_build/).  Each oracle's header lists what is pinned and what is not.
"""
