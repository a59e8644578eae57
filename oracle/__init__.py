"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the MoSh++ Stage-II hot path.  Nothing in the
product package (`moshpp_amd/`) may import from here; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do.

PARITY UNPINNED: the reference's arithmetic lives in chumpy / psbody.smpl,
neither of which is present in /root/reference nor installable here, and the
reference ships no tests or golden vectors.
"""
