"""TEST INFRASTRUCTURE ONLY.

CPU restatement (torch-CPU, fp32) of the JPD-SE training hot path of the
reference (`ctu.models` / `ctu.trainers`).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this package; the product (`jpd-se_amd/`) never does and fails loudly when its
HIP extension is missing.

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md §4, §8c), so the restatement is pinned against outputs of the
reference itself, imported in the build container by
`oracle/check_against_reference.py` / `oracle/make_golden.py` (fixtures
committed under `tests/golden/`).
"""
