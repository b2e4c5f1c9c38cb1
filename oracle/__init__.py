"""CPU oracle of the seq2seq-vc hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under seq2seq_vc_amd/ imports this package.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may use it, and only as the checker / the timed CPU baseline.
Each function cites the reference file:line it restates (paths relative to the reference root
unilight/seq2seq-vc @ 2024_08_07).  Pinning: tests/golden/*.npz are outputs of the imported
reference itself (generator: tools/gen_golden.py); tests/test_oracle_golden.py checks every oracle
function against them.
"""
