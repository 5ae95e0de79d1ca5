"""CPU oracle for the Jukebox sampling path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-numpy restatement of the reference's algorithm for the hot path
(SURVEY.md section 8a): transformer decode/prefill with the factored attention
patterns and k/v cache, ConditionalAutoregressive2D.sample/primed_sample,
SimplePrior conditioning, Conditioner / VQ-VAE conv stacks, bottleneck
quantise/dequantise and the window plan of sample.py.  Every function cites the
reference file:line it follows (paths relative to the reference tree).

Who may import this package: `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` -- as the checker / the timed CPU baseline, never
as the thing shipped.  `jukebox_amd/` never imports it; the product path fails
loudly when the HIP library is missing.

Parity pinning: the reference ships no golden vectors for this path
(SURVEY.md section 8c), so the oracle is pinned against outputs of the reference
itself, run in the build container under `tests/golden/refshim.py` by
`tests/golden/gen_golden.py`; the resulting fixtures are committed under
`tests/golden/*.npz` and checked by `tests/test_oracle_golden.py` (CPU suite).

Arithmetic: float32 numpy.  `fp16=True` emulates the reference's half path by
rounding to IEEE half at exactly the points where the reference materialises a
half tensor (factored_attention.py:86-98, ops.py:24,99, transformer.py:170-171).
"""
