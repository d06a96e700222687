"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's hot path, used as the parity checker by `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`. Nothing under
`clip_assisted_data_labeling_amd/` may import this package: the product path is the HIP library
and it fails loudly when that library is missing.
"""
