"""CPU oracle for the compression inference path -- TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package.
The product package `autoencoder_based_image_compression_amd` never does.
"""
