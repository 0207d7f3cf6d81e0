"""CPU oracle for the Hallo denoising hot path -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
hallo_amd/ never does: the product path runs exclusively on the HIP kernels in hallo_amd/csrc.
"""
