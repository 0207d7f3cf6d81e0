"""CPU oracle for the Hallo denoising hot path -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and the cpu_baseline legs of bench.py / tools/w2v_bench.py may import this package.
hallo_amd/ never does: the product path runs exclusively on the HIP kernels in hallo_amd/csrc.

  hallo_ref.py     restatement of the reference's modules and pipelines (FaceAnimatePipeline, StaticPipeline): pinned
                   bit-exact against the reference's own classes (tests/test_oracle_vs_reference.py)
  _standin/        the third-party import surface (diffusers 0.27.2, xformers) the reference's modules need
  driver_ref.py    sliding-window driver, audio windows, uint8 conversion: golden vectors cut from the reference
  wav2vec_ref.py   wav2vec2 audio front-end: pinned against the reference's own Wav2VecModel class + golden vectors
  ops_ref.py       fp32 expressions of the individual operators for the kernel-level parity tests
  harness.py       builders that give oracle and native models identical synthetic weights
"""
