"""Closes the loop for tests/emu_ops.py: the BODIES of the GPU operator parity tests (tests/test_ops_gpu.py -- the same
inputs, the same fp32 oracle expressions of oracle/ops_ref.py, the same tolerances) are replayed on the CPU with the torch
emulation in place of the HIP library.  On the GPU those tests establish kernel == oracle expression; here they establish
emulation == oracle expression, so the host-logic tests that run the native models on the emulation
(tests/test_host_emulated_cpu.py) exercise the semantics the kernels were verified to have.  One small case per test."""
import pytest
import torch


@pytest.fixture()
def T(monkeypatch):
    import emu_ops
    import test_ops_gpu as T
    emu_ops.install(monkeypatch)
    monkeypatch.setattr(T, "_dev", lambda: torch.device("cpu"))
    return T


DT = [torch.float16, torch.bfloat16]


@pytest.mark.parametrize("dtype", DT)
def test_gemm_family(T, dtype):
    rep = []
    T.test_gemm_plain(dtype, 256, 320, 320, rep)
    T.test_gemm_plain(dtype, 70, 24, 2560, rep)
    T.test_gemm_asymmetric_layout(dtype, rep)
    T.test_gemm_epilogues(dtype, rep)
    T.test_gemm_fused_layernorm(dtype, 1000, 328, 640, rep)
    T.test_gemm_geglu_fused_layernorm(dtype, 300, 320, rep)
    T.test_gemm_geglu(dtype, 300, 320, rep)
    T.test_gemm_batched(dtype, rep)
    assert len(rep) >= 10


@pytest.mark.parametrize("dtype", DT)
def test_conv_family(T, dtype):
    rep = []
    cfgs = [m.args[1] for m in T.test_conv3x3.pytestmark if m.name == "parametrize" and m.args[0] == "cfg"][0]
    small = sorted(cfgs, key=lambda c: str(c))[:3]
    for c in small:
        T.test_conv3x3(dtype, c, rep)
    T.test_conv3x3_asym_pad_and_epilogue(dtype, rep)
    assert rep


@pytest.mark.parametrize("dtype", DT)
def test_attention_family(T, dtype):
    rep = []
    T.test_attention_single_segment(dtype, 40, 200, 100, rep)
    T.test_attention_single_segment(dtype, 80, 128, 64, rep)
    T.test_attention_reference_segment_cfg(dtype, 80, 100, rep)
    T.test_attention_audio_branches_one_launch(dtype, 160, 16, rep)
    T.test_attention_prescaled_q(dtype, 40, 64, 4, rep)
    T.test_face_xattn_fused(dtype, 160, 2, 128, 2, rep)
    T.test_face_xattn_fused(dtype, 320, 8, 256, 2, rep)
    T.test_temporal_attention(dtype, 320, 9, 10, 2, rep)
    assert len(rep) >= 8


@pytest.mark.parametrize("dtype", DT)
def test_norm_and_elementwise_family(T, dtype):
    rep = []
    T.test_groupnorm(dtype, 3, 256, 320, True, 1e-5, rep)
    T.test_layernorm(dtype, 333, 640, rep)
    T.test_layernorm_with_pe(dtype, rep)
    T.test_softmax_rows(dtype, rep)
    T.test_timestep_embedding(dtype, rep)
    T.test_cfg_ddim_step(dtype, False, rep)
    T.test_cfg_ddim_step(dtype, True, rep)
    assert rep


def test_layout_and_copy(T):
    T.test_layout_and_copy([])


@pytest.mark.parametrize("dtype", DT)
def test_wav2vec_kernel_family(monkeypatch, dtype):
    import emu_ops
    import test_wav2vec_gpu as Wt
    emu_ops.install(monkeypatch)
    monkeypatch.setattr(Wt, "_dev", lambda: torch.device("cpu"))
    rep = []
    Wt.test_conv0_groupnorm_gelu(dtype, 4321, 32, 10, 5, rep)
    Wt.test_lerp_rows(dtype, 7, 29, 32, rep)
    Wt.test_lerp_rows(dtype, 5, 1, 64, rep)
    for act in ("gelu", "gelu_pre"):
        Wt.test_gemm_gelu_over_overlapping_windows(dtype, act, 37, 16, 256, 16, rep)
        Wt.test_gemm_gelu_over_overlapping_windows(dtype, act, 50, 48, 6144, 48, rep)
    assert len(rep) >= 7
