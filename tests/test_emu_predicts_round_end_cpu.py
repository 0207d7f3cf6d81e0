"""The bodies of tests/test_zz_round_end_gpu.py (written after the round's GPU budget was spent) replayed on the CPU with
the operator emulation running in the STORAGE dtype (fp16 / bf16: fp32 arithmetic inside an operator, one rounding at its
output, as the kernels do).  On cases that have been on hardware the emulation in half precision lands close to the
measured GPU errors (clip pipeline: 1.5e-3 / 65 dB predicted vs 1.9e-3 / 63 dB measured in fp16, 1.2e-2 / 46.5 dB vs
1.6e-2 / 44.8 dB in bf16), so a pass here says the tolerances of the not-yet-run GPU tests are realistic and their plumbing
(devices, dtypes, shapes, callbacks) is right.  It does not exercise the kernels."""
import pytest
import torch


@pytest.fixture()
def Z(monkeypatch):
    import emu_ops
    import test_zz_round_end_gpu as Z
    emu_ops.install(monkeypatch)
    monkeypatch.setattr(Z, "DEV", "cpu")
    return Z


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_static_pipeline_prediction(Z, dtype):
    rep = []
    Z.test_static_pipeline(dtype, 3.5, rep)
    lat = [r for r in rep if "latents" in r["test"]][0]["rel_l2"]
    psnr = [r for r in rep if "psnr" in r["test"]][0]["psnr_db"]
    assert lat < (5e-3 if dtype == torch.float16 else 3e-2) and psnr > (55.0 if dtype == torch.float16 else 40.0)


@pytest.mark.slow
def test_config0_plumbing_prediction(Z):
    rep = []
    Z.test_inference_plumbing_config0(rep)
    assert [r for r in rep if r["test"] == "config0_plumbing_psnr"][0]["psnr_db"] > 50.0
