"""The bodies of tests/test_full_size_gpu.py replayed on the CPU at reduced width (ARCH = "small") with the operator
emulation running in the storage dtype: checks their plumbing (shared fp16/bf16-representable weights, oracle caching,
bank hand-over, CFG case, schedule indices) without GPU minutes.  It does not exercise the kernels and says nothing
about the full-width numbers -- those are the -m gpu run's."""
import pytest
import torch


@pytest.fixture()
def Fz(monkeypatch):
    import emu_ops
    import test_full_size_gpu as Fz
    emu_ops.install(monkeypatch)
    monkeypatch.setattr(Fz, "DEV", "cpu")
    monkeypatch.setattr(Fz, "ARCH", "small")
    yield Fz
    Fz._CACHE.clear()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_full_size_bodies_replay(Fz, dtype):
    rep = []
    Fz.test_full_referencenet_banks(dtype, rep)
    for case in Fz.CASES:
        Fz.test_full_unet3d_forward(dtype, case, rep)
    Fz.test_full_vae(dtype, rep)
    Fz.test_full_pipeline_config0_geometry(dtype, rep)
    for name in Fz.TRAJ:                       # the round-4 trajectory tests (eager on the emulation: no graphs on the CPU)
        Fz.test_full_pipeline_trajectory(dtype, name, "latency", rep)
    Fz.test_full_pipeline_trajectory(dtype, "pipeline40cfg", "latency+cfg_split", rep)     # round 5: the two-halves form of the CFG trajectory
    if dtype == torch.bfloat16:                # round 5: the in-flight identity test's "alone" half (inputs, routing attribute, scratch scope)
        Fz.test_clips_in_flight_identity_at_the_benchmarked_configuration(rep)
    Fz.test_call_batch_trajectory(dtype, rep)        # round 6: four clips per evaluation, the stored-trajectory clip in batch position 2
    Fz.test_zz_release_cache(rep)
    assert len(rep) == 1 + len(Fz.CASES) + 2 + 2 + 3 * len(Fz.TRAJ) + 3 + 2 and all(r["arch"] == "small" for r in rep)


def test_round_both_is_exact_in_both_types():
    from oracle import harness as Hn
    g = torch.Generator().manual_seed(0)
    v = torch.cat([torch.randn(4096, generator=g) * s for s in (1e-6, 1e-4, 1e-2, 1.0, 100.0)])
    r = Hn.round_both(v)
    assert torch.equal(r.to(torch.float16).float(), r) and torch.equal(r.to(torch.bfloat16).float(), r)
