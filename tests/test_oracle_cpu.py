"""CPU tests of the oracle side of the parity harness (no GPU, no /root/reference needed):
  * sensitivity: each conditioning sub-path (audio tokens, per-branch motion_scale, motion-frame features,
    face-locator feature, reference bank) moves the oracle's UNet output by several times the parity
    tolerance, so the GPU parity tests CAN fail on a bug in that sub-path;
  * schedule known answers: trailing timesteps, prev_t, alphas_cumprod values -- for both the
    oracle's scheduler and hallo_amd.scheduler.DDIMScheduler, which must agree bit-for-bit.
"""
import numpy as np
import pytest
import torch

from oracle import hallo_ref as H
from oracle import harness as Hn


@pytest.fixture(scope="module")
def nets():
    return Hn.oracle_nets(dtype=torch.float16)


def test_subpaths_are_numerically_visible(nets):
    o = nets
    B, Fr, h = 1, 4, 16
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.randn(s, generator=g)
    enc = r(B, 4, 64)
    with torch.no_grad():
        banks = [b.to(torch.float16) for b in o["reference_unet"](r(3, 4, h, h), torch.tensor(0), enc)]
    lat, audio, fm = r(B, 4, Fr, h, h), r(B, Fr, 32, Hn.SMALL_AUDIO_DIM), r(B, 80, Fr, h, h)
    masks = lambda: [torch.rand((B * Fr, (h // 2 ** l) ** 2), generator=g) for l in range(4)]
    full, face, lip = masks(), masks(), masks()
    run = lambda a, ms, bk, f: o["denoising_unet"](lat, torch.tensor(500), enc, bk, audio_embedding=a, mask_cond_fea=f,
                                                   full_mask=full, face_mask=face, lip_mask=lip, motion_scale=ms)
    thr = 3e-2   # 3x the fp16 UNet tolerance
    with torch.no_grad():
        base = run(audio, [1.0, 1.0, 1.0], banks, fm)
        sens = {"audio": Hn.rel_l2(run(audio * 0, [1.0, 1.0, 1.0], banks, fm), base),
                "motion_scale_face": Hn.rel_l2(run(audio, [1.0, 0.0, 1.0], banks, fm), base),
                "face_locator": Hn.rel_l2(run(audio, [1.0, 1.0, 1.0], banks, fm * 0), base)}
        b2 = [b.clone() for b in banks]
        for b in b2:
            b.view(B, 3, *b.shape[1:])[:, 1:] *= 0
        sens["motion_frames"] = Hn.rel_l2(run(audio, [1.0, 1.0, 1.0], b2, fm), base)
        b3 = [b.clone() for b in banks]
        for b in b3:
            b.view(B, 3, *b.shape[1:])[:, 0] *= 0
        sens["reference_bank"] = Hn.rel_l2(run(audio, [1.0, 1.0, 1.0], b3, fm), base)
    print(sens)
    for k, v in sens.items():
        assert v > thr, (k, v)


@pytest.mark.parametrize("n,first,last", [(10, 999, 99), (25, 999, 39), (40, 999, 24)])
def test_schedule_known_answers(n, first, last):
    from hallo_amd.scheduler import DDIMScheduler
    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
              prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    so, sn = H.make_scheduler(), DDIMScheduler(**kw)
    so.set_timesteps(n)
    sn.set_timesteps(n)
    assert torch.equal(so.timesteps, sn.timesteps)
    ts = sn.timesteps.tolist()
    assert ts[0] == first and ts[-1] == last and len(ts) == n
    assert ts == [int(x) for x in (np.round(np.arange(1000, 0, -1000 / n)) - 1)]
    assert torch.equal(so.alphas_cumprod, sn.alphas_cumprod)           # bit-exact fp32 table
    ac = sn.alphas_cumprod
    assert float(ac[999]) == 0.0                                       # zero terminal SNR
    assert abs(float(ac[0]) - 0.9991499782) < 1e-7 and abs(float(ac[500]) - 0.1415126324) < 1e-7
    for t in ts:
        tt, pt = sn.step_indices(t)
        assert pt == tt - 1000 // n
        a_t, a_p = sn.step_alphas(t)
        assert a_t == float(so.alphas_cumprod[tt])
        assert a_p == (float(so.alphas_cumprod[pt]) if pt >= 0 else 1.0)
    assert sn.step_indices(ts[-1])[1] < 0                              # last step lands on alpha_prev = 1


def test_ddim_step_formula_matches_oracle_scheduler():
    """x_prev of the fused kernel's formula (evaluated in fp64 here) == diffusers-style scheduler.step."""
    so = H.make_scheduler()
    so.set_timesteps(25)
    from hallo_amd.scheduler import DDIMScheduler
    sn = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                       prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    sn.set_timesteps(25)
    g = torch.Generator().manual_seed(0)
    x, v = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    for t in so.timesteps[[0, 7, 24]]:
        ref = so.step(v, t, x, eta=0.0, return_dict=False)[0]
        a_t, a_p = sn.step_alphas(t)
        sa, sb, pa, pb = a_t ** 0.5, (1 - a_t) ** 0.5, a_p ** 0.5, (1 - a_p) ** 0.5
        xd, vd = x.double(), v.double()
        got = pa * (sa * xd - sb * vd) + pb * (sa * vd + sb * xd)
        assert torch.allclose(got.float(), ref, atol=2e-6, rtol=1e-5)
