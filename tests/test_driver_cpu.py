"""Sliding-window driver: oracle restatement vs the reference's own code (golden vectors cut out
of /root/reference by tests/golden/make_golden.py), and the native driver's host logic vs the oracle restatement with a
recording fake pipeline (no GPU, no kernels: what is checked is the data flow -- motion-frame carry, audio windows,
mask tiling, the shared generator stream, concatenation and trim)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "driver_golden.npz")


def test_oracle_process_audio_emb_matches_reference_golden():
    from oracle import driver_ref as D
    g = np.load(GOLD)
    for T in (1, 2, 5, 37):
        out = D.process_audio_emb(torch.from_numpy(g[f"audio_in_{T}"]))
        assert out.shape == (T, 5, 3, 4)
        assert np.array_equal(out.numpy(), g[f"audio_out_{T}"]), T       # pure indexing: bit-exact


def test_native_process_audio_emb_matches_golden():
    from hallo_amd.animate import video as V
    g = np.load(GOLD)
    for T in (1, 2, 5, 37):
        out = V.process_audio_emb(torch.from_numpy(g[f"audio_in_{T}"]))
        assert np.array_equal(out.numpy(), g[f"audio_out_{T}"]), T


def test_oracle_frames_to_uint8_matches_reference_golden():
    from oracle import driver_ref as D
    g = np.load(GOLD)
    u8 = D.frames_to_uint8(torch.from_numpy(g["video_in"]))
    assert u8.dtype == np.uint8 and np.array_equal(u8, g["video_u8"])
    # known answers: 0 -> 0, 1 -> 255, just below 1 -> 254 (truncation, not rounding), below 0 / above 1 clamp
    assert tuple(g["video_u8"][0, 0, :4, 0]) == (0, 255, 254, 254)
    assert g["video_u8"].min() == 0 and g["video_u8"].max() == 255


class _FakeOut:
    def __init__(self, v):
        self.videos = v


class _FakePipe:
    """Deterministic stand-in for FaceAnimatePipeline.__call__: records its inputs, draws the 'latents' from the shared
    generator exactly like prepare_latents does, and returns frames that depend on every input."""

    def __init__(self):
        self.calls = []

    def __call__(self, **kw):
        kw.pop("output_type", None)
        self.calls.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items() if k != "generator"})
        Fr, H, W = kw["video_length"], kw["height"], kw["width"]
        noise = torch.randn((1, 3, Fr, H, W), generator=kw["generator"])
        ref = kw["ref_image"][0]                                           # (1 + n_motion, 3, H, W)
        base = ref.mean(dim=0)[None, :, None] * 0.25 + 0.5                 # the motion frames influence the output
        aud = kw["audio_tensor"].float().mean(dim=(2, 3)).view(1, 1, Fr, 1, 1)
        return _FakeOut((base + 0.05 * noise + 0.01 * aud).clamp(0, 1))


def _inputs(T=50, S=16):
    g = torch.Generator().manual_seed(7)
    src = torch.rand((3, S, S), generator=g) * 2 - 1
    region = (torch.rand((3, S, S), generator=g) > 0.5).float()
    emb = torch.randn((512,), generator=g)
    masks = lambda: [torch.rand((1, (S // 2 ** l) ** 2), generator=g) for l in range(4)]
    audio = torch.randn((T, 12, 8), generator=g)
    return src, region, emb, masks(), masks(), masks(), audio


@pytest.mark.parametrize("T,audio_length", [(50, 47), (16, 16), (33, 30)])
def test_native_driver_data_flow_equals_oracle(T, audio_length):
    from hallo_amd.animate import video as V
    from oracle import driver_ref as D
    src, region, emb, fm, cm, lm, audio = _inputs(T)
    audioproj = lambda a: a.flatten(2)[:, :, None, :].repeat(1, 1, 2, 1)   # (1, F, 5*12*8) -> (1, F, 2, 480): any fixed map
    p_o, p_n = _FakePipe(), _FakePipe()
    kw = dict(clip_length=16, n_motion_frames=2, img_size=(16, 16), inference_steps=3, cfg_scale=3.5,
              motion_scale=[1.0, 0.5, 1.2], audio_length=audio_length)
    vo = D.generate_video(p_o, audioproj, src, region, emb, fm, cm, lm, audio, seed=42, **kw)
    vn = V.generate_video(p_n, audioproj, src, region, emb, fm, cm, lm, audio, seed=42, **kw)
    times = T // 16
    assert len(p_o.calls) == len(p_n.calls) == times
    for co, cn in zip(p_o.calls, p_n.calls):
        assert co.keys() == cn.keys()
        for k in co:
            a, b = co[k], cn[k]
            if torch.is_tensor(a):
                assert torch.equal(a, b), k
            elif isinstance(a, list) and a and torch.is_tensor(a[0]):
                assert all(torch.equal(x, y) for x, y in zip(a, b)), k
            else:
                assert a == b, k
    assert vn.shape == vo.shape == (3, min(audio_length, times * 16), 16, 16)
    assert torch.equal(vn, vo)
    # known answers of the carry: clip 0 sees the source image three times; clip t > 0 sees the last two frames of
    # clip t-1 mapped back to [-1, 1]
    c0 = p_n.calls[0]["ref_image"][0]
    assert torch.equal(c0[1], c0[0]) and torch.equal(c0[2], c0[0])
    if times > 1:
        prev_last = vo[:, 14:16].permute(1, 0, 2, 3) * 2 - 1
        assert torch.equal(p_n.calls[1]["ref_image"][0][1:], prev_last)
        assert torch.equal(p_n.calls[1]["ref_image"][0][0], src)


def test_uint8_output_needs_the_gpu_kernel():
    from hallo_amd.animate import video as V
    from hallo_amd.lib import HalloLibraryError
    src, region, emb, fm, cm, lm, audio = _inputs(16)
    with pytest.raises(HalloLibraryError):
        V.generate_video(_FakePipe(), lambda a: a.flatten(2)[:, :, None, :], src, region, emb, fm, cm, lm, audio,
                         clip_length=16, img_size=(16, 16), inference_steps=1, output="uint8")


def test_tensor_to_video_hands_the_reference_bytes_to_the_encoder(tmp_path):
    """SURVEY 8f row 3 (hallo/utils/util.py:297-322): the uint8 frames go to an ffmpeg subprocess as raw RGB24 with the
    reference's container settings (fps, libx264 / aac, audio cut to the video's duration).  A stand-in `ffmpeg` script records
    its arguments and stdin: the bytes must be exactly the reference's `np.clip(x * 255, 0, 255).astype(np.uint8)` frames."""
    import json
    import os
    import stat
    import numpy as np
    import pytest
    import torch
    from hallo_amd.animate import video as V
    fake = tmp_path / "ffmpeg"
    fake.write_text("#!/bin/sh\nprintf '%s\\n' \"$@\" > \"$0.args\"\ncat > \"$0.stdin\"\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    g = torch.Generator().manual_seed(5)
    vid = torch.rand((3, 7, 16, 24), generator=g) * 1.2 - 0.1                     # the reference's (c, f, h, w) tensor, slightly out of range
    out = tmp_path / "out.mp4"
    V.tensor_to_video(vid, out, audio_source="speech.wav", fps=25, ffmpeg=str(fake))
    want = np.clip(vid.permute(1, 2, 3, 0).numpy() * 255, 0, 255).astype(np.uint8)   # util.py:308-312
    got = np.frombuffer(open(str(fake) + ".stdin", "rb").read(), dtype=np.uint8).reshape(want.shape)
    assert np.array_equal(got, want)
    args = open(str(fake) + ".args").read().split("\n")
    for token in ("rawvideo", "24x16", "rgb24", "25", "speech.wav", "aac", "libx264", "yuv420p", str(out)):
        assert token in args, token
    assert args[args.index("-t") + 1] == f"{7 / 25:.6f}"                            # AudioFileClip(...).subclip(0, frames / fps)
    # uint8 frames (what generate_video(output="uint8") returns) pass through untouched; no audio -> -an
    V.tensor_to_video(torch.from_numpy(want.copy()), out, fps=30, ffmpeg=str(fake))
    assert np.array_equal(np.frombuffer(open(str(fake) + ".stdin", "rb").read(), dtype=np.uint8).reshape(want.shape), want)
    assert "-an" in open(str(fake) + ".args").read().split("\n")
    # a failing encoder and a missing one are loud
    bad = tmp_path / "ffmpeg_bad"
    bad.write_text("#!/bin/sh\ncat > /dev/null\necho boom >&2\nexit 3\n")
    bad.chmod(bad.stat().st_mode | stat.S_IEXEC)
    with pytest.raises(RuntimeError, match="boom"):
        V.tensor_to_video(vid, out, ffmpeg=str(bad))
    old = os.environ.get("PATH", "")
    os.environ["PATH"] = str(tmp_path / "nowhere")
    try:
        with pytest.raises(RuntimeError, match="no ffmpeg"):
            V.tensor_to_video(vid, out)
    finally:
        os.environ["PATH"] = old
    meta = V.write_raw_rgb24(torch.from_numpy(want.copy()), tmp_path / "frames.rgb", fps=25)
    assert meta == {"width": 24, "height": 16, "fps": 25, "frames": 7, "pix_fmt": "rgb24"}
    assert json.load(open(str(tmp_path / "frames.rgb") + ".json")) == meta
    assert os.path.getsize(tmp_path / "frames.rgb") == want.size
