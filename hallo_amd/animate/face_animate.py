"""FaceAnimatePipeline -- one sliding-window clip of audio-driven portrait animation on one MI355X.

Reference: hallo/animate/face_animate.py:58-442 (`__call__` 249-442, `prepare_latents` 136-188,
`decode_latents` 222-246).  Constructor, `to`, `__call__` signature, callback protocol and the output
(`FaceAnimatePipelineOutput.videos`: fp32 CPU tensor (1, 3, F, H, W) in [0, 1]) are the reference's.

Execution plan (not the reference's):
  * one-off per clip: face tokens, VAE-encode(3 images), FaceLocator on ONE mask frame (the F frames
    are identical copies, face_animate.py:339-341), ReferenceNet write pass, reference/face/audio
    K/V projections (ClipCache);
  * per DDIM step: one UNet evaluation (batch 2 under CFG) on token-major activations, then ONE fused
    kernel for CFG combine + DDIM update on fp32 latents that also writes the next UNet input;
    timestep / alpha lookups are host integers, so there is no device sync inside the loop
    (the reference syncs twice per step: CPU `alphas_cumprod[t]` with a device `t`, and the CPU `uc_mask`);
  * decode: all F frames as one batch, clamp + NCHW fp32 conversion fused, one D2H copy;
  * `use_graph=True`: the UNet evaluation is captured ONCE per (geometry, batch, dtype) as a hipGraph and replayed for steps
    1.. of every clip (step 0 of a clip runs eagerly and refreshes the per-clip constants in place); the timestep lives in
    a device tensor, latents / mask feature / constants in buffers that keep their addresses from clip to clip.

Deviation from the reference, on purpose: `guidance_scale <= 1` works (the reference doubles the audio
tensor unconditionally, face_animate.py:377-379, and then dies in the audio cross-attention -- SURVEY F5);
here nothing is doubled without CFG.  BASELINE configs[1] (25 steps, no CFG) needs this.
"""
import contextlib
from dataclasses import dataclass

import torch

from .. import ops
from ..models.attention import CLIP_BATCH, SKIP_BANK, ClipCache
from ..models.mutual_self_attention import ReferenceAttentionControl
from ..models.unet_3d import pack_masks
from .image_processor import preprocess_image


@dataclass
class FaceAnimatePipelineOutput:
    videos: torch.Tensor


class StepGraph:
    """Static buffers + the captured hipGraph of one `UNet3DConditionModel.forward_tokens` call.  ~690 kernel launches per
    evaluation go through ctypes at 3-4 us of host time each; behind the sub-10-us kernels of the 8x8 / 16x16 levels the GPU
    waits for the host (3 % of a clip in launch gaps, profiles/r2_bench_launch_gaps.json).  A replay is one host call."""

    def __init__(self, B, Fr, L, C0, device, dtype, split=False, out_channels=4):
        self.x_in = torch.zeros((B * Fr, L, 8), device=device, dtype=dtype)
        self.mask_cond = torch.zeros((B * Fr, L, C0), device=device, dtype=dtype)
        self.t_dev = torch.zeros((B,), device=device, dtype=torch.float32)
        self.cache = ClipCache()
        self.graph, self.out = None, None
        self.replays = 0
        # cfg_split: the uncond / cond halves of a CFG evaluation as two B = 1 evaluations (own clip cache, own graph, own
        # timestep tensor each) that read their halves of x_in / mask_cond and write their halves of ONE output buffer
        self.halves = [StepGraph(1, 0, L, 8, device, dtype) for _ in range(2)] if split else None
        self.v_out = torch.empty((B * Fr, L, out_channels), device=device, dtype=dtype) if split else None

    def capture(self, fn):
        g = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of OTHER host threads (e.g. the RCCL watchdog of a multi-GPU run) must not invalidate the capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self.out = fn()
        self.graph = g


class FaceAnimatePipeline:
    MAX_GRAPHS = 2          # captured (geometry, batch, ...) keys kept alive at a time

    def __init__(self, vae, reference_unet, denoising_unet, face_locator, image_proj, scheduler, use_graph=False, routing=None,
                 cfg_split=False):
        """cfg_split: under classifier-free guidance run the uncond and the cond half of every UNet evaluation as two B = 1
        evaluations on two HIP streams, joined at the fused CFG + DDIM kernel (they share nothing but read-only weights and
        banks, hallo/animate/face_animate.py:397-417; the uncond half has no bank segment at all).  The clips of ONE video are
        sequential (scripts/inference.py:302-310), so this is how a second evaluation gets in flight inside one clip: the CUs
        one half leaves idle (16 x 16 / 8 x 8 levels, launch tails, dependent-launch bubbles) take the other half's kernels.
        routing: the kernel routing this pipeline's launches are enqueued (and its graphs captured) under -- "latency" (the
        library defaults: one clip at a time), "throughput" (several pipelines in flight on one GPU: ops.THROUGHPUT_OPTIONS), a
        dict of hallo_set_option values, or None = whatever the process has set.  Applied around the enqueue calls of a clip
        (ops.routing) and restored afterwards: a process may run pipelines of both kinds side by side."""
        self.vae, self.reference_unet, self.denoising_unet = vae, reference_unet, denoising_unet
        self.face_locator, self.image_proj, self.scheduler = face_locator, image_proj, scheduler
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
        self.timings = {}
        self.use_graph = use_graph
        self.routing = routing
        self.cfg_split = cfg_split
        self._graphs = {}
        self._scratch = self._scratch_aux = self._aux_stream = None
        self._scratch_decode = self._decode_stream = None

    @property
    def scratch(self):
        """Launch scratch (split-K slab, GroupNorm partial statistics) of THIS pipeline: its eager launches and its captured
        graphs use nothing else, so pipelines in flight on different streams never share a producer -> consumer buffer."""
        if self._scratch is None:
            self._scratch = ops.Scratch(self.device)
        return self._scratch

    @property
    def scratch_aux(self):
        """Launch scratch of this pipeline's SECOND stream (the uncond half under cfg_split, the trailing VAE decode of
        animate.video.generate_video(overlap_decode=True))."""
        if self._scratch_aux is None:
            self._scratch_aux = ops.Scratch(self.device)
        return self._scratch_aux

    @property
    def aux_stream(self):
        if self._aux_stream is None:
            self._aux_stream = torch.cuda.Stream(self.device)
        return self._aux_stream

    @property
    def decode_side(self):
        """(stream, launch scratch) of the trailing VAE decode of animate.video.generate_video(overlap_decode=True)."""
        if self._decode_stream is None:
            self._decode_stream = torch.cuda.Stream(self.device)
            self._scratch_decode = ops.Scratch(self.device)
        return self._decode_stream, self._scratch_decode

    def reset_graphs(self):
        """Drop every captured UNet graph (after changing a kernel option with ops.set_option, or to release their memory)."""
        self._graphs.clear()

    def to(self, device=None, dtype=None):
        for m in (self.vae, self.reference_unet, self.denoising_unet, self.face_locator, self.image_proj):
            m.to(device=device, dtype=dtype)
            m.prepare()
        # launch scratch, side streams and captured graphs belong to the device they were made on
        self._scratch = self._scratch_aux = self._aux_stream = self._scratch_decode = self._decode_stream = None
        self._graphs.clear()
        return self

    @property
    def device(self):
        return self.denoising_unet.device

    def progress_bar(self, iterable=None, total=None):
        return iterable if iterable is not None else range(total)

    # ------------------------------------------------------------------------------------------
    def prepare_latents(self, batch_size, num_channels_latents, width, height, video_length, dtype, device, generator,
                        latents=None):
        """face_animate.py:136-188 via diffusers randn_tensor: a CPU generator samples on the CPU in the
        weight dtype, then the tensor moves to the device; scaled by init_noise_sigma (= 1)."""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if latents is None:
            gdev = generator.device if generator is not None else torch.device("cpu")
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype)
        return latents.to(device) * self.scheduler.init_noise_sigma

    def _image_tokens(self, img, dtype):
        """(n, 3, H, W) any float dtype -> token-major [n, H*W, 8]"""
        n, Cin, H, W = img.shape
        x = img.to(self.device).float().reshape(n, Cin, H * W).contiguous()
        return ops.nchw_to_nhwc(x, n, Cin, H * W, 8, dtype)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, *args, **kwargs):
        with ops.routing(self.routing), ops.scratch_scope(self.scratch):
            return self._call(*args, **kwargs)

    @torch.no_grad()
    def call_batch(self, clips, width, height, video_length, num_inference_steps, guidance_scale=1.0, **kwargs):
        """K INDEPENDENT clips through ONE denoising loop: every UNet evaluation runs on the K x video_length frames of all clips
        (the reference already batches two evaluations for CFG, hallo/animate/face_animate.py:397-417; hallo/models/unet_3d.py:
        510-527 takes any batch).  Each clip is computed exactly as `__call__` would compute it alone -- its own reference /
        motion-frame banks (frame row r reads the bank of clip r // video_length, not the reference's CFG tiling r % batch), face
        tokens, audio tokens, masks, latents -- but the weights are read once for K clips and the 16 x 16 / 8 x 8 levels present
        K times the rows to the GEMM / convolution tiles (4096 / 1024 rows per clip leave most of a 256-CU chip idle or force a
        split-K slab + reduce pass).  Independent clips are the shard unit of the path (scripts/inference.py:285-347 with each
        clip given its reference / motion frames: DESIGN section 8); inside ONE video the clips are sequential and cannot be
        batched.
        clips: list of dicts with the per-clip arguments of `__call__` (ref_image, face_emb, audio_tensor, face_mask,
        pixel_values_full_mask, pixel_values_face_mask, pixel_values_lip_mask, and optionally latents / generator); the other
        arguments are shared.  No classifier-free guidance (guidance_scale <= 1) when K > 1.
        Returns a list of K outputs, each what `__call__` returns for that clip (with decode=False: the latents (1, C, F, h, w))."""
        with ops.routing(self.routing), ops.scratch_scope(self.scratch):
            return self._call_clips(list(clips), width, height, video_length, num_inference_steps, guidance_scale, **kwargs)

    def _call(self, ref_image, face_emb, audio_tensor, face_mask, pixel_values_full_mask, pixel_values_face_mask,
              pixel_values_lip_mask, width, height, video_length, num_inference_steps, guidance_scale,
              num_images_per_prompt=1, eta=0.0, motion_scale=None, generator=None, output_type="tensor",
              return_dict=True, callback=None, callback_steps=1, latents=None, decode=True, **kwargs):
        clip = dict(ref_image=ref_image, face_emb=face_emb, audio_tensor=audio_tensor, face_mask=face_mask,
                    pixel_values_full_mask=pixel_values_full_mask, pixel_values_face_mask=pixel_values_face_mask,
                    pixel_values_lip_mask=pixel_values_lip_mask, generator=generator, latents=latents)
        return self._call_clips([clip], width, height, video_length, num_inference_steps, guidance_scale, eta=eta,
                                motion_scale=motion_scale, output_type=output_type, return_dict=return_dict, callback=callback,
                                callback_steps=callback_steps, decode=decode)[0]

    def _call_clips(self, clips, width, height, video_length, num_inference_steps, guidance_scale, num_images_per_prompt=1,
                    eta=0.0, motion_scale=None, output_type="tensor", return_dict=True, callback=None, callback_steps=1,
                    decode=True, **kwargs):
        if eta != 0.0:
            raise ValueError("the Hallo path runs DDIM with eta = 0 (face_animate.py:420)")
        K = len(clips)
        dev = self.device
        dt = self.denoising_unet.dtype
        den, refnet = self.denoising_unet, self.reference_unet
        do_cfg = guidance_scale > 1.0
        if K < 1 or (K > 1 and do_cfg):
            raise ValueError("call_batch: K >= 1 clips, and no classifier-free guidance for K > 1 (the reference tiles the "
                             "banks of a CFG pair over the batch axis, mutual_self_attention.py:235-247: a batch of CFG pairs "
                             "has no single bank map)")
        if K * video_length * (height // self.vae_scale_factor) * (width // self.vae_scale_factor) > (1 << 20):
            raise ValueError("call_batch: K x frames x latent pixels is limited to 2^20 token rows per evaluation (16 clips of 16 frames at 512 x 512): "
                             "the widest row-major intermediates (rows x 1280 ... 2560) are indexed with 32-bit element counts inside a tile walk")
        B = 2 if do_cfg else K                     # batch entries of one UNet evaluation
        mode = CLIP_BATCH if K > 1 else do_cfg     # the bank rule of the spatial self-attention (models/attention.py)
        Fr = video_length
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        L = h * w
        self.scheduler.set_timesteps(num_inference_steps)
        timesteps = self.scheduler.timesteps

        # -- face tokens (face_animate.py:291-298)
        face_emb = clips[0]["face_emb"] if K == 1 else torch.cat([c["face_emb"].reshape(-1, c["face_emb"].shape[-1]) for c in clips])
        cond = self.image_proj(face_emb)
        if do_cfg:
            enc = torch.cat([self.image_proj(torch.zeros_like(face_emb)), cond], dim=0)
        else:
            enc = cond

        writer = ReferenceAttentionControl(refnet, do_classifier_free_guidance=do_cfg, mode="write", batch_size=1,
                                           fusion_blocks="full")
        reader = ReferenceAttentionControl(den, do_classifier_free_guidance=do_cfg, mode="read", batch_size=1,
                                           fusion_blocks="full")

        # -- latents: fp32 token-major state [K*F*L, C] + the UNet input buffer [B*F, L, 8]
        C_lat = den.in_channels
        lat = torch.cat([self.prepare_latents(1, C_lat, width, height, Fr, dt, dev, c.get("generator"), c.get("latents"))[0]
                         .permute(1, 2, 3, 0).reshape(Fr * L, C_lat).float() for c in clips]).contiguous()
        C0 = den.config.block_out_channels[0]
        split = bool(do_cfg and self.cfg_split)
        sg = None
        ref0, audio0 = clips[0]["ref_image"], clips[0]["audio_tensor"]
        if self.use_graph:
            # prepare() is lazy: after den.load_state_dict() / den.to() the weight images (and prepare_epoch) of the NEXT forward
            # differ from what the attribute says now.  Re-prepare here, so that the epoch in the key is the one step 0 runs
            # with -- otherwise step 0 would rebuild (and free) the images under a graph captured from the old ones.
            den.prepare()
            refnet.prepare()
            ms_key = None if motion_scale is None else tuple(float(m) for m in motion_scale)
            key = (B, K, Fr, h, w, dt, str(dev), ref0.shape[1] if ref0.dim() == 5 else ref0.shape[0], ms_key,
                   tuple(audio0.shape[-2:]), tuple(enc.shape[1:]), ops.options_fingerprint(),
                   bool(getattr(den, "fp8_projections", False)), bool(do_cfg and self.cfg_split), den.prepare_epoch)
            sg = self._graphs.get(key)
            if sg is None:
                # graphs of re-prepared weights (another prepare_epoch) point at freed weight images, and every graph pins the
                # intermediates of one UNet evaluation (GBs): keep the current epoch's, at most MAX_GRAPHS of them
                for k in [k for k in self._graphs if k[-1] != den.prepare_epoch]:
                    del self._graphs[k]
                while len(self._graphs) >= self.MAX_GRAPHS:
                    del self._graphs[next(iter(self._graphs))]
                sg = self._graphs[key] = StepGraph(B, Fr, L, C0, dev, dt, split=split, out_channels=den.conv_out.cout)
        x_in = sg.x_in if sg is not None else torch.zeros((B * Fr, L, 8), device=dev, dtype=dt)
        if do_cfg:
            x_in.view(B, Fr * L, 8)[:, :, :C_lat] = lat.to(dt)
        else:
            x_in.view(K * Fr * L, 8)[:, :C_lat] = lat.to(dt)

        # -- reference + motion frames -> latents (face_animate.py:332-336); per clip: the processor's normalisation rule looks at
        # the whole tensor it is given
        def ref_tokens(r):
            imgs = r.reshape(-1, *r.shape[2:]) if r.dim() == 5 else r
            return preprocess_image(imgs, height, width, normalize=True)          # ref_image_processor.preprocess (:119-121, 333)
        imgs = torch.cat([ref_tokens(c["ref_image"]) for c in clips])
        n_ref = imgs.shape[0]
        per_clip = n_ref // K                                                       # 1 reference + n motion frames
        ref_lat, _, _ = self.vae.encode_tokens(self._image_tokens(imgs, dt), n_ref, height, width, scale=0.18215)

        # -- face locator on one frame per clip, broadcast over the F identical frames (face_animate.py:339-343)
        fm = self._image_tokens(torch.cat([c["face_mask"].reshape(-1, *c["face_mask"].shape[-3:])[:1] for c in clips]), dt)
        fea, _, _ = self.face_locator.forward_tokens(fm, K, height, width)          # [K, L, C0]
        assert fea.shape[-1] == C0
        mask_cond = sg.mask_cond.view(B, Fr, L, C0) if sg is not None else torch.zeros((B, Fr, L, C0), device=dev, dtype=dt)
        if do_cfg:
            mask_cond[B - 1] = fea[0]                                               # uncond half stays zero
        else:
            mask_cond[:] = fea[:, None]
        mask_cond = mask_cond.view(B * Fr, L, C0)

        # -- masks (face_animate.py:345-374) and audio tokens (:377-379)
        def mask_rows(name):
            per = [c[name] for c in clips]
            if do_cfg:
                return [torch.cat([m] * 2) for m in per[0]]
            return list(per[0]) if K == 1 else [torch.cat([p_[d] for p_ in per]) for d in range(len(per[0]))]
        masks = pack_masks(mask_rows("pixel_values_full_mask"), mask_rows("pixel_values_face_mask"),
                           mask_rows("pixel_values_lip_mask"), dev, dt)
        audio = torch.cat([c["audio_tensor"].to(dev, dt).reshape(Fr, *c["audio_tensor"].shape[-2:]) for c in clips])
        if do_cfg:
            audio = torch.cat([torch.zeros_like(audio), audio], dim=0)
        audio = audio.contiguous()

        if split:
            halves = sg.halves if sg is not None else [StepGraph(1, 0, L, 8, dev, dt) for _ in range(2)]
            v_out = sg.v_out if sg is not None else torch.empty((B * Fr, L, den.conv_out.cout), device=dev, dtype=dt)
            for hf in halves:
                hf.cache.begin_clip()
            cache = None
        elif sg is not None:
            cache = sg.cache
            cache.begin_clip()          # step 0 (eager) refreshes the per-clip constants inside their old storage
        else:
            cache = ClipCache()
        lat_view = lambda: lat.view(K, Fr, h, w, C_lat).permute(0, 4, 1, 2, 3)        # (K, C, F, h, w)
        for i, t in enumerate(self.progress_bar(timesteps)):
            if i == 0:
                # ReferenceNet write pass on [ref, m1, m2] (x2 under CFG) at t = 0 (face_animate.py:386-395).  K > 1: the images
                # of clip c attend to clip c's face tokens (the reference TILES enc over the image axis, which is only defined
                # for a CFG pair: mutual_self_attention.py:341-349)
                if K > 1:
                    refnet.written_banks = refnet.forward_tokens(ref_lat, 0, enc.repeat_interleave(per_clip, 0), h, w)
                else:
                    refnet.written_banks = refnet.forward_tokens(ref_lat.repeat(B, 1, 1), 0, enc, h, w)
                reader.update(writer)
            if split:
                v = self._split_eval(halves, sg is not None and i > 0, t, x_in, v_out, enc, den.reference_bank, audio, mask_cond,
                                     masks, motion_scale, Fr, h, w)
                if sg is not None and i > 0:
                    sg.replays += 1
            elif sg is not None and i > 0:
                sg.t_dev.fill_(float(t))
                if sg.graph is None:
                    sg.capture(lambda: den.forward_tokens(x_in, sg.t_dev, enc, den.reference_bank, audio, mask_cond, masks,
                                                          motion_scale, B, Fr, h, w, mode, cache))
                sg.graph.replay()
                sg.replays += 1
                v = sg.out
            else:
                v = den.forward_tokens(x_in, int(t), enc, den.reference_bank, audio, mask_cond, masks, motion_scale, B, Fr,
                                       h, w, mode, cache)
            a_t, a_p = self.scheduler.step_alphas(t)
            ops.cfg_ddim_step(v, lat, x_in, K * Fr * L, C_lat, do_cfg, guidance_scale, a_t, a_p, self.scheduler.step_mode)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, lat_view().to(dt))
        reader.clear()
        writer.clear()
        if sg is None and cache is not None:
            cache.clear()
        outs = []
        for c in range(K):
            lat_c = lat[c * Fr * L:(c + 1) * Fr * L]
            if not decode:
                outs.append(lat_c.view(Fr, h, w, C_lat).permute(3, 0, 1, 2).unsqueeze(0))
                continue
            if output_type == "device":
                # frames stay in HBM ((1, 3, F, H, W) fp32): the sliding-window driver slices the next clip's motion frames
                # from them and converts / copies asynchronously (hallo_amd/animate/video.py)
                v, H, W = self.decode_latents_device(lat_c, Fr, h, w)
                video = v.view(Fr, -1, H, W).permute(1, 0, 2, 3).unsqueeze(0)
            else:
                video = self.decode_latents(lat_c, Fr, h, w)
            outs.append(video if not return_dict else FaceAnimatePipelineOutput(videos=video))
        return outs

    def _split_eval(self, halves, graphed, t, x_in, v_out, enc, banks, audio, mask_cond, masks, motion_scale, Fr, h, w):
        """One CFG evaluation as two B = 1 evaluations: uncond (rows 0..Fr of every [2 Fr, ...] tensor, bank rows 0..2, no bank
        segment in the spatial self-attention) on the auxiliary stream with its own launch scratch, cond (rows Fr.., motion-frame
        features = bank rows 4..5, reference features ALTERNATING between bank rows 0 and 3 over the frames: the reference tiles
        the bank over the batch axis, mutual_self_attention.py:235-247) on the current stream; both write their half of `v_out`;
        the current stream then waits for the auxiliary one, so the caller's fused CFG + DDIM kernel sees both halves.
        graphed: replay (capture on first use) each half's hipGraph."""
        den = self.denoising_unet
        on_gpu = self.device.type == "cuda"
        if on_gpu:
            main, aux = torch.cuda.current_stream(), self.aux_stream
            aux.wait_stream(main)           # x_in (the previous DDIM update) and this clip's banks / constants are ready
        for hi, flag in enumerate((SKIP_BANK, False)):
            hf, rows = halves[hi], slice(hi * Fr, (hi + 1) * Fr)
            with (torch.cuda.stream(aux if hi == 0 else main) if on_gpu else contextlib.nullcontext()), \
                    ops.scratch_scope(self.scratch_aux if hi == 0 else self.scratch):
                args = (enc[hi:hi + 1], banks, audio[rows], mask_cond[rows],
                        [tuple(m[rows] for m in md) for md in masks], motion_scale, 1, Fr, h, w, flag, hf.cache)
                kw = dict(out=v_out[rows], bank_layout=(2, hi, hi * Fr))
                if graphed:
                    hf.t_dev.fill_(float(t))
                    if hf.graph is None:
                        hf.capture(lambda: den.forward_tokens(x_in[rows], hf.t_dev, *args, **kw))
                    hf.graph.replay()
                    hf.replays += 1
                else:
                    den.forward_tokens(x_in[rows], int(t), *args, **kw)
        if on_gpu:
            main.wait_stream(aux)
        return v_out

    # ------------------------------------------------------------------------------------------
    def decode_latents_device(self, lat, frames, h, w, scratch=None):
        """fp32 token-major latents [F*L, C] -> fp32 device tensor [F, 3, H*W] in [0, 1].  scratch: the launch scratch of the
        stream this decode is enqueued on when that is not the pipeline's own (decode_side)."""
        with ops.routing(self.routing), ops.scratch_scope(scratch if scratch is not None else self.scratch):
            return self._decode_latents_device(lat, frames, h, w)

    def _decode_latents_device(self, lat, frames, h, w):
        dt = self.denoising_unet.dtype
        C_lat = lat.shape[-1]
        z = torch.zeros((frames, h * w, 8), device=lat.device, dtype=dt)
        z[:, :, :C_lat] = (lat.view(frames, h * w, C_lat) * (1.0 / 0.18215)).to(dt)
        img, H, W = self.vae.decode_tokens(z, frames, h, w)
        # (x / 2 + 0.5).clamp(0, 1) fused with the token-major -> planar fp32 conversion (face_animate.py:243)
        return ops.nhwc_to_nchw_f32(img, frames, img.shape[-1], H * W, mul=0.5, add=0.5, lo=0.0, hi=1.0), H, W

    def decode_latents(self, lat, frames, h, w):
        """face_animate.py:222-246 -> (1, 3, F, H, W) fp32 on the CPU."""
        v, H, W = self.decode_latents_device(lat, frames, h, w)
        v = v.view(frames, -1, H, W).cpu()
        return v.permute(1, 0, 2, 3).unsqueeze(0)
