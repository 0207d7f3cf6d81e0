"""Transformer blocks of the Hallo hot path on token-major `[frames, H*W, C]` activations.

Reference classes (hallo/models/attention.py): BasicTransformerBlock 79-407 (ReferenceNet),
TemporalBasicTransformerBlock 410-618 (spatial block of the denoising UNet),
AudioTemporalBasicTransformerBlock 621-907 (hierarchical audio cross-attention), with the
forwards that ReferenceAttentionControl installs on the first two
(hallo/models/mutual_self_attention.py:174-368).

What is different from the reference, by design:
  * the reference-feature coupling is explicit data flow (a `ReferenceBank` written by the
    ReferenceNet blocks and read by the denoising blocks) instead of forward monkey-patching;
  * [self ; reference] K/V is never concatenated: hallo_attention walks two K/V segments, and the
    CFG rule "uncond rows attend to themselves only" (mutual_self_attention.py:264-284) is a per-row
    segment extent, not a second attention pass over half the batch + masked scatter;
  * everything that is constant across the DDIM steps of a clip -- K/V of the reference bank,
    of the 4 face tokens and of the 32 audio tokens per frame -- is projected once per clip
    (`ClipCache`), not once per step;
  * q/k/v projections that share an input are single fused GEMMs, GEGLU and residual adds run in
    GEMM epilogues, the audio mask multiply is a GEMM row scale.
"""
import torch
from torch import nn

from .. import ops
from .layers import Attention, Conv1x1, FeedForward, LayerNorm


class ClipCache:
    """Per-clip store of step-invariant tensors, keyed by (module id, tag).  Owned by the pipeline
    (or absent: then everything is recomputed per call, which is what a bare
    `UNet3DConditionModel.forward` drop-in call does).

    `begin_clip()` (graph mode, hallo_amd/animate/face_animate.py) keeps the entries of the previous clip as STORAGE: the
    first `get` of a key in the new clip computes the new value and copies it INTO the old tensors, so the addresses a
    captured hipGraph of the UNet evaluation reads stay valid from clip to clip."""

    def __init__(self):
        self._d = {}
        self._stale = set()

    def get(self, mod, tag, make):
        k = (id(mod), tag)
        v = self._d.get(k)
        if v is None:
            v = make()
            self._d[k] = v
        elif k in self._stale:
            _copy_into(v, make())
            self._stale.discard(k)
        return v

    def begin_clip(self):
        self._stale = set(self._d)

    def clear(self):
        self._d.clear()
        self._stale.clear()


def _copy_into(dst, src):
    if torch.is_tensor(dst):
        if dst.shape != src.shape or dst.dtype != src.dtype:
            raise ValueError("a cached per-clip constant changed shape between clips of one graph key")
        if any(st == 0 and sz > 1 for st, sz in zip(dst.stride(), dst.shape)):
            # a broadcast (expanded) view, e.g. the face tokens repeated over the frames: write its un-broadcast slice
            idx = tuple(slice(0, 1) if (st == 0 and sz > 1) else slice(None) for st, sz in zip(dst.stride(), dst.shape))
            dst[idx].copy_(src[idx])
        else:
            dst.copy_(src)
    elif isinstance(dst, (tuple, list)):
        if len(dst) != len(src):
            raise ValueError("a cached per-clip constant changed arity between clips of one graph key")
        for a, b in zip(dst, src):
            _copy_into(a, b)
    elif dst != src:
        raise ValueError("a cached per-clip constant that is not a tensor changed between clips of one graph key")


class _NoCache:
    def get(self, mod, tag, make):
        return make()


NO_CACHE = _NoCache()
SKIP_BANK = "skip_bank"      # `do_cfg` value of a B = 1 evaluation of the UNCOND half of a CFG pair: no row reads the reference bank
AUDIO_K_PAD_TO_TILE = True   # False (bench.py --audio-kpad8, A/B): K = 3D + 8 for the fused audio-branch GEMM, as in rounds 1-5 (read at prepare())
CLIP_BATCH = "clip_batch"    # `do_cfg` value of a batch of INDEPENDENT clips without CFG: frame row r reads the bank of clip r // frames


class BasicTransformerBlock(nn.Module):
    """ReferenceNet block in write mode (mutual_self_attention.py:223-232, 329-368): banks norm1(x), then
    self-attention, face-token cross-attention, GEGLU feed-forward."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim):
        super().__init__()
        self.norm1 = LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, head_dim)
        self.norm2 = LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, head_dim)
        self.norm3 = LayerNorm(dim)
        self.ff = FeedForward(dim)

    def run(self, x, enc, bank_out):
        """x [n, L, C]; enc [Be, T, Cx]; bank_out: list that receives norm1(x) (the reference feature)."""
        n = x.shape[0]
        nh = self.norm1.run(x)
        bank_out.append(nh)
        _, q, k, v = self.attn1.qkv(nh)
        a = ops.attention(q, k, v, self.attn1.heads, q_prescaled=True)
        x = self.attn1.out(a, residual=x)
        nh = self.norm2.run(x)
        # mutual_self_attention.py:341-349: `encoder_hidden_states.repeat(tmp, 1, 1)` TILES the face tokens
        # over the image axis, so image i attends to enc[i % Be] (uncond/cond alternate under CFG).
        k2, v2 = self.attn2.kv(enc)
        rep = n // enc.shape[0]
        k2, v2 = k2.repeat(rep, 1, 1), v2.repeat(rep, 1, 1)
        q2 = self.attn2.q(nh.view(-1, nh.shape[-1])).view(n, -1, self.attn2.inner)
        a = ops.attention(q2, k2, v2, self.attn2.heads, q_prescaled=True)
        x = self.attn2.out(a, residual=x)
        return self.ff.run(self.norm3.run(x), residual=x)


class TemporalBasicTransformerBlock(nn.Module):
    """Spatial block of the denoising UNet in read mode (mutual_self_attention.py:174-327)."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim):
        super().__init__()
        self.attn1 = Attention(dim, None, heads, head_dim)
        self.norm1 = LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, head_dim)
        self.norm2 = LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.norm3 = LayerNorm(dim)

    def _prepare(self):
        # LayerNorm is never a kernel in this block: norm1 / norm3 are folded into the q|k|v / GEGLU GEMMs (hallo_gemm
        # ln_colsum), norm2 into hallo_face_xattn
        self.attn1._prepare()
        self.attn1.fold_norm(self.norm1)
        self.ff.fold_norm(self.norm3)

    def run(self, x, enc, bank, video_length, do_cfg, cache=NO_CACHE, bank_layout=None, stats=None):
        """x [n = b*f, L, C]; enc [b, T, Cx] face tokens; bank [b*s, L, C] (s = 1 reference + motion frames,
        fp16-rounded: ReferenceAttentionControl.update casts to fp16 whatever the run dtype, :404,452).
        bank_layout = (bank batches bb, this call's first batch, its first global frame row row0): the call evaluates a SLICE
        of the batch (cfg_split: one half of a CFG pair at b = 1) against the bank of the whole batch -- global frame row r
        reads the reference features of bank batch r % bb whatever batch it belongs to (the tiling rule below).
        do_cfg == CLIP_BATCH: the b batch entries are INDEPENDENT clips, each evaluated exactly as a batch-1 call would
        (FaceAnimatePipeline.call_batch): frame row r reads the bank of ITS clip, r // f."""
        n, L, Cd = x.shape
        b = n // video_length
        bb, _, row0 = bank_layout if bank_layout is not None else (b, 0, 0)
        a1 = self.attn1
        _, q, k, v = a1.qkv_ln(x, stats=stats, kv_head_major=True)       # stats: norm1's statistics from proj_in's epilogue
        # head-major K / V (4-D: the 320-channel level on the row-stationary GEMM): the bank's K / V follow, re-laid once per clip
        hm = k.dim() == 4
        bank_tag = "bank_kv_hm" if hm else "bank_kv"
        lay = (lambda kv2: ops.head_major(kv2[0], kv2[1], a1.heads)) if hm else (lambda kv2: kv2)
        hmk = dict(kv1_head_major=True, kv2_head_major=True) if hm else {}
        if do_cfg == CLIP_BATCH:
            assert bank_layout is None
            k2, v2 = cache.get(self, bank_tag, lambda: lay(a1.kv(bank.view(b, -1, L, Cd)[:, 0].to(x.dtype).contiguous())))
            a = ops.attention(q, k, v, a1.heads, k2=k2, v2=v2, kv2_batch_div=video_length, kv2_batch_mod=0, kv2_first_batch=0,
                              q_prescaled=True, **hmk)
        elif do_cfg == SKIP_BANK:
            # the uncond half of a CFG evaluation run on its own (FaceAnimatePipeline(cfg_split=True)): its rows attend to
            # themselves only (mutual_self_attention.py:264-284), the bank segment does not exist for this call
            a = ops.attention(q, k, v, a1.heads, q_prescaled=True, **({"kv1_head_major": True} if hm else {}))
        else:
            def bank_kv():
                ref = bank.view(bb, -1, L, Cd)[:, 0].to(x.dtype)     # d_b[:, 0]: the reference image's features
                if row0 % bb:
                    ref = ref.roll(-(row0 % bb), 0)                  # local row j is global row row0 + j
                return lay(a1.kv(ref.contiguous()))
            k2, v2 = cache.get(self, bank_tag, bank_kv)
            # K/V = [self ; bank]: frame row r reads bank entry r % b (the reference's `.repeat(1, f, 1, 1)` on the
            # 3-D tensor tiles the batch axis, mutual_self_attention.py:235-247); with CFG the first half of the
            # rows (uncond) skips the bank segment (:264-284).
            a = ops.attention(q, k, v, a1.heads, k2=k2, v2=v2, kv2_batch_div=1, kv2_batch_mod=bb,
                              kv2_first_batch=(n // 2 if do_cfg else 0), q_prescaled=True, **hmk)
        x = a1.out(a, residual=x)

        a2 = self.attn2

        def face_kv():
            kf, vf = a2.kv(enc)                                   # [b, T, C]
            T = kf.shape[1]
            # "b n c -> (b f) n c" (transformer_3d.py:189-192): frame row r uses enc[r // f]
            ex = lambda t: t.unsqueeze(1).expand(b, video_length, T, Cd).reshape(n, T, Cd)
            return ex(kf), ex(vf)
        T = enc.shape[1]
        if T == 4 and a2.heads <= 8 and Cd % 32 == 0 and (video_length * L) % 32 == 0:
            # norm2 + to_q + SDPA over the 4 face tokens + to_out + residual as ONE pass over x: with H*T <= 32
            # (head, token) pairs both projections are per-clip constants (hallo_face_xattn)
            def face_consts():
                k0, v0 = a2.kv(enc)                                # [b, T, C]
                return ops.face_xattn_constants(a2.to_q.weight, k0, v0, a2.to_out[0].weight, self.norm2.weight,
                                                self.norm2.bias, a2.heads, x.dtype)
            sg, g, bb, owp = cache.get(self, "face_fused", face_consts)
            if self.ff.takes_stats(n * L):
                # the kernel holds whole output rows: norm3's statistics leave with them
                x, st3 = ops.face_xattn(x.view(n * L, Cd), sg, g, bb, owp, a2.to_out[0].bias, video_length * L, self.norm2.eps,
                                        stats_eps=self.norm3.eps)
                return self.ff.run_ln(x.view(n, L, Cd), stats=st3)
            x = ops.face_xattn(x.view(n * L, Cd), sg, g, bb, owp, a2.to_out[0].bias, video_length * L,
                               self.norm2.eps).view(n, L, Cd)
        else:
            kf, vf = cache.get(self, "face_kv", face_kv)
            nh = self.norm2.run(x)
            q2 = a2.q(nh.view(n * L, Cd)).view(n, L, Cd)
            a = ops.attention(q2, kf, vf, a2.heads, q_prescaled=True)
            x = a2.out(a, residual=x)
        return self.ff.run_ln(x)


class AudioTemporalBasicTransformerBlock(nn.Module):
    """hallo/models/attention.py:621-907: self-attention, then three cross-attentions to the frame's 32
    audio tokens, each masked by the full / face / lip region mask of this block's depth, passed through a
    zero-initialised 1x1 conv, weighted by motion_scale and summed into the residual; GEGLU feed-forward."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim, depth):
        super().__init__()
        self.depth = depth
        self.zero_conv_full = Conv1x1(dim, dim)
        self.zero_conv_face = Conv1x1(dim, dim)
        self.zero_conv_lip = Conv1x1(dim, dim)
        self.attn1 = Attention(dim, None, heads, head_dim)
        self.norm1 = LayerNorm(dim)
        self.attn2_0 = Attention(dim, cross_attention_dim, heads, head_dim)
        self.attn2_1 = Attention(dim, cross_attention_dim, heads, head_dim)
        self.attn2_2 = Attention(dim, cross_attention_dim, heads, head_dim)
        self.norm2 = LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.norm3 = LayerNorm(dim)

    def _prepare(self):
        xs = (self.attn2_0, self.attn2_1, self.attn2_2)
        cv = (self.zero_conv_full, self.zero_conv_face, self.zero_conv_lip)
        D = self.zero_conv_full.weight.shape[0]
        dt = self.zero_conv_full.weight.dtype
        self.w_q3 = torch.cat([a.to_q.weight for a in xs], dim=0).contiguous()         # [3D, D]
        # K rows of the three branches first, then the V rows: q3 / k / v of the three branches form ONE attention
        # problem over 3 x heads heads
        self.w_kv3 = torch.cat([a.to_k.weight for a in xs] + [a.to_v.weight for a in xs], 0).contiguous()  # [6D, Ca]
        # to_out followed by the zero conv is one linear map per branch (the mask between them is a per-row scalar):
        #   zero_conv_i(mask * (a_i Wo_i^T + bo_i)) = (mask * a_i) (Wz_i Wo_i)^T + mask * (Wz_i bo_i) + bz_i
        # so the three branches + their sum are ONE GEMM over K = 3D (+ three mask columns carrying Wz_i bo_i, padded to +64 so that K
        # stays a multiple of the 64-deep K tile: 3D + 8 = 968 / 1928 / 3848 ran 14 / 21 / 56 % slower per launch than 3D + 64 = 1024 / 1984
        # / 3904 -- a ragged last K tile, and no big-tile kernel -- tools/cbench, round 6).
        wz = [c.weight.view(D, D).float() for c in cv]
        wc = [wz[i] @ xs[i].to_out[0].weight.float() for i in range(3)]
        cc = [(wz[i] @ xs[i].to_out[0].bias.float())[:, None] for i in range(3)]
        self.kpad = 64 if ((3 * D) % 64 == 0 and AUDIO_K_PAD_TO_TILE) else 8
        self.w_fused = torch.cat(wc + cc + [torch.zeros((D, self.kpad - 3), device=wc[0].device)], dim=1).to(dt).contiguous()   # [D, 3D+kpad]
        self.bz3 = torch.stack([c.bias.float() for c in cv])                                                   # [3, D] fp32
        # the three LayerNorms fold into the GEMMs that consume them (hallo_gemm ln_colsum)
        self.attn1._prepare()
        self.attn1.fold_norm(self.norm1)
        self.ff.fold_norm(self.norm3)
        self.w_q3_ln, self.g_q3_ln, self.b_q3_ln = ops.fold_layernorm(self.norm2.weight, self.norm2.bias, self.w_q3)

    def run(self, x, audio, masks, motion_scale, cache=NO_CACHE, stats=None):
        """x [n, L, D]; audio [n, 32, Ca]; masks = (full, face, lip), each fp32 [n, L] for this block's depth.
        stats: norm1's statistics of x from proj_in's epilogue, if any."""
        n, L, D = x.shape
        _, q, k, v = self.attn1.qkv_ln(x, stats=stats, kv_head_major=True)
        a = ops.attention(q, k, v, self.attn1.heads, q_prescaled=True, **({"kv1_head_major": True} if k.dim() == 4 else {}))
        st2 = None
        if ops.wants_stats(n * L, 3 * D, D):
            x, st2 = self.attn1.out(a, residual=x, row_parts=True)     # norm2's statistics from to_out's epilogue
        else:
            x = self.attn1.out(a, residual=x)

        def audio_kv():
            T = audio.shape[1]
            return ops.gemm(audio.reshape(n * T, -1), self.w_kv3).view(n, T, 6 * D)
        kv3 = cache.get(self, "audio_kv", audio_kv)
        ms = [1.0, 1.0, 1.0] if motion_scale is None else [float(m) for m in motion_scale]

        # Step-invariant per depth (shared by every audio block of this depth through the clip cache):
        # the fp32 row scales motion_scale[i] * mask_i and the staging buffer of the fused GEMM's A operand, whose
        # last 8 columns hold those scales (columns 3D..3D+2) and zeros.
        def scales():
            return torch.stack([masks[i].reshape(-1).float() * ms[i] for i in range(3)]).contiguous()     # [3, n*L]
        msmask = cache.get(AudioTemporalBasicTransformerBlock, ("msmask", self.depth, n, L), scales)

        KP = 3 * D + self.kpad

        def abuf():
            buf = torch.zeros((n * L, KP), device=x.device, dtype=x.dtype)
            buf[:, 3 * D:3 * D + 3] = msmask.t().to(x.dtype)
            return buf
        A = cache.get(AudioTemporalBasicTransformerBlock, ("abuf", self.depth, n, L, D), abuf)
        bias_c = cache.get(self, "bias_c", lambda: (torch.tensor(ms, device=x.device)[:, None] * self.bz3).sum(0).to(x.dtype))

        x2 = x.view(n * L, D)
        q3 = ops.gemm(x2, self.w_q3_ln, self.b_q3_ln, alpha=ops.q_scale(self.attn2_0.dim_head), ln_colsum=self.g_q3_ln,
                      ln_eps=self.norm2.eps, ln_stats=ops.ln_stats(x2, 3 * D, self.norm2.eps, given=st2)).view(n, L, 3 * D)
        # three branches x heads as one attention launch; output rows pre-scaled by motion_scale[i] * mask_i
        # (attention.py:853-903) and written straight into the fused GEMM's A operand
        ops.attention(q3, kv3[:, :, :3 * D], kv3[:, :, 3 * D:], 3 * self.attn2_0.heads,
                      out=A.view(n, L, KP)[:, :, :3 * D], rowscale=msmask, rowscale_head_div=self.attn2_0.heads,
                      q_prescaled=True)
        if self.ff.takes_stats(n * L):
            x, st3 = ops.gemm(A, self.w_fused, bias_c, residual=x.view(n * L, D), row_parts=True)     # norm3's statistics
            return self.ff.run_ln(x.view(n, L, D), stats=st3)
        x = ops.gemm(A, self.w_fused, bias_c, residual=x.view(n * L, D)).view(n, L, D)
        return self.ff.run_ln(x)
