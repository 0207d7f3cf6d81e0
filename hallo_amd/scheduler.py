"""DDIMScheduler for the Hallo path (third-party in the reference: diffusers 0.27.2 `DDIMScheduler`,
constructed at scripts/inference.py:185-192 with configs/inference/default.yaml:77-88; used at
hallo/animate/face_animate.py:285-286,399,420).

Integer schedule arithmetic (timesteps, prev_timestep, alphas_cumprod index) is host-side and must
be BIT-EXACT with diffusers: `timesteps = round(arange(T, 0, -T/n)) - 1` ("trailing"),
`prev_t = t - T // n`, alpha_prev = 1 when prev_t < 0 (set_alpha_to_one).  The fp32 `alphas_cumprod`
table is built with the same torch CPU ops as diffusers (linspace -> zero-terminal-SNR rescale ->
cumprod) so its values are bit-identical too.  The per-element update runs in the fused
hallo_cfg_ddim_step kernel (CFG combine + v-prediction DDIM, eta = 0) on fp32 latents.
"""
import numpy as np
import torch


def rescale_zero_terminal_snr(betas):
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    abs_ = alphas_cumprod.sqrt()
    a0, aT = abs_[0].clone(), abs_[-1].clone()
    abs_ -= aT
    abs_ *= a0 / (a0 - aT)
    ab = abs_ ** 2
    alphas = torch.cat([ab[0:1], ab[1:] / ab[:-1]])
    return 1 - alphas


class _Cfg(dict):
    """Attribute view of the config dict; a missing key is an AttributeError so hasattr / getattr(cfg, k, default) work."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 timestep_spacing="leading", rescale_betas_zero_snr=False, thresholding=False, clip_sample_range=1.0,
                 **unused):
        # everything the fused step kernel does not implement is refused HERE, never silently ignored
        if prediction_type not in ("v_prediction", "epsilon", "sample"):
            raise ValueError(f"prediction_type {prediction_type!r}: v_prediction, epsilon or sample")
        if thresholding:
            raise NotImplementedError("DDIMScheduler(thresholding=True) is not implemented by hallo_cfg_ddim_step")
        if clip_sample and float(clip_sample_range) != 1.0:
            raise NotImplementedError("clip_sample_range != 1.0 is not implemented by hallo_cfg_ddim_step")
        self.config = _Cfg(thresholding=False, clip_sample_range=1.0,
                           num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                           steps_offset=steps_offset, prediction_type=prediction_type,
                           timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr)
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:
            betas = rescale_zero_terminal_snr(betas)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "trailing":
            ts = np.round(np.arange(T, 0, -T / num_inference_steps)).astype(np.int64) - 1
        elif sp == "leading":
            ts = (np.arange(0, num_inference_steps) * (T // num_inference_steps)).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        elif sp == "linspace":
            ts = np.linspace(0, T - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        else:
            raise ValueError(sp)
        self.timesteps = torch.from_numpy(ts)   # host-side: indexing never syncs with the device

    @property
    def step_mode(self):
        """`mode` of ops.cfg_ddim_step (HALLO_DDIM_PRED_* | HALLO_DDIM_CLIP_SAMPLE) for this scheduler's configuration."""
        from .ops import DDIM_CLIP_SAMPLE, DDIM_PRED
        return DDIM_PRED[self.config.prediction_type] | (DDIM_CLIP_SAMPLE if self.config.clip_sample else 0)

    def step_indices(self, timestep):
        """(t, prev_t) as python ints."""
        t = int(timestep)
        return t, t - self.config.num_train_timesteps // self.num_inference_steps

    def step_alphas(self, timestep):
        """(alpha_prod_t, alpha_prod_t_prev) as python floats holding the exact fp32 table values."""
        t, pt = self.step_indices(timestep)
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[pt]) if pt >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_p
