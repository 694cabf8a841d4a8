"""DDPM noise scheduler (Ho et al. 2020), the subset the policy uses.

The reference instantiates two ``diffusers.DDPMScheduler``s (mindmap/diffuser_actor/diffuser_actor.py:147-156:
100 training steps, epsilon prediction, ``scaled_linear`` for position and ``squaredcos_cap_v2`` for rotation).
`diffusers` is not available here; this restates the published algorithm with that library's defaults
(beta range 1e-4..0.02, ``fixed_small`` variance, sample clipping to [-1, 1]).  Numerically unpinned against diffusers.
"""
import math

import torch


class DDPMScheduler:
    def __init__(self, num_train_timesteps: int = 100, beta_schedule: str = "scaled_linear", beta_start: float = 1e-4,
                 beta_end: float = 0.02, clip_sample: bool = True, clip_sample_range: float = 1.0):
        self.num_train_timesteps = int(num_train_timesteps)
        T = self.num_train_timesteps
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, T, dtype=torch.float64)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float64) ** 2
        elif beta_schedule == "squaredcos_cap_v2":
            def alpha_bar(s):
                return math.cos((s + 0.008) / 1.008 * math.pi / 2) ** 2

            betas = torch.tensor([min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), 0.999) for i in range(T)], dtype=torch.float64)
        else:
            raise ValueError(f"unknown beta_schedule {beta_schedule}")
        self.betas = betas.to(torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).to(torch.float32)
        self.clip_sample = clip_sample
        self.clip_sample_range = clip_sample_range
        self.timesteps = torch.arange(T - 1, -1, -1)

    def set_timesteps(self, num_inference_steps: int) -> None:
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps must not exceed num_train_timesteps")
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = (torch.arange(0, num_inference_steps) * ratio).flip(0)

    def _acp_on(self, device) -> torch.Tensor:
        """alphas_cumprod on `device`, copied there once (a host-to-device copy per call would synchronise, and cannot be
        captured in a HIP graph: training.graphed captures the whole training forward)."""
        device = torch.device(device)
        if device.type == "cpu":
            return self.alphas_cumprod
        cache = self.__dict__.setdefault("_acp_cache", {})
        t = cache.get(device)
        if t is None:
            t = cache[device] = self.alphas_cumprod.to(device)
        return t

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps, per-sample timestep (B,)."""
        acp = self._acp_on(original.device)[timesteps]
        shape = (-1,) + (1,) * (original.ndim - 1)
        return acp.sqrt().reshape(shape) * original + (1.0 - acp).sqrt().reshape(shape) * noise

    def step_coefficients(self, t: int):
        """{s1, inv_s2, c0, c1, sigma, clip} of `step` at timestep t as float32 values: x0 = (x - s1*eps)*inv_s2 [clamped];
        prev = c0*x0 + c1*x + sigma*noise.  (torch divides a tensor by a Python scalar as a multiplication by the
        float32 reciprocal; inv_s2 is that reciprocal.)"""
        import numpy as np

        t = int(t)
        stride = self.num_train_timesteps // len(self.timesteps)
        prev_t = t - stride
        acp_t = float(self.alphas_cumprod[t])
        acp_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
        beta_prod_t, beta_prod_prev = 1.0 - acp_t, 1.0 - acp_prev
        alpha_t = acp_t / acp_prev
        beta_t = 1.0 - alpha_t
        sigma = (max(beta_prod_prev / beta_prod_t * beta_t, 1e-20) ** 0.5) if t > 0 else 0.0
        f32 = np.float32
        return [float(f32(beta_prod_t ** 0.5)), float(f32(1.0) / f32(acp_t ** 0.5)), float(f32(acp_prev ** 0.5 * beta_t / beta_prod_t)),
                float(f32(alpha_t ** 0.5 * beta_prod_prev / beta_prod_t)), float(f32(sigma)),
                float(self.clip_sample_range) if self.clip_sample else 0.0]

    def step(self, model_output: torch.Tensor, t: int, sample: torch.Tensor, generator=None, noise: torch.Tensor = None) -> torch.Tensor:
        """One reverse step x_t -> x_{t-1} from the predicted noise (the timestep grid may be strided).  `noise`: standard
        normal tensor of the sample's shape to use for the variance term instead of drawing one (pre-drawn noise makes the
        step a pure function of tensors: capturable in a HIP graph, and identical between eager and captured sampling)."""
        t = int(t)
        stride = self.num_train_timesteps // len(self.timesteps)
        prev_t = t - stride
        acp_t = float(self.alphas_cumprod[t])
        acp_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
        beta_prod_t, beta_prod_prev = 1.0 - acp_t, 1.0 - acp_prev
        alpha_t = acp_t / acp_prev
        beta_t = 1.0 - alpha_t
        x0 = (sample - beta_prod_t ** 0.5 * model_output) / acp_t ** 0.5
        if self.clip_sample:
            x0 = x0.clamp(-self.clip_sample_range, self.clip_sample_range)
        c_x0 = acp_prev ** 0.5 * beta_t / beta_prod_t
        c_xt = alpha_t ** 0.5 * beta_prod_prev / beta_prod_t
        prev = c_x0 * x0 + c_xt * sample
        if t > 0:
            variance = max(beta_prod_prev / beta_prod_t * beta_t, 1e-20)
            if noise is None:
                noise = torch.randn(sample.shape, dtype=sample.dtype, device=sample.device, generator=generator)
            prev = prev + variance ** 0.5 * noise
        return prev
