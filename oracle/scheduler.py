"""LCMScheduler restated (diffusers is absent; published algorithm).  Oracle only.

Call sites in the reference: backends/cuda_worker.py:88 (from_config),
backends/rknnlcm.py:559-560 (set_timesteps), :596-599 (step).  SURVEY A.3.
"""
from __future__ import annotations

import numpy as np


class LCMSchedulerOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 original_inference_steps=50, timestep_scaling=10.0, sigma_data=0.5):
        self.num_train_timesteps = num_train_timesteps
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod((1.0 - betas).astype(np.float32), dtype=np.float32)
        self.final_alpha_cumprod = np.float32(1.0)
        self.original_inference_steps = original_inference_steps
        self.timestep_scaling = timestep_scaling
        self.sigma_data = sigma_data
        self.init_noise_sigma = 1.0
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int):
        k = self.num_train_timesteps // self.original_inference_steps
        origin = np.arange(1, self.original_inference_steps + 1) * k - 1
        if num_inference_steps > len(origin):
            raise ValueError("num_inference_steps larger than original_inference_steps")
        origin = origin[::-1].copy()
        idx = np.floor(np.linspace(0, len(origin), num=num_inference_steps, endpoint=False)).astype(np.int64)
        self.timesteps = origin[idx].astype(np.int64)
        return self.timesteps

    def scalings(self, t: int):
        s = t * self.timestep_scaling
        c_skip = self.sigma_data ** 2 / (s ** 2 + self.sigma_data ** 2)
        c_out = s / (s ** 2 + self.sigma_data ** 2) ** 0.5
        return c_skip, c_out

    def coefficients(self, i: int):
        """(sqrt_alpha_t, sqrt_beta_t, c_skip, c_out, sqrt_alpha_prev, sqrt_beta_prev, last)."""
        t = int(self.timesteps[i])
        last = i == len(self.timesteps) - 1
        tp = t if last else int(self.timesteps[i + 1])
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[tp]) if tp >= 0 else float(self.final_alpha_cumprod)
        c_skip, c_out = self.scalings(t)
        return (a_t ** 0.5, (1 - a_t) ** 0.5, c_skip, c_out, a_p ** 0.5, (1 - a_p) ** 0.5, last)

    def step(self, eps, i: int, sample, noise=None):
        """epsilon prediction; returns (prev_sample, denoised)."""
        sa, sb, c_skip, c_out, sap, sbp, last = self.coefficients(i)
        x0 = (sample - sb * eps) / sa
        den = c_out * x0 + c_skip * sample
        if last:
            return den, den
        return sap * den + sbp * noise, den
