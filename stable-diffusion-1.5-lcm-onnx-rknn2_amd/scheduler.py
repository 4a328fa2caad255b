"""Host-side LCMScheduler tables for the HIP path (the per-step arithmetic itself is the
``lcm_scheduler_step`` kernel).  Mirrors ``LCMScheduler.from_config`` / ``set_timesteps`` as used at
backends/cuda_worker.py:88 and backends/rknnlcm.py:559-560; values in a checkpoint's
``scheduler/scheduler_config.json`` (backends/base.py:45-46) override the defaults.  SURVEY.md A.3.
"""
from __future__ import annotations

import json
import os

import numpy as np


class LCMSchedule:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 original_inference_steps=50, timestep_scaling=10.0, sigma_data=0.5, set_alpha_to_one=True, **_ignored):
        if beta_schedule != "scaled_linear":
            raise ValueError(f"unsupported beta_schedule {beta_schedule!r}")
        self.num_train_timesteps = int(num_train_timesteps)
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, self.num_train_timesteps, dtype=np.float32) ** 2
        self.alphas_cumprod = np.cumprod((1.0 - betas).astype(np.float32), dtype=np.float32)
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.original_inference_steps = int(original_inference_steps)
        self.timestep_scaling = float(timestep_scaling)
        self.sigma_data = float(sigma_data)
        self.init_noise_sigma = 1.0

    @classmethod
    def from_config_file(cls, path: str):
        if not os.path.exists(path):
            return cls()
        with open(path) as f:
            cfg = json.load(f)
        keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "original_inference_steps",
                "timestep_scaling", "sigma_data", "set_alpha_to_one")
        return cls(**{k: cfg[k] for k in keys if k in cfg})

    def timesteps(self, n: int) -> np.ndarray:
        if n < 1:
            raise ValueError("num_inference_steps must be >= 1")
        k = self.num_train_timesteps // self.original_inference_steps
        origin = (np.arange(1, self.original_inference_steps + 1) * k - 1)[::-1].copy()
        if n > len(origin):
            raise ValueError(f"num_inference_steps={n} exceeds original_inference_steps={len(origin)}")
        idx = np.floor(np.linspace(0, len(origin), num=n, endpoint=False)).astype(np.int64)
        return origin[idx].astype(np.int64)

    def step_coefficients(self, ts: np.ndarray, i: int):
        """-> ([sqrt_a_t, sqrt_b_t, c_skip, c_out, sqrt_a_prev, sqrt_b_prev], last)."""
        t = int(ts[i])
        last = i == len(ts) - 1
        tp = t if last else int(ts[i + 1])
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[tp]) if tp >= 0 else self.final_alpha_cumprod
        s = t * self.timestep_scaling
        c_skip = self.sigma_data ** 2 / (s ** 2 + self.sigma_data ** 2)
        c_out = s / (s ** 2 + self.sigma_data ** 2) ** 0.5
        return [a_t ** 0.5, (1 - a_t) ** 0.5, c_skip, c_out, a_p ** 0.5, (1 - a_p) ** 0.5], last
