"""Host-side mirror of the reference's worker interface (backends/base.py:8-39): same names,
same argument meaning, so a job object built for the reference's workers runs unchanged here."""
from __future__ import annotations

from concurrent.futures import Future
from dataclasses import dataclass, field
from typing import Any, Optional, Protocol, Tuple


@dataclass(frozen=True)
class StyleLora:                      # backends/base.py:15-18
    style: Optional[str] = None
    level: int = 0


@dataclass
class GenSpec:                        # backends/base.py:20-27
    prompt: str
    size: str
    steps: int
    cfg: float
    seed: Optional[int] = None
    style_lora: StyleLora = StyleLora()


@dataclass
class GenerateRequest:                # field names of server/lcm_sr_server.py:117-135 read by workers
    prompt: str
    size: str = "512x512"
    num_inference_steps: int = 4
    guidance_scale: float = 1.0
    seed: Optional[int] = None
    style_lora: Any = field(default_factory=StyleLora)


@dataclass
class Job:                            # backends/base.py:8-12
    req: Any
    fut: Future = field(default_factory=Future)
    submitted_at: float = 0.0


class PipelineWorker(Protocol):       # backends/base.py:29-39
    worker_id: int

    def run_job(self, job) -> Tuple[bytes, int]:
        """Return (png_bytes, seed_used)."""

    def run_job_with_latents(self, job) -> Tuple[bytes, int, bytes]:
        """Return (png_bytes, seed_used, latents_bytes); latents_bytes = [1,4,8,8] NCHW little-endian fp16."""
