"""Style -> LoRA registry, mirroring backends/styles.py:6-82 of the reference (one LoRA per style, exclusive
selection, 1-indexed ladder of adapter weights).  The file location can be overridden per style with
LCM_STYLE_<ID>_PATH (the reference hard-codes /models/loras/...)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Optional, Sequence


@dataclass(frozen=True)
class StyleDef:
    id: str
    title: str
    lora_path: str
    adapter_name: str
    levels: Sequence[float]
    required_cross_attention_dim: Optional[int] = 768

    def path(self) -> str:
        return os.environ.get(f"LCM_STYLE_{self.id.upper()}_PATH", self.lora_path)

    def weight_for(self, level: int) -> float:
        lvl = max(1, min(int(level), len(self.levels)))           # clamp 1..N (cuda_worker.py:181-183)
        return float(self.levels[lvl - 1])


STYLE_REGISTRY: Dict[str, StyleDef] = {
    "papercut": StyleDef(id="papercut", title="Papercut", lora_path="/models/loras/PaperCut_SDXL.safetensors",
                         adapter_name="style_papercut", levels=[0.80, 0.90, 1.00, 1.15]),
}


def register_style(sd: StyleDef) -> None:
    STYLE_REGISTRY[sd.id] = sd
