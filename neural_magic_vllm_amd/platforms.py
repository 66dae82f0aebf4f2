"""Platform query used by the capability gates (reference: vllm/platforms/rocm.py:11-15,
vllm/platforms/interface.py).  gfx950 reports (9, 5) -> 95, which passes GPTQ-Marlin's
`>= 80` gate (gptq_marlin.py:113-114) and makes fp8 use the native scaled-mm path
(fp8.py:114-118: use_marlin = capability < 89)."""
from typing import Tuple

import torch


class RocmGfx950Platform:

    @staticmethod
    def get_device_capability(device_id: int = 0) -> Tuple[int, int]:
        if torch.cuda.is_available():
            return torch.cuda.get_device_capability(device_id)
        return (9, 5)  # build box without a GPU: the only supported target

    @staticmethod
    def get_device_name(device_id: int = 0) -> str:
        return torch.cuda.get_device_name(device_id) if torch.cuda.is_available() else "gfx950"


current_platform = RocmGfx950Platform()
