"""``LinearWarmupCosineAnnealingLR`` as a pure function of the epoch (src/models/lr_scheduler.py:14-64): linear warm-up
from ``warmup_start_lr`` to the base rate over ``warmup_epochs``, then a cosine decay to ``eta_min`` at ``max_epochs``.
The optimizer it would drive (fused AdamW over four parameter groups) is not built yet."""
from __future__ import annotations

import math
from typing import List, Sequence


def warmup_cosine_lr(epoch: int, base_lrs: Sequence[float], warmup_epochs: int, max_epochs: int,
                     warmup_start_lr: float, eta_min: float = 0.0) -> List[float]:
    warmup_epochs, max_epochs = max(0, int(warmup_epochs)), max(1, int(max_epochs))
    if warmup_epochs > 0 and epoch < warmup_epochs:
        prog = epoch / float(warmup_epochs)
        return [warmup_start_lr + (lr - warmup_start_lr) * prog for lr in base_lrs]
    total = max(1, max_epochs - warmup_epochs)
    prog = min((epoch - warmup_epochs) / float(total), 1.0)
    return [eta_min + (lr - eta_min) * 0.5 * (1.0 + math.cos(math.pi * prog)) for lr in base_lrs]
