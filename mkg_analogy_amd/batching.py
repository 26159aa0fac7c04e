"""Device-side batch assembly for the image slots.

The reference collator (MarT/data/data_module.py:126-142) looks every entity up with ``list.index`` (O(E) per slot), stacks
two [3,224,224] CPU tensors per example and ships ~1.2 MB/example over PCIe.  Here the table stays in HBM
(``model.set_image_table``), the collator only emits an int32 [B,2] row index (-1 = the zero image of a missing slot) and
the patch matrix is gathered by ``mart_patchify_gather``.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch


class DeviceImageTable:
    def __init__(self, entities: Sequence[str]):
        self.row: Dict[str, int] = {}
        for i, e in enumerate(entities):           # first occurrence wins, like list.index
            self.row.setdefault(e, i)

    def slots(self, head_ent: Sequence[Optional[str]], tail_ent: Sequence[Optional[str]]) -> torch.Tensor:
        """[B,2] int32 row indices with the reference's slot rules:
        both present -> (head, tail); otherwise entity = head if head is not None else tail -> (entity or zero, zero)."""
        out = torch.full((len(head_ent), 2), -1, dtype=torch.int32)
        for b, (h, t) in enumerate(zip(head_ent, tail_ent)):
            if h and t:
                out[b, 0], out[b, 1] = self.row[h], self.row[t]
            else:
                e = h if h is not None else t
                if e:
                    out[b, 0] = self.row[e]
        return out
