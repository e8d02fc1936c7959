"""The per-block parameter message of the reference (SubSample.py:4-17), kept as a plain struct for API
parity.  On the GPU path nothing is serialised: factors stay in HBM and `make_subsample` hands out views."""
from dataclasses import dataclass
from typing import Any, Tuple


@dataclass
class SubSample:
    block_pos: Tuple[int, int]      # (row, col) of the grid block
    idx_offset: Tuple[int, int]     # (first user id, first item id) of the block
    P: Any                          # user-factor slice of the block
    Q: Any                          # item-factor slice
    b_u: Any
    b_i: Any
    b: float
    y: Any                          # full implicit-factor matrix (or False when not shipped)
    alpha: float
    beta1: float
    beta2: float
