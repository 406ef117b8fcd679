"""Multi-GPU use of the forest operator: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI on ROCm).  The reference is single-GPU (no parallelism, no collectives); this is the
8e row of the scope table.

Two ways to split one predict over `world` ranks:
  * rows  -- every rank holds the whole forest and a contiguous block of rows.  Rows are independent, so
             there is no data-path collective and every row's float32 sum is bit-identical to 1 GPU.
  * trees -- rank k holds trees [T*k/world, T*(k+1)/world) and all rows; per-row partial float32 sums
             are combined by ONE all-reduce (4 bytes per row), then the output transform runs on the
             total.  Use when the forest, not the batch, is the large object.  The all-reduce adds the
             `world` partials in a different order than the CPU's single sequential sum, so results
             agree with the 1-GPU sums to float32 rounding (<= 1e-6 relative unless the sum cancels),
             not bit for bit.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of n items for `rank` of `world` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return (n * rank) // world, (n * (rank + 1)) // world


def tree_shard_nodes(nodes: np.ndarray, num_trees: int, depth: int, rank: int, world: int):
    """Slice of the reference-encoded node array (tree-major, Struct.h:44-48) owned by `rank`."""
    per_tree = (1 << (depth + 1)) - 1
    lo, hi = shard_bounds(num_trees, rank, world)
    return nodes[lo * per_tree: hi * per_tree], hi - lo, (lo, hi)


def choose_sharding(num_trees: int, depth: int, bits_bytes: int = 2, cache_budget_bytes: int = 200 << 20) -> str:
    """SURVEY.md 8e selector: shard trees when the forest (n*T*(4+b) bytes in the reference's adaptive
    format) exceeds the per-GPU cache budget (256 MiB Infinity Cache minus headroom), else shard rows."""
    forest_bytes = num_trees * ((1 << (depth + 1)) - 1) * (4 + bits_bytes)
    return "trees" if forest_bytes > cache_budget_bytes else "rows"


class TreeShardedForest:
    """Rank-local part of a tree-sharded forest.

    `make_local(nodes, num_trees)` builds the rank's operator (tahoe_amd.Forest on a GPU; the tests pass a
    CPU stand-in to exercise the control flow with gloo).  It must provide predict_raw(data, out=None).
    `finish(sums)` applies the output transform in place (tahoe_transform_preds on a GPU).
    """

    def __init__(self, nodes: np.ndarray, num_trees: int, depth: int, make_local: Callable, finish: Callable,
                 rank: Optional[int] = None, world: Optional[int] = None, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.num_trees = num_trees
        my_nodes, my_trees, self.tree_range = tree_shard_nodes(nodes, num_trees, depth, self.rank, self.world)
        self.local = make_local(my_nodes, my_trees)
        self.finish = finish

    def predict(self, data, out=None):
        sums = self.local.predict_raw(data, out)
        # one collective per batch: 4 bytes per row (a 4 MB message at 1M rows)
        self.dist.all_reduce(sums, op=self.dist.ReduceOp.SUM, group=self.group)
        return self.finish(sums)


def gpu_tree_sharded_forest(nodes, num_trees, depth, num_cols, missing=0.0, output=0, threshold=0.0, global_bias=0.0,
                            strategy=0, group=None) -> TreeShardedForest:
    """TreeShardedForest over tahoe_amd.Forest (raw sums) + tahoe_transform_preds (finish)."""
    from . import capi

    def make_local(my_nodes, my_trees):
        f = capi.Forest(my_nodes, my_trees, depth, num_cols, missing=missing)
        f.set_strategy(strategy)
        return f

    def finish(sums):
        return capi.transform_preds(sums, output, num_trees, threshold, global_bias)

    return TreeShardedForest(nodes, num_trees, depth, make_local, finish, group=group)
