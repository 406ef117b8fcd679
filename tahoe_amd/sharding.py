"""Multi-GPU use of the forest operator: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI on ROCm).  The reference is single-GPU (no parallelism, no collectives); this is the
8e row of the scope table.

Ways to split one predict over `world` ranks:
  * rows  -- every rank holds the whole forest and a contiguous block of rows.  Rows are independent, so
             there is no data-path collective and every row's float32 sum is bit-identical to 1 GPU.  This is
             what `choose_sharding` picks whenever the forest fits one GPU's memory (288 GB): measured on one
             MI355X, the 8000-tree K4 forest walks 125 k rows in 1/8 of the time of 1 M rows (tree groups keep
             each 2000-tree group cache-resident), so nothing is gained by cutting the forest.
  * trees -- rank k holds trees [T*k/world, T*(k+1)/world) and all rows (north_star's shape for forests that are
             the large object).  Three ways to combine the per-rank sums:
      - "allreduce64" (default): each rank's sequential float32 partial sum (bit-identical to the CPU's partial
        sum over those trees) is widened to float64, ONE all-reduce of 8 bytes per row adds the `world` partials
        without rounding, the total is rounded to float32 once.  Not bit-identical to the CPU's single sequential
        float32 sum -- no association of `world` partials can be -- but provably at least as close to the exact
        sum: error <= gamma(T/world) * sum|leaf| + u * |sum| against the CPU's own gamma(T) * sum|leaf|
        (`sum_error_bound`); the tests and bench.py assert that bound and report both errors against a float64
        CPU sum.  north_star's "1e-6 relative to the CPU sum" is NOT promised by this mode for T in the thousands
        (measured at K4: ~3e-4 relative on rows whose sum nearly cancels, where the CPU's float32 sum is itself
        that far from the exact value).
      - "allreduce32": the same with float32 partials on the wire (4 bytes per row); kept for comparison.
      - "chain": bit-exact.  The batch is cut into row chunks; rank k receives the running float32 sums of chunk
        c from rank k-1 (point-to-point over xGMI, 4 bytes per row), continues them through its own trees
        (tahoe_forest_predict_accumulate) and sends them on while rank k-1 already works on chunk c+1: a
        pipeline of depth `world` whose result, on the last rank, IS the single sequential float32 sum of
        predict_on_cpu (BaseTahoeTest.h:462-466).  Costs (world - 1) chunk times of fill/drain per batch.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np

U32 = 2.0 ** -24  # unit roundoff of float32


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


def choose_sharding(num_trees: int, depth: int, bits_bytes: int = 2, hbm_budget_bytes: int = 200 << 30) -> str:
    """Rows unless the forest itself does not fit one GPU.

    Round 1 cut the forest whenever its reference-format size exceeded the 256 MiB Infinity Cache; the
    measurement (bench.py `k4` legs) says otherwise: a forest larger than the cache is walked one tree
    group at a time, each group stays cache-resident while the rank's rows pass, and row shards need no
    collective and are bit-exact.  Tree shards remain for forests beyond one GPU's HBM (device layout ~ 32 bytes
    per node here against 4 + b in the reference's adaptive format)."""
    device_bytes = num_trees * ((1 << (depth + 1)) - 1) * 32
    return "trees" if device_bytes > hbm_budget_bytes else "rows"


def gamma(n: int) -> float:
    """Higham's gamma_n = n u / (1 - n u) for float32: bound factor of an n-term sequential sum."""
    return n * U32 / (1.0 - n * U32)


def sum_error_bound(abs_leaf_sum, total, trees_per_shard: int):
    """|result - exact| bound for "allreduce64": per-shard sequential float32 sums (<= gamma(n-1) * sum|leaf| each,
    together <= gamma(n-1) * sum over all trees), an exact float64 combination, one rounding to float32."""
    return gamma(max(trees_per_shard - 1, 0)) * np.asarray(abs_leaf_sum, dtype=np.float64) + U32 * np.abs(
        np.asarray(total, dtype=np.float64)) + 1e-300


class TreeShardedForest:
    """Rank-local part of a tree-sharded forest.

    `make_local(nodes, num_trees)` builds the rank's operator (tahoe_amd.Forest on a GPU; the tests pass a
    CPU stand-in to exercise the control flow with gloo).  It must provide predict_raw(data, out=None) and, for
    mode "chain", predict_accumulate(data, sums).  `finish(sums)` applies the output transform in place
    (tahoe_transform_preds on a GPU)."""

    MODES = ("allreduce64", "allreduce32", "chain")

    def __init__(self, nodes: np.ndarray, num_trees: int, depth: int, make_local: Callable, finish: Callable,
                 rank: Optional[int] = None, world: Optional[int] = None, group=None, mode: str = "allreduce64",
                 chunk_rows: int = 32768):
        import torch.distributed as dist

        if mode not in self.MODES:
            raise ValueError(f"mode must be one of {self.MODES}")
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.num_trees = num_trees
        self.mode = mode
        self.chunk_rows = max(int(chunk_rows), 1)
        my_nodes, my_trees, self.tree_range = tree_shard_nodes(nodes, num_trees, depth, self.rank, self.world)
        self.local = make_local(my_nodes, my_trees)
        self.finish = finish
        self._wide = None  # float64 scratch of the all-reduce

    # -- collectives ---------------------------------------------------------------------------------
    def _neighbour(self, offset: int) -> int:
        r = self.rank + offset
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def predict_sums(self, data, out=None):
        """Raw float32 sums over the whole forest.  "chain": valid on the LAST rank (others hold a prefix)."""
        import torch

        if self.mode == "chain":
            rows = data.shape[0]
            sums = out if out is not None else torch.empty(rows, dtype=torch.float32, device=data.device)
            for lo in range(0, rows, self.chunk_rows):
                hi = min(lo + self.chunk_rows, rows)
                part = sums[lo:hi]
                if self.rank > 0:
                    self.dist.recv(part, src=self._neighbour(-1), group=self.group)  # sums of the trees before mine
                else:
                    part.zero_()
                self.local.predict_accumulate(data[lo:hi], part)
                if self.rank < self.world - 1:
                    self.dist.send(part, dst=self._neighbour(+1), group=self.group)
            return sums
        sums = self.local.predict_raw(data, out)
        if self.mode == "allreduce32":
            self.dist.all_reduce(sums, op=self.dist.ReduceOp.SUM, group=self.group)  # 4 bytes per row
            return sums
        if self._wide is None or self._wide.shape != sums.shape or self._wide.device != sums.device:
            self._wide = torch.empty(sums.shape, dtype=torch.float64, device=sums.device)
        self._wide.copy_(sums)  # exact
        self.dist.all_reduce(self._wide, op=self.dist.ReduceOp.SUM, group=self.group)  # 8 bytes per row
        sums.copy_(self._wide)  # one rounding to float32
        return sums

    @property
    def result_rank(self) -> int:
        """Rank (within the group) that holds the complete result: every rank, or the last one for "chain"."""
        return self.world - 1 if self.mode == "chain" else self.rank

    def predict(self, data, out=None, broadcast: bool = False):
        sums = self.predict_sums(data, out)
        if self.mode == "chain" and broadcast and self.world > 1:
            self.dist.broadcast(sums, src=self._neighbour(self.world - 1 - self.rank), group=self.group)
        if self.mode != "chain" or broadcast or self.rank == self.world - 1:
            return self.finish(sums)
        return sums  # a prefix sum on the other ranks of a chain: not a prediction


def gpu_tree_sharded_forest(nodes, num_trees, depth, num_cols, missing=0.0, output=0, threshold=0.0, global_bias=0.0,
                            strategy=0, group=None, mode: str = "allreduce64", chunk_rows: int = 32768) -> TreeShardedForest:
    """TreeShardedForest over tahoe_amd.Forest (raw / continued sums) + tahoe_transform_preds (finish)."""
    from . import capi

    def make_local(my_nodes, my_trees):
        f = capi.Forest(my_nodes, my_trees, depth, num_cols, missing=missing)
        f.set_strategy(strategy)
        return f

    def finish(sums):
        return capi.transform_preds(sums, output, num_trees, threshold, global_bias)

    return TreeShardedForest(nodes, num_trees, depth, make_local, finish, group=group, mode=mode, chunk_rows=chunk_rows)
