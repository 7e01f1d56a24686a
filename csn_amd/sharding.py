"""Sharding of the K-neighbour shape graph across the GPUs of one node.

The CSA path shards by query shape: rank r owns shapes [r*B, (r+1)*B) of an S = B*world collection and
computes their cross-shape attention.  The only data-path exchange is the neighbours' point features
(constants: they carry no gradient, so there is no reverse exchange); the 11 weight gradients are summed
at the end of the step.  One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
MI355X node; "gloo" in the CPU tests).

Nothing here touches the attention arithmetic — it is host logic that runs identically on CPU tensors,
which is how tests/test_sharding_gloo.py covers the N > 1 path without a GPU.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional

import numpy as np
import torch
import torch.distributed as dist


def regular_graph(n_shapes: int, K: int, seed: int = 4321) -> np.ndarray:
    """(S, K) int64: K distinct neighbours per shape drawn from the other S-1 shapes (never self), the same on
    every rank (seeded).  Stand-in for the kNN graph that get_knn_graph produces (csa_models.py:270-280)."""
    rng = np.random.default_rng(seed)
    rows = [(s + 1 + rng.choice(n_shapes - 1, size=K, replace=False)) % n_shapes for s in range(n_shapes)]
    return np.stack(rows).astype(np.int64)


class ShapeGraphShard:
    """Rank-local view of a shape collection sharded by contiguous ranges of shape ids."""

    def __init__(self, graph: np.ndarray, shapes_per_rank: int, rank: int, world: int, device: torch.device):
        S, K = graph.shape
        if S != shapes_per_rank * world:
            raise ValueError(f"graph has {S} shapes, expected {shapes_per_rank} x {world}")
        self.rank, self.world, self.B, self.K, self.S = rank, world, shapes_per_rank, K, S
        self.device = device
        self.first = rank * shapes_per_rank
        own = graph[self.first:self.first + shapes_per_rank]
        if (own == np.arange(self.first, self.first + shapes_per_rank)[:, None]).any():
            raise ValueError("a shape may not be its own neighbour (slot 0 already is the shape itself)")
        self.local_graph = torch.from_numpy(own).to(device)                   # (B, K) global shape ids
        # slot table of the neighbour stack: row b = [own shape b, its K neighbours] as global shape ids
        ids = np.concatenate((np.arange(self.first, self.first + shapes_per_rank)[:, None], own), axis=1)
        self.stack_ids = torch.from_numpy(ids.reshape(-1)).to(device)          # (B * (K+1),)
        self._gathered: Optional[torch.Tensor] = None
        self._stack: Optional[torch.Tensor] = None
        # neighbour-only exchange plan (the same on every rank: it is a function of the global graph).  need[r][s] = sorted
        # global ids of the shapes owned by rank s that some shape of rank r has as a neighbour (s != r)
        B = shapes_per_rank
        need = [[np.zeros(0, np.int64) for _ in range(world)] for _ in range(world)]
        for r in range(world):
            nb = np.unique(graph[r * B:(r + 1) * B].reshape(-1))
            for src in range(world):
                if src != r:
                    need[r][src] = nb[(nb >= src * B) & (nb < (src + 1) * B)]
        self._send_ids = torch.from_numpy(np.concatenate([need[r][rank] - self.first for r in range(world)])).to(device)
        self._send_splits = [int(need[r][rank].size) for r in range(world)]
        self._recv_splits = [int(need[rank][src].size) for src in range(world)]
        recv_ids = np.concatenate([need[rank][src] for src in range(world)])          # global ids in arrival order
        # neighbour-stack slots (b, k >= 1): row of the pool [received shapes ; own shapes] that fills them
        pos = {int(g): i for i, g in enumerate(recv_ids)}
        n_recv = int(recv_ids.size)
        pool_rows = np.empty((B, K + 1), np.int64)
        pool_rows[:, 0] = n_recv + np.arange(B)
        for b in range(B):
            for k in range(K):
                g = int(own[b, k])
                pool_rows[b, k + 1] = n_recv + (g - self.first) if self.first <= g < self.first + B else pos[g]
        self._pool_rows = torch.from_numpy(pool_rows.reshape(-1)).to(device)
        self._n_recv = n_recv
        self._pool: Optional[torch.Tensor] = None

    # -- the one data-path collective ----------------------------------------------------------------------
    def exchange(self, feats: torch.Tensor) -> torch.Tensor:
        """feats (B, C, N) of the owned shapes -> (S, C, N) of the whole collection (all-gather)."""
        if self.world == 1:
            return feats
        if self._gathered is None or self._gathered.shape[1:] != feats.shape[1:]:
            self._gathered = torch.empty((self.S,) + tuple(feats.shape[1:]), device=feats.device, dtype=feats.dtype)
        dist.all_gather_into_tensor(self._gathered, feats.contiguous())
        return self._gathered

    def exchange_neighbours(self, feats: torch.Tensor) -> torch.Tensor:
        """Neighbour-only form of exchange + neighbour_stack: every rank sends each other rank only the shapes that rank's
        graph rows reference (one all-to-all with uneven splits: at 8 ranks and K = 3 at most 96 of the 224 remote shapes),
        and the (B, K+1, C, N, 1) stack is gathered out of [received ; own].  The default exchange of bench.py and of
        exchange_async."""
        if self.world == 1:
            raise ValueError("exchange_neighbours needs world > 1")
        self._start_alltoall(feats.contiguous()).wait()
        return self._finish_alltoall(feats)

    def exchange_async(self, feats: torch.Tensor, mode: str = "alltoall", reuse_descriptors: bool = True,
                       payload_dtype: Optional[torch.dtype] = None) -> "PendingStack":
        """Start the exchange and return at once: ``wait()`` on the result yields the neighbour stack.  A model that is handed
        the pending object (CrossShapeAt accepts it in place of the neighbour tensor) runs the evaluations that need no
        neighbour data — the self-attention of its own shapes — while the exchange is in flight.
        mode "alltoall" (default): neighbour-only exchange, every rank receives just the shapes its graph rows name (one
        all_to_all_single with uneven splits; at 8 ranks and K = 3 at most 96 of the 224 remote shapes); "allgather": the
        whole collection to every rank (RCCL's stock all-gather), kept as the fallback.
        reuse_descriptors: the pending object also offers ``gather_pooled`` (see PendingStack), with which the model takes
        the neighbours' pooled SSA descriptors from their owners instead of recomputing SSA(x_k) for every use.
        payload_dtype (mode "alltoall"): the type the features cross xGMI in — torch.bfloat16 halves the bytes (164 MB per rank
        and 32 shapes instead of 328 MB, what SURVEY.md §8(e) budgets for config 4); the received shapes are widened back to the
        features' own type, which is exact for a consumer that rounds its operands to bf16 anyway (math mode ``bf16``) and a
        2^-9 relative perturbation of the neighbour features otherwise."""
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            raise ValueError("exchange_async needs a process group (world > 1, or a world of one rank for rehearsals)")
        feats = feats.contiguous()
        if mode == "allgather":
            if self._gathered is None or self._gathered.shape[1:] != feats.shape[1:]:
                self._gathered = torch.empty((self.S,) + tuple(feats.shape[1:]), device=feats.device, dtype=feats.dtype)
            work = dist.all_gather_into_tensor(self._gathered, feats, async_op=True)
        elif mode == "alltoall":
            work = self._start_alltoall(feats, payload_dtype)
        else:
            raise ValueError(f"unknown exchange mode {mode!r}")
        return PendingStack(self, feats, work, mode, reuse_descriptors)

    def _start_alltoall(self, feats: torch.Tensor, payload_dtype: Optional[torch.dtype] = None):
        tail = tuple(feats.shape[1:])
        if self._pool is None or self._pool.shape[1:] != tail or self._pool.dtype != feats.dtype:
            self._pool = torch.empty((self._n_recv + self.B,) + tail, device=feats.device, dtype=feats.dtype)
        self._send = (feats.index_select(0, self._send_ids) if self._send_ids.numel() else feats[:0]).contiguous()
        self._recv_narrow = None
        recv = self._pool[:self._n_recv]
        if payload_dtype is not None and payload_dtype != feats.dtype:
            self._send = self._send.to(payload_dtype)
            recv = self._recv_narrow = torch.empty((self._n_recv,) + tail, device=feats.device, dtype=payload_dtype)
        self.payload_bytes = (self._send.numel() * self._send.element_size(), recv.numel() * recv.element_size())   # (sent, received)
        return dist.all_to_all_single(recv, self._send, self._recv_splits, self._send_splits, async_op=True)

    def _finish_alltoall(self, feats: torch.Tensor) -> torch.Tensor:
        tail = tuple(feats.shape[1:])
        if getattr(self, "_recv_narrow", None) is not None:
            self._pool[:self._n_recv].copy_(self._recv_narrow)             # widened to the features' type
        self._pool[self._n_recv:].copy_(feats)
        shape = (self.B * (self.K + 1),) + tail
        if self._stack is None or self._stack.shape != shape or self._stack.dtype != feats.dtype:
            self._stack = torch.empty(shape, device=feats.device, dtype=feats.dtype)
        torch.index_select(self._pool, 0, self._pool_rows, out=self._stack)
        return self._stack.view((self.B, self.K + 1) + tail).unsqueeze(-1)

    def neighbour_stack(self, feats: torch.Tensor, collection: torch.Tensor) -> torch.Tensor:
        """(B, K+1, C, N, 1) exactly as CSADatasetK hands it to the model (features_data_loader.py:124-140):
        slot 0 = the shape itself, slots 1..K = its neighbours in graph order."""
        if self.world == 1 or collection.shape[0] != self.S:
            nb = collection[self.local_graph]                                  # (B, K, C, N)
            return torch.cat((feats[:, None], nb), dim=1).unsqueeze(-1)
        # one indexed gather out of the all-gathered collection (slot 0 is the rank's own copy inside it): a single pass over
        # the 1.3 GB stack instead of a gather plus a concatenation
        shape = (self.B * (self.K + 1),) + tuple(collection.shape[1:])
        if self._stack is None or self._stack.shape != shape or self._stack.dtype != collection.dtype:
            self._stack = torch.empty(shape, device=collection.device, dtype=collection.dtype)
        torch.index_select(collection, 0, self.stack_ids, out=self._stack)
        return self._stack.view((self.B, self.K + 1) + tuple(collection.shape[1:])).unsqueeze(-1)

    # -- gradient reduction ------------------------------------------------------------------------------------
    def allreduce_grads(self, params: Iterable[torch.nn.Parameter], average: bool = True) -> None:
        """Sum (or average) the weight gradients over ranks in ONE bucket (≈0.4 M parameters: latency-bound)."""
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            return
        _allreduce_bucket(list(params), self.world, None, average)


def _allreduce_bucket(plist: List[torch.nn.Parameter], world: int, group, average: bool) -> None:
    """Sum (or average) the gradients of `plist` over the ranks in ONE bucket.  The bucket covers EVERY parameter handed in, a
    missing gradient as zeros, so its size is the same on every rank by construction (a bucket of "whoever has a grad" hangs or
    corrupts the collective the day the sets differ).  A parameter that had no gradient on ANY rank keeps ``grad = None``
    afterwards — as in a single process, where optimizers skip such parameters (no weight decay, no moment update): the bucket
    carries one presence count per parameter for that, read only on a rank that has a missing gradient itself — a host sync
    (``.tolist()``) that only such a rank pays: correct, but it shows as per-rank skew in a max-over-ranks step time.
    One bucket = one tensor: all parameters must share dtype and device (asserted; true of this model, fp32 on one GPU).
    An empty list is a no-op — on EVERY rank or on none: the collective is skipped, so the lists must agree in length."""
    if not plist:
        return
    dev, dt = plist[0].device, plist[0].dtype
    assert all(p.device == dev and p.dtype == dt for p in plist), "gradient bucket: parameters of one dtype on one device"
    missing = [p.grad is None for p in plist]
    present = torch.tensor([0.0 if m else 1.0 for m in missing], device=dev, dtype=dt)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in plist] + [present])
    dist.all_reduce(flat, group=group)
    n_par = len(plist)
    counts = flat[-n_par:].tolist() if any(missing) else None          # (a host sync only where a gradient is missing locally)
    if average:
        flat[:-n_par] /= world
    off = 0
    for i, p in enumerate(plist):
        piece = flat[off:off + p.numel()].view_as(p)
        if p.grad is not None:
            p.grad.copy_(piece)
        elif counts[i] > 0:
            p.grad = piece.clone()                                   # another rank used this parameter: its gradient counts here too
        off += p.numel()


class _GatherDescriptors(torch.autograd.Function):
    """(B, C) per-rank rows -> (S, C) table of all ranks' rows (all-gather, rank order).  Backward: the table's gradient is
    summed over the ranks (every consumer's contribution to a descriptor goes back to its owner) and the own rows returned
    — the "descriptor-gradient reduce-scatter" of SURVEY §8e (2), as one all-reduce of S*C floats (256 KB at 256 shapes)."""

    @staticmethod
    def forward(ctx, own: torch.Tensor, first: int, n_shapes: int):
        ctx.first, ctx.rows = first, own.shape[0]
        table = own.new_empty((n_shapes,) + tuple(own.shape[1:]))
        dist.all_gather_into_tensor(table, own.contiguous())
        return table

    @staticmethod
    def backward(ctx, dtable):
        dtable = dtable.contiguous().clone()
        dist.all_reduce(dtable)
        return dtable[ctx.first:ctx.first + ctx.rows], None, None


class PendingStack:
    """A neighbour stack whose exchange is still in flight (ShapeGraphShard.exchange_async)."""

    def __init__(self, shard: ShapeGraphShard, feats: torch.Tensor, work, mode: str = "allgather", reuse_descriptors: bool = False):
        self._shard, self._feats, self._work, self._mode = shard, feats, work, mode
        self._stack: Optional[torch.Tensor] = None
        self.reuse_descriptors = reuse_descriptors

    def wait(self) -> torch.Tensor:
        if self._stack is None:
            self._work.wait()                                  # orders the compute stream behind the collective
            if self._mode == "alltoall":
                self._stack = self._shard._finish_alltoall(self._feats)
            else:
                self._stack = self._shard.neighbour_stack(self._feats, self._shard._gathered)
        return self._stack

    def gather_pooled(self, own_pooled: torch.Tensor) -> torch.Tensor:
        """own_pooled (B, C): the pooled SSA descriptors of the rank's own shapes (mean over the points of SSA(x),
        csa_models.py:211-212).  Returns (B, K, C): the descriptors of every shape's K neighbours, taken from the neighbours'
        OWNERS through one small all-gather (differentiable: the gradient a consumer puts on a neighbour's descriptor is
        summed back into the owner's evaluation).  A shape's descriptor is thus computed once in the whole job instead of
        once per use (csa_models.py:214-220 recomputes SSA(x_k) for every (shape, neighbour) pair: K of the 2K+2
        evaluations per shape)."""
        table = _GatherDescriptors.apply(own_pooled, self._shard.first, self._shard.S)
        return table[self._shard.local_graph]


# -- mini-batch training over a collection that is resident across the ranks (SURVEY.md §8f row 4 + §8e) ---------------------
class BatchExchangePlan:
    """What one training step of one rank moves: the rank's batch (global shape ids), the owned shapes it sends to each other
    rank, what it receives, and where every slot of its (B, K+1) neighbour stack comes from — the rank's own cache or the
    receive pool.  A pure function of (the step's batches of ALL ranks, the neighbour table, the ownership ranges)."""
    __slots__ = ("ids", "ids_local", "send_local", "send_splits", "recv_splits", "n_recv", "rows_local", "src_local", "rows_remote",
                 "src_remote", "B", "K")


class PendingBatchStack:
    """The neighbour stack of a mini-batch whose exchange is in flight; CrossShapeAt takes it in place of the neighbour tensor
    and runs the self-attention of the batch's own shapes under the exchange (``wait()`` -> (B, K+1, C, N, 1), slot 0 = the
    shape itself).  No descriptor reuse: a neighbour is in general not part of any rank's batch in this step, so its pooled
    descriptor is computed by the consumer, as the reference does (csa_models.py:214-220)."""
    reuse_descriptors = False

    def __init__(self, coll: "ResidentCollection", plan: BatchExchangePlan, work, pool: torch.Tensor):
        self._coll, self._plan, self._work, self._pool = coll, plan, work, pool
        self._stack: Optional[torch.Tensor] = None

    def wait(self) -> torch.Tensor:
        if self._stack is None:
            if self._work is not None:
                self._work.wait()
            self._stack = self._coll._assemble(self._plan, self._pool)
        return self._stack


class ResidentCollection:
    """A training collection of any size held in the HBM of all ranks together: rank r owns the contiguous id range
    [bounds[r], bounds[r+1]) as a csn_amd.data.DeviceFeatureCache (every feature file read once, by its owner), and a
    training step takes a mini-batch of B OWNED shapes per rank whose K neighbours live anywhere.

    The reference iterates batches of ~4 shapes over hundreds to thousands (csa_training.py:191-222) and re-reads K neighbour
    files per item (features_data_loader.py:107-140); here a step's neighbour features cross xGMI in ONE neighbour-only
    all-to-all with per-step splits (``exchange_async``), overlapped by the model with the self-attention of the batch's own
    shapes.  Batches are drawn by a sampler that is a pure function of (seed, epoch) and identical on every rank
    (``epoch_batches``), so every rank derives all ranks' batches — and with them the step's exchange plan — without any
    metadata traffic; plans are cached per batch composition.  ShapeGraphShard is the special case "the collection IS the
    batch" that the weak-scaling bench times."""

    def __init__(self, cache, nbr_table, rank: int, world: int, bounds=None, group=None):
        self.cache, self.rank, self.world, self.group = cache, rank, world, group
        self.table = np.asarray(nbr_table, dtype=np.int64)                      # (S, K) global neighbour ids (data.neighbour_table)
        self.S, self.K = self.table.shape
        self.bounds = np.asarray(self.split_bounds(self.S, world) if bounds is None else bounds, dtype=np.int64)
        if self.bounds.shape != (world + 1,) or self.bounds[0] != 0 or self.bounds[-1] != self.S or (np.diff(self.bounds) < 0).any():
            raise ValueError("bounds must be world + 1 non-decreasing shape ids from 0 to S")
        if cache.first != self.bounds[rank] or len(cache) != self.bounds[rank + 1] - self.bounds[rank]:
            raise ValueError(f"rank {rank} owns [{self.bounds[rank]}, {self.bounds[rank + 1]}) but its cache holds "
                             f"[{cache.first}, {cache.first + len(cache)})")
        self.device = cache.device
        self._plans = {}
        self._pool: Optional[torch.Tensor] = None

    @staticmethod
    def split_bounds(n_shapes: int, world: int) -> np.ndarray:
        """Ownership ranges of an n_shapes collection over world ranks: contiguous, sizes differing by at most one."""
        base, extra = divmod(n_shapes, world)
        return np.concatenate(([0], np.cumsum([base + (r < extra) for r in range(world)]))).astype(np.int64)

    @classmethod
    def from_source(cls, source, nbr_table, device, rank: int, world: int, n_points: Optional[int] = None, group=None):
        """Read this rank's share of a feature dataset (csn_amd.data.FeaturesDataset / CSADatasetK / any dataset with their
        item contract) into its HBM and wrap it."""
        from .data import DeviceFeatureCache, N_POINTS
        b = cls.split_bounds(len(source), world)
        cache = DeviceFeatureCache(source, device, first=int(b[rank]), count=int(b[rank + 1] - b[rank]),
                                   n_points=N_POINTS if n_points is None else n_points)
        return cls(cache, nbr_table, rank, world, bounds=b, group=group)

    def owner(self, ids) -> np.ndarray:
        return np.searchsorted(self.bounds, np.asarray(ids), side="right") - 1

    # -- the sampler: the same on every rank ---------------------------------------------------------------------------
    def epoch_batches(self, batch_size: int, epoch: int = 0, shuffle: bool = True, seed: int = 0):
        """The steps of one epoch: a list of steps, each a list over ranks of B global shape ids owned by that rank.  Every rank
        walks ITS shapes (in a permutation drawn from (seed, epoch, rank) when shuffling); the number of steps is that of the
        rank with the most shapes, ranks with fewer wrap around — every rank runs every step (the collectives line up) with a
        full batch.  Consequences, stated rather than hidden: ownership differs by at most one shape per rank (split_bounds), so
        a rank with fewer shapes revisits at most batch_size - 1 + 1 of them in the epoch's last step, and the value
        train_layers_sharded returns is the mean over this rank's STEPS, not over distinct shapes.  batch_size above the smallest
        ownership makes that rank's batches repeat shapes INSIDE a batch: allowed, with a warning."""
        per_rank = []
        for r in range(self.world):
            own = np.arange(self.bounds[r], self.bounds[r + 1])
            if own.size == 0:
                raise ValueError(f"rank {r} owns no shape: {self.S} shapes over {self.world} ranks")
            if shuffle:
                own = np.random.default_rng([seed, epoch, r]).permutation(own)
            per_rank.append(own)
        short = [r for r, o in enumerate(per_rank) if o.size < batch_size]
        if short and not getattr(self, "_warned_short", False):
            # a wrap-around inside ONE batch puts the same shape into it more than once: legal (the collectives still line up,
            # the replicas stay identical) but that rank's loss and gradient weigh the repeated shapes more — say so, once
            import warnings
            warnings.warn(f"batch_size {batch_size} exceeds the {min(o.size for o in per_rank)} shapes rank {short[0]} owns: its "
                          "batches repeat shapes (loss and gradients of that rank weigh them accordingly)", stacklevel=2)
            self._warned_short = True
        n_steps = max((o.size + batch_size - 1) // batch_size for o in per_rank)
        steps = []
        for t in range(n_steps):
            steps.append([np.take(o, np.arange(t * batch_size, (t + 1) * batch_size), mode="wrap") for o in per_rank])
        return steps

    # -- a step's exchange ---------------------------------------------------------------------------------------------
    def plan(self, batches) -> BatchExchangePlan:
        """The exchange plan of one step from the batches of ALL ranks (sequence over ranks of global shape ids)."""
        key = tuple(np.asarray(b, dtype=np.int64).tobytes() for b in batches)
        hit = self._plans.get(key)
        if hit is not None:
            return hit
        if len(batches) != self.world:
            raise ValueError(f"{len(batches)} batches for {self.world} ranks")
        me, lo, hi = self.rank, int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        need = []                                            # need[r][src]: sorted ids owned by src that rank r's batch references
        for r, ids in enumerate(batches):
            ids = np.asarray(ids, dtype=np.int64)
            if ids.size and ((self.owner(ids) != r).any()):
                raise ValueError(f"rank {r}'s batch holds shapes it does not own")
            nb = np.unique(self.table[ids].reshape(-1)) if ids.size else np.zeros(0, np.int64)
            own_r = self.owner(nb)
            need.append([nb[own_r == src] if src != r else np.zeros(0, np.int64) for src in range(self.world)])
        p = BatchExchangePlan()
        ids = np.asarray(batches[me], dtype=np.int64)
        p.ids, p.B, p.K = ids, int(ids.size), self.K
        p.ids_local = torch.from_numpy(ids - lo).to(self.device)           # validated above on the host: batch() needs no sync
        p.send_local = torch.from_numpy(np.concatenate([need[r][me] - lo for r in range(self.world)])).to(self.device)
        p.send_splits = [int(need[r][me].size) for r in range(self.world)]
        p.recv_splits = [int(need[me][src].size) for src in range(self.world)]
        recv_ids = np.concatenate([need[me][src] for src in range(self.world)])      # global ids in arrival order
        p.n_recv = int(recv_ids.size)
        pos = {int(g): i for i, g in enumerate(recv_ids)}
        # stack rows (b, k): k = 0 the shape itself, k >= 1 its neighbours in graph order
        slot_ids = np.concatenate((ids[:, None], self.table[ids]), axis=1).reshape(-1) if ids.size else np.zeros(0, np.int64)
        local = (slot_ids >= lo) & (slot_ids < hi)
        rows = np.arange(slot_ids.size)
        p.rows_local = torch.from_numpy(rows[local]).to(self.device)
        p.src_local = torch.from_numpy(slot_ids[local] - lo).to(self.device)
        p.rows_remote = torch.from_numpy(rows[~local]).to(self.device)
        p.src_remote = torch.from_numpy(np.array([pos[int(g)] for g in slot_ids[~local]], dtype=np.int64)).to(self.device)
        if len(self._plans) >= 4096:
            self._plans.clear()
        self._plans[key] = p
        return p

    def batch(self, plan: BatchExchangePlan):
        """(feats (B, C, N, 1), labels (B, N)) of the rank's batch — what the model and the loss take."""
        loc = getattr(plan, "ids_local", None)
        if loc is None:
            return self.cache.batch(plan.ids)
        return self.cache.feats.index_select(0, loc).unsqueeze(-1), self.cache.labels.index_select(0, loc)

    def exchange_async(self, plan: BatchExchangePlan) -> PendingBatchStack:
        """Start the step's neighbour-only all-to-all (uneven per-step splits) and return at once."""
        feats = self.cache.feats
        tail = tuple(feats.shape[1:])
        send = (feats.index_select(0, plan.send_local) if plan.send_local.numel() else feats[:0]).contiguous()
        pool = torch.empty((plan.n_recv,) + tail, device=feats.device, dtype=feats.dtype)
        work = None
        if self.world > 1:
            work = dist.all_to_all_single(pool, send, plan.recv_splits, plan.send_splits, group=self.group, async_op=True)
        pending = PendingBatchStack(self, plan, work, pool)
        pending._send = send                                  # (kept alive until the collective has consumed it)
        return pending

    def _assemble(self, plan: BatchExchangePlan, pool: torch.Tensor) -> torch.Tensor:
        feats = self.cache.feats
        tail = tuple(feats.shape[1:])
        stack = torch.empty((plan.B * (plan.K + 1),) + tail, device=feats.device, dtype=feats.dtype)
        if plan.rows_local.numel():
            stack.index_copy_(0, plan.rows_local, feats.index_select(0, plan.src_local))
        if plan.rows_remote.numel():
            stack.index_copy_(0, plan.rows_remote, pool.index_select(0, plan.src_remote))
        return stack.view((plan.B, plan.K + 1) + tail).unsqueeze(-1)

    def neighbour_stack(self, plan: BatchExchangePlan) -> torch.Tensor:
        """Blocking form of exchange_async(plan).wait()."""
        return self.exchange_async(plan).wait()

    def allreduce_grads(self, params: Iterable[torch.nn.Parameter], average: bool = True) -> None:
        """The weight gradients summed (or averaged) over the ranks in one bucket, as ShapeGraphShard.allreduce_grads."""
        if self.world == 1:
            return
        _allreduce_bucket(list(params), self.world, self.group, average)


# -- kNN shape graph, rows of the retrieval matrix sharded by query shape (collective 4) --------------------
def _gather_shards(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenation over ranks (in rank order) of per-rank tensors (n_r, ...) whose leading sizes may differ."""
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    counts = torch.empty((world,), device=local.device, dtype=torch.int64)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = counts.tolist()
    n_max = max(counts)
    if n_max == 0:
        return local
    padded = local if local.shape[0] == n_max else torch.cat(
        (local, local.new_zeros((n_max - local.shape[0],) + tuple(local.shape[1:]))), dim=0)
    out = local.new_empty((world * n_max,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(c == n_max for c in counts):
        return out
    return torch.cat([out[r * n_max:r * n_max + c] for r, c in enumerate(counts)], dim=0)


@torch.no_grad()
def knn_graph_sharded(query_local: torch.Tensor, K: int, measure: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                      cand_local: Optional[torch.Tensor] = None, pair_budget: int = 2 ** 28, group=None) -> torch.Tensor:
    """The (S_q, K+1) int64 kNN table of get_knn_graph (MID-FC/csa_models.py:270-280, called at csa_training.py:157-163),
    built by all ranks together: rank r holds the point-major SSA features (n_r, N, C) of its own query shapes
    (``query_local``) and of its share of the candidate shapes (``cand_local``; None = the queries are the candidates, the
    train-vs-train graph).  The candidates are all-gathered (the same payload as the point-feature exchange of the training
    step), every rank scores ITS query rows against all candidates with ``measure(f_q, f_c) -> (n_q, S_c)`` —
    csn_amd.functional.retrieval_measure on the GPU — takes topk(K+1) locally, and the index rows are all-gathered, so every
    rank returns the whole table, row order = rank order.  Every (query, candidate) score is computed by exactly one rank with
    the arithmetic of the single-process path, so the table is bit-identical to it."""
    cand = _gather_shards(query_local if cand_local is None else cand_local, group)
    n_q, N = query_local.shape[0], query_local.shape[1]
    S_c = cand.shape[0]
    if S_c < K + 1:
        raise ValueError(f"{S_c} candidate shapes cannot give {K + 1} neighbours")
    rows = max(1, min(max(n_q, 1), pair_budget // max(1, S_c * N)))          # bound the per-point maxima scratch
    parts = [measure(query_local[i:i + rows], cand).topk(K + 1, dim=-1)[1] for i in range(0, n_q, rows)]
    mine = torch.cat(parts, dim=0) if parts else torch.empty((0, K + 1), device=query_local.device, dtype=torch.int64)
    return _gather_shards(mine.to(torch.int64), group)
