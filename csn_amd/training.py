"""Caller-side pieces of the CSA hot path, restated from MID-FC/csa_training.py / ssa_training.py so the drop-in
module can be driven exactly like the reference drives its own.

Everything here is host logic around ``model(feats, mode, neighbor_feats)``: the masked cross-entropy, the part-IoU
accumulation, one training epoch with gradient accumulation, validation, the optimizer/schedule the reference uses,
the SSA -> CSA warm start, and the kNN shape-graph (re)construction.  Device-side sums replace the reference's
per-batch ``.item()`` host syncs.
"""
from __future__ import annotations

from typing import Iterable, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BIG_CLASSES = ("Chair", "Lamp", "StorageFurniture", "Table")       # csa_training.py:40


# ---------------------------------------------------------------------------------------------------------
# loss and metrics (csa_training.py:78-134)
# ---------------------------------------------------------------------------------------------------------
def _flatten(logit: torch.Tensor, label: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    n_cls = logit.shape[1]
    return logit.squeeze(-1).permute(0, 2, 1).reshape(-1, n_cls), label.reshape(-1)


def loss_functions_seg(logit: torch.Tensor, label_gt: torch.Tensor, num_class: int, weight_decay: float = 0.0,
                       mask: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Mean cross-entropy and accuracy over the points whose label is > mask (csa_training.py:94-108).  Device logits in fp32
    go through the fused HIP pair (csn_amd.functional.masked_cross_entropy: the class-major logits are read where they lie);
    host tensors — the CPU tests of this module's bookkeeping — take the reference's own sequence of torch calls."""
    if logit.is_cuda and logit.dtype == torch.float32 and label_gt.shape[1:] == logit.shape[2:3]:
        from .functional import masked_cross_entropy
        loss, accu, _ = masked_cross_entropy(logit, label_gt.long(), mask)
        return loss, accu
    flat, lab = _flatten(logit, label_gt)
    keep = torch.where(lab > mask)[0]
    sel, tgt = flat[keep], lab[keep].long()
    loss = F.cross_entropy(sel, tgt)
    accu = (sel.argmax(dim=1) == tgt).float().mean()
    return loss, accu


def IoU_per_shape(pred: torch.Tensor, label: torch.Tensor, class_num: int, mask: int = 0):
    """Per-class intersection / union counts over labelled points (csa_training.py:110-134), as two (class_num,)
    device tensors (the reference keeps python lists of 0-d tensors)."""
    flat, lab = _flatten(pred, label)
    keep = torch.where(lab > mask)[0]
    p, l = flat[keep].argmax(dim=1), lab[keep].long()
    classes = torch.arange(class_num, device=p.device)[:, None]
    pk, lk = p[None, :] == classes, l[None, :] == classes
    return (pk & lk).sum(dim=1).float(), (pk | lk).sum(dim=1).float()


def mean_iou(intsc: torch.Tensor, union: torch.Tensor) -> float:
    """csa_training.py:252-255: sum_k I_k / (U_k + 1e-10) / (class_num - 1)."""
    return float((intsc.double() / (union.double() + 1.0e-10)).sum().item() / (intsc.numel() - 1))


# ---------------------------------------------------------------------------------------------------------
# optimizer / schedule / warm start (csa_training.py:303-336, utils.py:29-39)
# ---------------------------------------------------------------------------------------------------------
def make_optimizer(model: torch.nn.Module, lr: float = 0.001, weight_decay: float = 0.0005):
    opt = torch.optim.Adam((p for p in model.parameters() if p.requires_grad), lr=lr, betas=(0.5, 0.999),
                           weight_decay=weight_decay)                                     # csa_training.py:307
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.1)                  # csa_training.py:308
    return opt, sched


def load_trained_ssa_layers(model: torch.nn.Module, ckpt) -> torch.nn.Module:
    """Copy every tensor of an SSA checkpoint into the (CSA) model by key (utils.py:29-39).  ``ckpt`` is a path or a
    state dict."""
    if isinstance(ckpt, str):
        ckpt = torch.load(ckpt, map_location="cpu")
    own = model.state_dict()
    with torch.no_grad():
        for k, v in ckpt.items():
            own[k].copy_(v)
    return model


# ---------------------------------------------------------------------------------------------------------
# epochs (csa_training.py:191-259, ssa_training.py:125-190)
# ---------------------------------------------------------------------------------------------------------
def _unpack(batch):
    if len(batch) == 3:
        return batch
    feats, label = batch
    return torch.squeeze(feats, dim=1) if feats.dim() == 5 else feats, label, None       # ssa_training.py:135


def train_layers(model, dataloader: Iterable, optimizer, num_class: int, device, accumulation_steps: int = 1,
                 max_batches: Optional[int] = None, reference_semantics: bool = True) -> float:
    """One pass over the loader in train mode (csa_training.py:191-222).  Returns the mean (accumulation-scaled) loss,
    accumulated on the device and read back once.

    ``reference_semantics=True`` (default) reproduces the reference's loop AS WRITTEN, including two things that look like
    slips: ``optimizer.zero_grad()`` runs at the top of EVERY iteration (:196), so with ``gradient_accumulation_steps`` > 1
    only the last micro-batch's (1/steps-scaled) gradient reaches ``optimizer.step()`` (:213-215); and a NaN loss is
    multiplied by 0 (:206-207), which leaves it NaN, so its backward poisons that step.  ``False`` is the repaired loop:
    gradients really accumulate between steps and a NaN loss contributes a zero gradient.  tests/test_training_host.py pins
    both behaviours."""
    model.train()
    total = torch.zeros((), device=device, dtype=torch.float64)
    n = 0
    batches = list(dataloader) if not hasattr(dataloader, "__len__") else dataloader
    n_batches = len(batches)
    optimizer.zero_grad()
    for i, batch in enumerate(batches):
        if reference_semantics:
            optimizer.zero_grad()                                                        # :196
        feats, label, nbrs = _unpack(batch)
        feats, label = feats.to(device), label.to(device)
        out = model(feats, "test", nbrs) if nbrs is not None else model(feats, "train")
        loss, _ = loss_functions_seg(out, label, num_class)
        loss = loss / accumulation_steps
        bad = torch.isnan(loss)
        total += torch.where(bad, torch.zeros_like(loss), loss).detach().double()        # running loss skips NaN (:206-209)
        if reference_semantics:
            loss = torch.where(bad, loss * 0.0, loss)             # :206-207 — NaN * 0 is NaN: the guard does not guard
            loss.backward()
        elif not bool(bad):                                       # repaired: a NaN micro-batch contributes nothing
            loss.backward()
        n += 1
        if (i + 1) % accumulation_steps == 0 or (i + 1) == n_batches:
            optimizer.step()
            optimizer.zero_grad()
        if max_batches is not None and n >= max_batches:
            break
    return float(total.item()) / max(n_batches, 1)


def train_layers_sharded(model, collection, optimizer, num_class: int, batch_size: int, epoch: int = 0, shuffle: bool = True,
                         seed: int = 0, max_steps: Optional[int] = None, steps=None, reference_semantics: bool = True) -> float:
    """One epoch of ``train_layers`` (csa_training.py:191-222, one optimizer step per batch) over a collection that is resident
    across the ranks of a torch.distributed job (csn_amd.sharding.ResidentCollection): every rank trains on mini-batches of
    ``batch_size`` shapes it owns, the batch's neighbour features arrive through the step's neighbour-only exchange — in
    flight under the self-attention of the batch's own shapes — and the weight gradients are averaged over the ranks before
    the optimizer step, so all replicas stay identical.  With world = 1 this is the single-process loop over the same
    batches.  ``steps``: an explicit list of steps (each a list over ranks of shape ids) instead of the epoch's sampler.
    Returns the mean loss of this rank's batches (accumulated on the device, read back once; NaN losses are left out of it, as in
    train_layers).  ``reference_semantics`` as in train_layers: True reproduces the reference's NaN "guard" as written (the NaN
    reaches the gradients — and through the all-reduce every rank), False zeroes the gradients of a NaN batch on its rank."""
    model.train()
    steps = collection.epoch_batches(batch_size, epoch, shuffle, seed) if steps is None else steps
    if max_steps is not None:
        steps = steps[:max_steps]
    params = [p for p in model.parameters() if p.requires_grad]
    total = torch.zeros((), device=collection.device, dtype=torch.float64)
    for batches in steps:
        plan = collection.plan(batches)
        pending = collection.exchange_async(plan)              # neighbour features on their way
        feats, label = collection.batch(plan)
        optimizer.zero_grad()
        out = model(feats, "train", pending)                   # own-shape self-attention first, then pending.wait()
        loss, _ = loss_functions_seg(out, label, num_class)
        bad = torch.isnan(loss)
        total += torch.where(bad, torch.zeros_like(loss), loss).detach().double()     # the running loss skips NaN (:206-209)
        if reference_semantics:
            torch.where(bad, loss * 0.0, loss).backward()      # :206-207 as written: NaN * 0 is NaN — the step is poisoned, as there
        else:
            # repaired: a NaN batch contributes a zero gradient on its rank (decided on the device: every rank still runs the
            # backward and the collective, so the ranks stay in step)
            torch.where(bad, torch.zeros_like(loss), loss).backward()
            for p in params:
                if p.grad is not None:
                    p.grad.nan_to_num_(nan=0.0, posinf=0.0, neginf=0.0)
        collection.allreduce_grads(params, average=True)
        optimizer.step()
    return float(total.item()) / max(len(steps), 1)


@torch.no_grad()
def validate_layers(model, dataloader: Iterable, class_num: int, device, max_batches: Optional[int] = None):
    """Part IoU and mean loss in eval mode (csa_training.py:224-259); no autograd graph is built."""
    model.eval()
    intsc = torch.zeros(class_num, device=device, dtype=torch.float64)
    union = torch.zeros(class_num, device=device, dtype=torch.float64)
    total = torch.zeros((), device=device, dtype=torch.float64)
    n = 0
    for batch in dataloader:
        feats, label, nbrs = _unpack(batch)
        feats, label = feats.to(device), label.to(device)
        out = model(feats, "test", nbrs.to(device)) if nbrs is not None else model(feats, "test")
        loss, _ = loss_functions_seg(out, label, class_num)
        ok = (~torch.isnan(loss)).double()                       # a NaN batch is skipped altogether (:239-240), no host sync
        total += torch.where(torch.isnan(loss), torch.zeros_like(loss), loss).double()
        i_b, u_b = IoU_per_shape(out, label, class_num)
        intsc += i_b.double() * ok
        union += u_b.double() * ok
        n += 1
        if max_batches is not None and n >= max_batches:
            break
    return mean_iou(intsc, union), float(total.item()) / max(n, 1)


# ---------------------------------------------------------------------------------------------------------
# kNN shape graph (csa_training.py:136-176; csa_models.py:270-404)
# ---------------------------------------------------------------------------------------------------------
@torch.no_grad()
def ssa_features(model, loader: Iterable, device) -> torch.Tensor:
    """Point-major SSA features (S, N, C) of every shape of a loader, kept on the device (get_all_feats, :282-300,
    pulls them to the host and the caller pushes them back, csa_training.py:157-161)."""
    model.eval()
    out = []
    for batch in loader:
        feats = batch[0]
        feats = torch.squeeze(feats, dim=1) if feats.dim() == 5 else feats
        out.append(model._ssa_cm(feats.to(device)).permute(0, 2, 1).contiguous())
    return torch.cat(out, dim=0)


@torch.no_grad()
def knn_graph_from_features(model, query: torch.Tensor, cand: torch.Tensor, K: int, pair_budget: int = 2 ** 28) -> torch.Tensor:
    """topk(K+1) candidate ids per query shape, int64 (Sq, K+1) (get_knn_graph, csa_models.py:270-280).  The retrieval
    matrix is built in row chunks so the per-point maxima scratch (Sq_chunk * Sc * N floats) stays bounded."""
    Sq, N, _ = query.shape
    Sc = cand.shape[0]
    rows = max(1, min(Sq, pair_budget // max(1, Sc * N)))
    parts = [model.get_retrieval_measure(query[i:i + rows], cand) for i in range(0, Sq, rows)]
    return torch.cat(parts, dim=0).topk(K + 1, dim=-1)[1]


@torch.no_grad()
def update_knn_graphs(model, train_loader, test_loader, K: int, device, big_category: bool = False):
    """Train/test kNN graphs with the current model (csa_training.py:136-163).  Small categories score every test/train
    shape against every train shape; big ones only against k-means centre shapes, whose candidate-relative ids are
    mapped back to dataset ids (:141-155)."""
    train_f = ssa_features(model, train_loader, device)
    test_f = ssa_features(model, test_loader, device)
    if big_category:
        centres = np.sort(np.asarray(model.get_center_shape_indices(train_loader)))
        cand = train_f[torch.from_numpy(centres).to(device)]
        remap = torch.from_numpy(centres).to(device)
        train_g = remap[knn_graph_from_features(model, train_f, cand, K)]
        test_g = remap[knn_graph_from_features(model, test_f, cand, K)]
    else:
        train_g = knn_graph_from_features(model, train_f, train_f, K)
        test_g = knn_graph_from_features(model, test_f, train_f, K)
    return train_g.cpu().numpy().astype(np.int64), test_g.cpu().numpy().astype(np.int64)


@torch.no_grad()
def update_knn_graphs_sharded(model, train_loader_local, test_loader_local, K: int, device, group=None):
    """update_knn_graphs with the shape collection sharded over the ranks of a torch.distributed job: every rank
    passes loaders over ITS shapes only (contiguous id ranges in rank order), computes their SSA features, and the train
    features are all-gathered as the candidate set; each rank scores its own rows.  Returns the whole (S_train, K+1) and
    (S_test, K+1) int64 tables on every rank — bit-identical to the single-process ones (small-category form; the k-means
    candidate subset of big categories is CPU code over global descriptors and stays single-process)."""
    from . import functional as CF
    from .sharding import knn_graph_sharded
    train_f = ssa_features(model, train_loader_local, device)
    test_f = ssa_features(model, test_loader_local, device)
    train_g = knn_graph_sharded(train_f, K, CF.retrieval_measure, group=group)
    test_g = knn_graph_sharded(test_f, K, CF.retrieval_measure, cand_local=train_f, group=group)
    return train_g.cpu().numpy().astype(np.int64), test_g.cpu().numpy().astype(np.int64)
