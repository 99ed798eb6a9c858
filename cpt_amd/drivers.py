"""Driver-side hot loops on the HIP path: counterparts of the reference's ``val`` and ``train_batch``.

  * ``val_queries``  <- /root/reference/Oscar/oscar/zeroshot/refcoco_cpt.py:208-288 (and the few-shot
    variant fewshot/refcoco_cpt.py:258-315): per query, score every proposal sequence at its [MASK]
    slot, gather the colour logits, argmax -> chosen rectangle; queries are sharded across ranks and
    the chosen indices are gathered with one fixed-shape all_gather.
  * ``train_batch``  <- fewshot/refcoco_cpt.py:225-255: label grid, LR schedule, zero_grad / loss /
    backward / step, RuntimeError per step logged and skipped.
Datasets, tokeniser and TSV decoding stay outside: a *query* here is already
tensors, in the layout ``test_collate`` (zeroshot/refcoco_cpt.py:159-172) produces.
"""
import logging

import torch

from . import dist as cdist
from . import scoring
from .train import get_lr_sched

logger = logging.getLogger(__name__)


def val_queries(model, queries, none_id, device, few_shot=False, batch_queries=4, colour_columns_only=True):
    """queries: list of dicts with tensors ``img_feats (P,Li,D)``, ``input_ids (P,Lt)``,
    ``segment_ids``, ``attention_mask (P,Lt+Li)``, ``mask_token_pos (P,)`` and python lists
    ``colors`` (per proposal sequence: colour-token ids) and ``rects`` (per sequence: rectangles).
    Returns {global query index: (max_idx, rect)} on every rank.
    colour_columns_only (default): the decoder scores only the vocabulary columns the selection reads -- every colour id of the queries + ``none_id``
    (zeroshot/refcoco_cpt.py:219 gathers exactly those out of the (P, V) scores) -- instead of all V; same chosen indices."""
    model.eval()
    rank, world = cdist.rank_world()
    lo, hi = cdist.shard_range(len(queries), rank, world)
    chosen = torch.full((hi - lo,), -1, dtype=torch.int64, device=device)
    col_ids, col_of, cols_dev = None, None, None
    if colour_columns_only:
        col_ids = sorted({int(c) for q in queries for s in q["colors"] for c in s} | {int(none_id)})
        col_of = {c: i for i, c in enumerate(col_ids)}
        cols_dev = torch.tensor(col_ids, dtype=torch.int64, device=device)
    for start in range(lo, hi, batch_queries):
        chunk = queries[start:min(start + batch_queries, hi)]
        cat = {k: torch.cat([q[k] for q in chunk], 0).to(device, non_blocking=True)
               for k in ("img_feats", "input_ids", "segment_ids", "attention_mask", "mask_token_pos")}
        with torch.no_grad():
            if cols_dev is not None:
                scores = model(cat["input_ids"], cat["segment_ids"], cat["attention_mask"], img_feats=cat["img_feats"],
                               mask_token_pos=cat["mask_token_pos"], vocab_columns=cols_dev)[0]
            else:
                scores = model(cat["input_ids"], cat["segment_ids"], cat["attention_mask"], img_feats=cat["img_feats"],
                               mask_token_pos=cat["mask_token_pos"])[0]
        # colour gather + per-query argmax run on the device (cpt_select_regions): only indices leave the GPU
        sets = [list(s) for q in chunk for s in q["colors"]]
        if col_of is not None:      # (ids -> positions in the column list the scores were computed for)
            sets = [[col_of[int(c)] for c in s] for s in sets]
        first = [0]
        for q in chunk:
            first.append(first[-1] + q["input_ids"].size(0))
        idx = scoring.select_regions_device(scores, sets, first, none_id if col_of is None else col_of[int(none_id)], few_shot=few_shot)
        chosen[start - lo:start - lo + len(chunk)] = idx
    allc = cdist.gather_fixed(chosen, len(queries), fill=-1).cpu().tolist()
    out = {}
    for gi, idx in enumerate(allc):
        rects = [r for rs in queries[gi]["rects"] for r in rs]
        # a query none of whose proposal sequences carries a colour has no prediction (idx -1): never index rects[-1]
        out[gi] = (idx, rects[idx] if 0 <= idx < len(rects) else None)
    return out


def _label_grid(batch):
    """(B, L) grid of -1 with each sequence's colour id at its [MASK] slot: what CrossEntropyLoss(ignore_index=-1) of
    modeling_rec.py:147-150 wants (fewshot/refcoco_cpt.py:231-233)."""
    grid = batch["attention_mask"].new_full(batch["attention_mask"].shape, -1, dtype=torch.long)
    # index assignment as the reference writes it: any integer dtype and negative (from-the-end) positions are accepted
    grid[torch.arange(grid.size(0), device=grid.device), batch["mask_token_pos"].long().view(-1)] = batch["colors"].long().view(-1)
    return grid


def _apply_lr(optimizer, lr, head_multiplier):
    """The reference's parameter groups (fewshot/refcoco_cpt.py:236-243, 318-343): groups 0-1 take lr x lr_mul, groups 2-3 the plain
    rate; like the reference, an optimizer with fewer groups trains and only a fifth group raises."""
    groups = optimizer.param_groups
    if len(groups) > 4:
        raise ValueError("at most the four parameter groups of build_optimizer are scheduled, got %d" % len(groups))
    for g in groups[:2]:
        g["lr"] = lr * head_multiplier
    for g in groups[2:]:
        g["lr"] = lr


def train_batch(model, optimizer, batches, opts, device, global_step=0):
    """One pass over ``batches`` (dicts with img_feats, input_ids, attention_mask, segment_ids, mask_token_pos, colors): the few-shot
    step of fewshot/refcoco_cpt.py:225-255 -- label grid, scheduled learning rate, zero_grad / forward / backward / step; a step that
    raises RuntimeError is logged and skipped without advancing the schedule.  Returns (global_step, [loss tensors])."""
    model.train()
    losses = []
    for index, host_batch in enumerate(batches):
        batch = {name: t.to(device) for name, t in host_batch.items()}
        _apply_lr(optimizer, get_lr_sched(global_step, opts), getattr(opts, "lr_mul", 1.0))
        try:
            optimizer.zero_grad()
            loss = model(batch["input_ids"], batch["segment_ids"], batch["attention_mask"], img_feats=batch["img_feats"],
                         masked_lm_labels=_label_grid(batch))[0]
            loss.backward()
            optimizer.step()
        except RuntimeError as err:
            logger.info("step %d of this pass skipped: %s", index, err)
            continue
        global_step += 1
        losses.append(loss.detach())
    return global_step, losses
