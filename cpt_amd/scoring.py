"""Score extraction and region selection and IoU accuracy.

Counterparts of /root/reference/Oscar/oscar/zeroshot/refcoco_cpt.py:219-254 (raw colour logits),
Oscar/oscar/fewshot/refcoco_cpt.py:277-295 (colour logit / "none" logit) and
Oscar/oscar/utils/iou.py:1-12.  Host-side: the inputs are the (B, V) [MASK]-row logits the HIP path
returns; only B x n_colour values are touched, and torch.argmax's first-max tie-break is kept.
"""
import torch


def gather_color_scores(mask_scores, color_id_sets, none_id, divide_by_none=False):
    """mask_scores (P, V): one row per proposal sequence of ONE query; color_id_sets: per sequence
    the colour-token ids of its painted proposals.  Returns the concatenated score vector."""
    out = []
    for row, ids in zip(mask_scores, color_id_sets):
        cur = row[list(ids) + [none_id]]
        out.append(cur[0:-1] / cur[-1] if divide_by_none else cur[0:-1])
    return torch.cat(out, -1)


def select_region(mask_scores, color_id_sets, rect_sets, none_id, few_shot=False):
    """-> (max_idx, chosen rect, scores).  zeroshot/refcoco_cpt.py:224-246; few_shot=True uses the
    ratio to the "none" logit (fewshot/refcoco_cpt.py:291)."""
    scores = gather_color_scores(mask_scores, color_id_sets, none_id, divide_by_none=few_shot)
    rects = [r for rs in rect_sets for r in rs]
    idx = int(scores.argmax())
    return idx, rects[idx], scores


def compute_iou(box1, box2):
    """Boxes [x, y, w, h] with inclusive pixel extents (Oscar/oscar/utils/iou.py)."""
    ix1, iy1 = max(box1[0], box2[0]), max(box1[1], box2[1])
    ix2 = min(box1[0] + box1[2] - 1, box2[0] + box2[2] - 1)
    iy2 = min(box1[1] + box1[3] - 1, box2[1] + box2[3] - 1)
    inter = (ix2 - ix1 + 1) * (iy2 - iy1 + 1) if (ix1 < ix2 and iy1 < iy2) else 0
    return float(inter) / (box1[2] * box1[3] + box2[2] * box2[3] - inter)


def accuracy(predictions, gts, thresh=0.5):
    """predictions {key: [x1,y1,x2,y2]}, gts {key: [x,y,w,h]} -> percent with IoU > thresh
    (zeroshot/refcoco_cpt.py:267-280: predicted xyxy is converted with +1 extents)."""
    hit = 0
    for k, p in predictions.items():
        assert p[2] > p[0] and p[3] > p[1]
        q = [p[0], p[1], p[2] - p[0] + 1, p[3] - p[1] + 1]
        hit += compute_iou(q, gts[k]) > thresh
    return hit / max(len(predictions), 1) * 100


def nsp_choice_labels(labels, interval, n_seq, device=None):
    """fewshot/vcr_nsp_cpt.py:433-436: class 0 for the correct answer choice of each question, 1 elsewhere."""
    cls_labels = torch.ones([n_seq], dtype=torch.long, device=device)
    for i, lb in enumerate(labels):
        cls_labels[i * interval + int(lb)] = 0
    return cls_labels


def nsp_choose(rel_scores, interval):
    """fewshot/vcr_nsp_cpt.py:597-604: choice score = 1 - softmax(rel)[:, 1]; first-max argmax inside each
    question's `interval` answer choices.  Returns (scores (N,), [pred per question])."""
    logits = 1 - (rel_scores[:, :].softmax(-1)[:, 1].view(-1))
    n = rel_scores.size(0) // interval
    return logits, [int(logits[q * interval:(q + 1) * interval].argmax()) for q in range(n)]


def select_regions_device(mask_scores, color_id_sets, query_first, none_id, few_shot=False, return_scores=False):
    """Device-side region selection for a whole batch of queries (include/cpt_hip.h cpt_select_regions): the
    (S, V) [MASK]-row logits stay on the GPU and only one index per query comes back.
    mask_scores (S, V) CUDA fp32; color_id_sets: per sequence its colour-token ids; query_first: first sequence
    index of every query plus the total (len Q+1).  Same result as ``select_region`` per query
    (zeroshot/refcoco_cpt.py:224-246, fewshot/refcoco_cpt.py:277-295), torch.argmax tie-break included."""
    from . import _lib as L
    if not mask_scores.is_cuda:
        raise RuntimeError("cpt_amd: select_regions_device needs the logits on the GPU (no CPU fallback)")
    dev = mask_scores.device
    S, V = mask_scores.shape
    C = max(1, max((len(c) for c in color_id_sets), default=1))
    ids = torch.full((S, C), -1, dtype=torch.int64)
    for i, c in enumerate(color_id_sets):
        if len(c):
            ids[i, :len(c)] = torch.as_tensor(list(c), dtype=torch.int64)
    ids = ids.to(dev)
    qf = torch.as_tensor(list(query_first), dtype=torch.int32, device=dev)
    Q = qf.numel() - 1
    out = torch.empty(Q, dtype=torch.int64, device=dev)
    sc = torch.empty(Q, dtype=torch.float32, device=dev) if return_scores else None
    ms = mask_scores.contiguous()
    L.check(L.lib().cpt_select_regions(ms.data_ptr(), V, ids.data_ptr(), C, qf.data_ptr(), Q, int(none_id), 1 if few_shot else 0,
                                       out.data_ptr(), L.ptr(sc), L.stream_ptr()), "cpt_select_regions")
    return (out, sc) if return_scores else out


def argmax_columns_device(scores, ids):
    """GQA answer scoring (gqa_cpt.py:598-601): argmax over the label-token columns, on the device."""
    from . import _lib as L
    if not scores.is_cuda:
        raise RuntimeError("cpt_amd: argmax_columns_device needs the logits on the GPU (no CPU fallback)")
    R, V = scores.shape
    idt = torch.as_tensor(ids, dtype=torch.int64, device=scores.device).contiguous()
    out = torch.empty(R, dtype=torch.int64, device=scores.device)
    sc = scores.contiguous()
    L.check(L.lib().cpt_argmax_columns(sc.data_ptr(), V, idt.data_ptr(), idt.numel(), R, out.data_ptr(), None, L.stream_ptr()),
            "cpt_argmax_columns")
    return out
