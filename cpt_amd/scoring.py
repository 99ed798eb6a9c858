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
