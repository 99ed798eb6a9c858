"""Prompt assembly for the RefCOCO colour-prompt data sets, host side; at the end of the file, round 6: the same for the
GQA and VCR few-shot drivers (gqa_cpt.py:109-267, vcr_nsp_cpt.py:141-311).

Counterpart of /root/reference/Oscar/oscar/datasets/refcoco_zsl_cpt_dataset.py:
  * ``TEMPLATES``            -- template1..6 (:18-54): where the ``[MASK]`` colour slot goes in the caption;
  * ``tokenize``             -- tokenize() (:211-302): [CLS] a [SEP] b [SEP], longest-first truncation to 70, zero padding,
                                the attention mask over 70 text + img_seq_len region slots, label grid of -1;
  * ``assemble_query``       -- ZSLColorFinetuneDataset.__getitem__ (:85-159): caption clean-up, the colour name spliced
                                into the proposal's own detection label, one prompted sequence per proposal, the [MASK]
                                position, ground-truth colour ids by IoU > 0.5 against the annotated box, and the
                                positive/negative sampling of the few-shot split.
What changes against the reference: the per-sequence tensors come back STACKED as int64 arrays (P, 70) / (P, 70 + img_seq_len)
ready for one H2D copy; the region features are not touched here (cpt_amd.io decodes them straight into pinned memory) --
only their per-proposal counts are needed for the mask.  Tokenisation itself is the caller's tokenizer object
(``tokenize(text) -> tokens``, ``convert_tokens_to_ids``), the reference's BertTokenizer in practice; the detection-label
text is the same for every proposal of an image except the spliced colour word, so its tokens are cached per label.
"""
import random as _random

import numpy as np

from .scoring import compute_iou

MASK_ID = 103           # the reference finds the slot with input_ids.index(103) (:116): [MASK] of bert-base-uncased


def template1(caption, posi_tokens):
    return caption + " is [MASK]."


def template2(caption, posi_tokens):
    return caption + " is [MASK] color."


def template3(caption, posi_tokens):
    return caption + " is in [MASK] color."


def template4(caption, posi_tokens):
    p = posi_tokens[-1]
    return caption[:p] + " in [MASK]." if p == len(caption) else caption[:p] + " in [MASK]" + caption[p:] + "."


def template5(caption, posi_tokens):
    p = posi_tokens[-1]
    return caption[:p] + " in [MASK] color." if p == len(caption) else caption[:p] + " in [MASK] color" + caption[p:] + "."


def template6(caption, posi_tokens):
    p = posi_tokens[0]
    return caption[:p] + "[MASK] " + caption[p:] + "."


TEMPLATES = {1: template1, 2: template2, 3: template3, 4: template4, 5: template5, 6: template6}


def truncate_seq_pair(tokens_a, tokens_b, max_length):
    """_truncate_seq_pair (:188-208): pop from the longer list until the pair fits."""
    while len(tokens_a) + len(tokens_b) > max_length:
        if len(tokens_a) > len(tokens_b):
            tokens_a.pop()
        else:
            tokens_b.pop()


def tokenize(tokenizer, text_a, text_b, n_img_feats, max_img_seq_len=50, max_seq_a_len=40, max_seq_len=70,
             cls_token_segment_id=0, pad_token_segment_id=0, sequence_a_segment_id=0, sequence_b_segment_id=1,
             tokens_b=None):
    """tokenize() of the reference (:211-302) -> (input_ids (max_seq_len,), input_mask (max_seq_len + max_img_seq_len,),
    segment_ids (max_seq_len,), lm_label_ids (max_seq_len + max_img_seq_len,)) as int64 numpy arrays.  ``n_img_feats`` is
    img_feat.shape[0]; ``tokens_b`` may carry text_b already tokenised (it is copied before truncation)."""
    tokens_a = tokenizer.tokenize(text_a)
    if tokens_b is not None:
        tokens_b = list(tokens_b)
    elif text_b:
        tokens_b = tokenizer.tokenize(text_b)
    if tokens_b:
        truncate_seq_pair(tokens_a, tokens_b, max_seq_len - 3)
    elif len(tokens_a) > max_seq_len - 2:
        tokens_a = tokens_a[:max_seq_len - 2]
    tokens = ["[CLS]"] + tokens_a + ["[SEP]"]
    n_a = len(tokens)
    if tokens_b:
        tokens = tokens + tokens_b + ["[SEP]"]
    ids = tokenizer.convert_tokens_to_ids(tokens)
    n = len(ids)
    if n > max_seq_len:
        raise AssertionError("sequence of %d tokens exceeds max_seq_len %d" % (n, max_seq_len))     # the reference asserts (:281)
    input_ids = np.zeros(max_seq_len, dtype=np.int64)
    input_ids[:n] = ids
    segment_ids = np.zeros(max_seq_len, dtype=np.int64)
    segment_ids[:n_a] = sequence_a_segment_id
    segment_ids[0] = cls_token_segment_id
    segment_ids[n_a:n] = sequence_b_segment_id
    segment_ids[n:] = pad_token_segment_id
    input_mask = np.zeros(max_seq_len + max(max_img_seq_len, 0), dtype=np.int64)
    input_mask[:n] = 1
    if max_img_seq_len > 0:
        # (:286-293) every region gets a 1 -- also beyond max_img_seq_len, where the reference's mask grows longer than the
        # padded feature tensor and the model call fails downstream; the mirror refuses that case up front
        if n_img_feats > max_img_seq_len:
            raise ValueError("%d regions exceed img_seq_len %d" % (n_img_feats, max_img_seq_len))
        input_mask[max_seq_len:max_seq_len + n_img_feats] = 1
    lm_label_ids = np.full(max_seq_len + max(max_img_seq_len, 0), -1, dtype=np.int64)
    return input_ids, input_mask, segment_ids, lm_label_ids


def coloured_od_labels(od_labels, cname, n_proposals):
    """(:98-99) proposal i's text_b: the image's detection labels with the colour name in front of label i."""
    return [" ".join([cname + " " + x if i == j else x for j, x in enumerate(od_labels)]) for i in range(n_proposals)]


def ground_truths(tokenizer, gt_bbox, colors, rects):
    """(:122-134) per proposal: the colour of the box that overlaps the annotated box best if IoU > 0.5, else "none"."""
    gts = []
    for color_set, boxes in zip(colors, rects):
        ious = [compute_iou(gt_bbox, [p[0], p[1], p[2] - p[0] + 1, p[3] - p[1] + 1]) for p in boxes]
        k = int(np.argmax(ious))
        assert len(color_set) == len(boxes)
        gts.append(color_set[k] if ious[k] > 0.5 else "none")
    return [tokenizer.convert_tokens_to_ids(c) for c in gts]


def sample_train(gts, na_id, n_items, rng=_random):
    """Few-shot sampling of a row's proposals (the rule of refcoco_zsl_cpt_dataset.py:138-155): keep ONE proposal whose ground truth
    is a colour (all of them when the data set has at most 8 rows; proposal 0 when there is none) and no more "none" proposals
    than colour ones.  Consumes the generator exactly like the reference -- one ``shuffle`` of the colour list when it is cut,
    then one of the "none" list when it is cut -- so a seeded run selects the same proposals (tests/golden/tiny_prompts.npz)."""
    coloured, plain = [], []
    for i, g in enumerate(gts):
        (plain if g == na_id else coloured).append(i)
    if not coloured:
        coloured = [0]
    elif len(coloured) > 1 and n_items > 8:
        rng.shuffle(coloured)
        del coloured[1:]
    if len(plain) > len(coloured):
        rng.shuffle(plain)
        del plain[len(coloured):]
    return coloured + plain


class PromptBuilder(object):
    """ZSLColorFinetuneDataset.__getitem__ without the feature decode: one call per TSV row.

    anns_dic: {str(image id): annotation with "bbox"} (the finetune_*.json split file, :70-71);
    det_dic: {image name: [detection label, ...]} (mydetections/*/dets.json, :76); template: key of TEMPLATES."""

    def __init__(self, tokenizer, anns_dic, det_dic, template=1, txt_seq_len=70, img_seq_len=50, is_train=False, n_items=0,
                 rng=_random):
        self.tokenizer, self.anns_dic, self.det_dic = tokenizer, anns_dic, det_dic
        self.template = TEMPLATES[template] if not callable(template) else template
        self.txt_seq_len, self.img_seq_len, self.is_train, self.n_items, self.rng = txt_seq_len, img_seq_len, is_train, n_items, rng
        self._label_tokens = {}

    def _tok(self, word):
        t = self._label_tokens.get(word)
        if t is None:
            t = self._label_tokens[word] = self.tokenizer.tokenize(word)
        return t

    def __call__(self, img_name, info, region_counts):
        """info: the parsed row ({"objects": [objs, caption, colors, rects]}; feature strings may be stripped);
        region_counts: regions per proposal.  Returns a dict of stacked int64 arrays + the row's colours / rectangles and
        ``keep`` (proposal indices kept, all of them unless is_train)."""
        objs, caption, colors, rects = info["objects"]
        P = len(region_counts)
        caption = caption.replace(".", "").strip()
        posi_token = 0
        od_labels = self.det_dic[img_name]
        cname = colors[0][0]
        text_a = self.template(caption, posi_token)
        # text_b of proposal i = labels with the colour word before label i: tokens of the words are cached, and a word's
        # tokens do not depend on its neighbours for a whitespace-splitting WordPiece tokenizer (BertTokenizer)
        base = [self._tok(x) for x in od_labels]
        ctoks = self._tok(cname)
        input_ids = np.zeros((P, self.txt_seq_len), dtype=np.int64)
        segment_ids = np.zeros((P, self.txt_seq_len), dtype=np.int64)
        input_mask = np.zeros((P, self.txt_seq_len + self.img_seq_len), dtype=np.int64)
        mask_pos = np.zeros(P, dtype=np.int64)
        for i in range(P):
            tb = []
            for j, t in enumerate(base):
                if i == j:
                    tb += ctoks
                tb += t
            assert isinstance(colors[i][0], str)
            ids, msk, seg, _ = tokenize(self.tokenizer, text_a, None, region_counts[i], max_img_seq_len=self.img_seq_len,
                                        max_seq_a_len=40, max_seq_len=self.txt_seq_len, tokens_b=tb)
            input_ids[i], input_mask[i], segment_ids[i] = ids, msk, seg
            hit = np.nonzero(ids == MASK_ID)[0]
            if hit.size == 0:
                raise ValueError("103 is not in list")          # what list.index raises in the reference (:116)
            mask_pos[i] = hit[0]
        gts = ground_truths(self.tokenizer, self.anns_dic[str(img_name)]["bbox"], colors, rects)
        keep = list(range(P))
        if self.is_train:
            keep = sample_train(gts, self.tokenizer.convert_tokens_to_ids("none"), self.n_items, self.rng)
        k = np.asarray(keep, dtype=np.int64)
        return {"img_name": img_name, "input_ids": input_ids[k], "input_mask": input_mask[k], "segment_ids": segment_ids[k],
                "mask_token_pos": mask_pos[k], "gts": [gts[i] for i in keep], "colors": colors, "rects": rects, "keep": keep}


# ---- GQA and VCR few-shot drivers (round 6; SURVEY 8(f).2 cites the RefCOCO data sets only -- these are the next callers out) ----------------------
# Counterparts of GQADataset.tensorize_example / get_img_feature (/root/reference/Oscar/oscar/fewshot/gqa_cpt.py:109-193, 231-267) and of
# VCRDataset._vcr_textize / tensorize_example / _tensorize (fewshot/vcr_nsp_cpt.py:156-311).  As above: the region features stay with cpt_amd.io (only
# their count enters the mask), the tokenizer object is the caller's, arrays come back as int64 numpy, one row per sequence.

def tokenize_pair(tokenizer, text_a, text_b, n_img_feats, max_seq_length=165, max_img_seq_length=45, cls_token="[CLS]", sep_token="[SEP]",
                  pad_token=0, sequence_a_segment_id=0, sequence_b_segment_id=1, cls_token_segment_id=0, pad_token_segment_id=0,
                  split_b_on_semicolons=False):
    """The text side both drivers share (gqa_cpt.py:117-160, vcr_nsp_cpt.py:205-262; the BERT branch: [CLS] in front, padding on the right):
    tokens = [CLS] a [SEP] (b [SEP]); the pair is cut longest-first to max_seq_length - 3 (a alone to max_seq_length - 2); segment ids 0 / 1 with the
    [CLS] slot's own id; the attention mask covers the text and then one slot per region, zero-padded to max_img_seq_length (regions beyond it are
    cut, as the drivers cut the feature rows: gqa_cpt.py:163-167).  ``split_b_on_semicolons``: VCR's text_b handling (';' separates object-label
    groups; they are joined by blanks before tokenising, vcr_nsp_cpt.py:213-222).
    Returns (input_ids (max_seq_length,), input_mask (max_seq_length + max_img_seq_length,), segment_ids (max_seq_length,), mask_positions [list of the
    positions holding id 103])."""
    tokens_a = tokenizer.tokenize(text_a)
    tokens_b = None
    if text_b:
        tb = text_b.replace(";", " ").strip() if split_b_on_semicolons else text_b
        tokens_b = tokenizer.tokenize(tb)
        truncate_seq_pair(tokens_a, tokens_b, max_seq_length - 3)
    elif len(tokens_a) > max_seq_length - 2:
        tokens_a = tokens_a[:max_seq_length - 2]
    tokens = tokens_a + [sep_token]
    seg = [sequence_a_segment_id] * len(tokens)
    if tokens_b:
        tokens += tokens_b + [sep_token]
        seg += [sequence_b_segment_id] * (len(tokens_b) + 1)
    tokens = [cls_token] + tokens
    seg = [cls_token_segment_id] + seg
    ids = tokenizer.convert_tokens_to_ids(tokens)
    n = len(ids)
    if n > max_seq_length:
        raise AssertionError("sequence of %d tokens exceeds max_seq_length %d" % (n, max_seq_length))
    input_ids = np.full(max_seq_length, pad_token, dtype=np.int64)
    input_ids[:n] = ids
    segment_ids = np.full(max_seq_length, pad_token_segment_id, dtype=np.int64)
    segment_ids[:n] = seg
    L_img = max(max_img_seq_length, 0)
    input_mask = np.zeros(max_seq_length + L_img, dtype=np.int64)
    input_mask[:n] = 1
    if L_img > 0:
        input_mask[max_seq_length:max_seq_length + min(int(n_img_feats), L_img)] = 1
    return input_ids, input_mask, segment_ids, [int(p) for p in np.nonzero(input_ids == MASK_ID)[0]]


def gqa_question(question, positions_and_colors):
    """gqa_cpt.py:237-249: the colour words of the painted objects spliced into the question at their character positions
    (meta_infos[0] of the colour-feature row: [[[position, ...], colour], ...])."""
    positions = [0] + [x[0][0] for x in positions_and_colors]
    colors = [x[1] for x in positions_and_colors]
    parts = []
    for i in range(len(positions) - 1):
        parts.append(question[positions[i]:positions[i + 1]])
        parts.append(colors[i] + " ")
    parts.append(question[positions[-1]:])
    return "".join(parts)


def assemble_gqa(tokenizer, question, label, n_labels, n_regions, positions_and_colors=None, q_id=0, max_seq_length=165, max_img_seq_length=45):
    """GQADataset.tensorize_example (gqa_cpt.py:109-193) for one question: text_a = the question (with the colour words spliced in when the
    colour-feature row carries them), text_b = "[MASK]"; label ids (None / [] -> [0]) and the n_labels-wide 0/1 target (target_tensor, :765-771)."""
    text_a = gqa_question(question, positions_and_colors) if positions_and_colors is not None else question
    ids, msk, seg, mpos = tokenize_pair(tokenizer, text_a, "[MASK]", n_regions, max_seq_length, max_img_seq_length)
    label_id = [0] if (label is None or len(label) == 0) else [int(l) for l in label]
    target = np.zeros(n_labels, dtype=np.float32)
    for l in label_id:
        target[l] = 1.0
    return {"text_a": text_a, "input_ids": ids, "input_mask": msk, "segment_ids": seg, "label_id": np.array([label_id[0]], dtype=np.int64),
            "target": target, "q_id": np.array([int(q_id)], dtype=np.int64), "mask_token_pos": mpos}


def vcr_textize(sentence, colors, names, colorful=True):
    """VCRDataset._vcr_textize (vcr_nsp_cpt.py:156-166): words stay, object references (lists of box indices) become the objects' names -- with
    " in <colour>" behind a painted one when ``colorful``; colors / names are keyed by the sorted indices joined with '_'."""
    def word(w):
        k = "_".join(str(y) for y in sorted(w))
        if k in colors and colorful:
            return names[k] + " in {}".format(colors[k])
        return names[k]
    return " ".join(word(w) if type(w) is list else w for w in sentence)


def assemble_vcr(tokenizer, question, choices, colors, names, n_regions, label=None, q_id=0, max_seq_length=165, max_img_seq_length=45):
    """VCRDataset.tensorize_example + _tensorize (vcr_nsp_cpt.py:168-311): one sequence per answer choice -- text_a = the question, text_b = the
    choice, both with the objects' names (and colours) written out.  Stacked (n_choices, ...) int64 arrays."""
    text_a = vcr_textize(question, colors, names)
    ids, msk, seg, mpos, texts_b = [], [], [], [], []
    for c in choices:
        tb = vcr_textize(c, colors, names, colorful=True)
        i, m, s, p = tokenize_pair(tokenizer, text_a, tb, n_regions, max_seq_length, max_img_seq_length, split_b_on_semicolons=True)
        ids.append(i); msk.append(m); seg.append(s); mpos.append(p); texts_b.append(tb)
    return {"text_a": text_a, "texts_b": texts_b, "input_ids": np.stack(ids), "input_mask": np.stack(msk), "segment_ids": np.stack(seg),
            "mask_token_pos": mpos, "label": [0] if label is None else label, "q_id": int(q_id)}
