"""TEST INFRASTRUCTURE (oracle side, BUILD-CONTAINER ONLY): golden vectors for the prompt assembly of the GQA and VCR few-shot drivers, produced by the
reference's OWN data-set classes -- GQADataset.tensorize_example (/root/reference/Oscar/oscar/fewshot/gqa_cpt.py:109-267) and
VCRDataset.tensorize_example (fewshot/vcr_nsp_cpt.py:141-311) -- on small rows written here.  The objects are built without their __init__ (which
wants the GQA / VCR annotation trees and label pickles on disk); everything tensorize_example touches is set by hand.  The drivers import
pytorch_transformers names that oracle/ref_stub.py does not model (tokenizer, optimizer, schedules): they only have to exist for the import.

    python oracle/make_golden_gqa_vcr.py        # writes tests/golden/tiny_gqa_vcr_prompts.npz + tiny_gqa_rows.tsv + tiny_vcr_rows.tsv"""
import base64
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
OUT = os.path.join(ROOT, "tests", "golden")

import ref_stub  # noqa: E402

pkg = ref_stub.install()
from transformers import BertTokenizer  # noqa: E402

pkg.WEIGHTS_NAME = "pytorch_model.bin"
pkg.BertTokenizer = BertTokenizer
for _n in ("AdamW", "WarmupLinearSchedule", "WarmupConstantSchedule"):
    setattr(pkg, _n, type(_n, (), {}))
sys.path.insert(0, "/root/reference/Oscar")
import oscar.fewshot.gqa_cpt as G  # noqa: E402
import oscar.fewshot.vcr_nsp_cpt as V  # noqa: E402
from oscar.utils.task_utils import InputInstance  # noqa: E402
from oscar.utils.tsv_file import TSVFile  # noqa: E402

VOCAB = (["[PAD]"] + ["[unused%d]" % i for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] +
         ["the", "a", "is", "what", "color", "of", "dog", "cat", "man", "woman", "person", "on", "in", "left", "right", "red", "blue", "green", "yellow",
          "?", ".", ",", "table", "chair", "holding", "cup", "why", "because", "she", "he", "they", "are", "and", "to", "sitting", "standing", "next",
          "looking", "at", "##s", "##ing", "1", "2", "3", "frisbee", "wants", "it", "yes", "no"])


def tokenizer(tmp):
    vf = os.path.join(tmp, "vocab.txt")
    with open(vf, "w") as f:
        f.write("\n".join(VOCAB) + "\n")
    return BertTokenizer(vf, do_lower_case=True)


def boxes(rng, n):
    return [{"rect": [float(v) for v in rng.integers(0, 200, 4)], "class": ["dog", "cat", "person", "table"][j % 4], "conf": 0.5,
             "feature": base64.b64encode(rng.standard_normal(2054).astype(np.float32).tobytes()).decode("utf-8")} for j in range(n)]


def main():
    rng = np.random.Generator(np.random.PCG64(606))
    tmp = tempfile.mkdtemp()
    tok = tokenizer(tmp)
    g = {"vocab": np.array(VOCAB)}
    args = types.SimpleNamespace(max_seq_length=40, max_img_seq_length=6, output_mode="classification", model_type="bert", load_fast=False,
                                 img_feature_type="faster_r-cnn")
    # ---- GQA: two colour-feature rows (question with colour words spliced in) + one plain row (question as it is) + one long question (truncation)
    gq = [("101", "what is the man holding?", [[[12], "red"]], [3], 7),                                   # one painted object, 7 regions (> 6: cut)
          ("102", "is the dog on the left of the cat?", [[[7], "blue"], [[30], "green"]], [1, 5], 3),
          ("103", "what color is the cup on the table?", None, [], 4),                                   # no colour row: plain features, empty label list
          ("104", "the man and the woman are sitting next to the table and the chair and the dog and the cat and the cup and the frisbee " * 2 + "?",
           [[[4], "yellow"]], None, 2)]
    color_rows, plain_rows, examples = [], [], []
    for qid, q, pc, lab, nbox in gq:
        bl = boxes(rng, nbox)
        if pc is not None:
            color_rows.append((qid, json.dumps({"objects": [bl, [pc]]})))
        plain_rows.append(("img" + qid, json.dumps({"objects": bl})))
        examples.append(InputInstance(guid=qid, text_a=q, text_b=None, label=lab, score=[1.0] * len(lab or []), img_key="img" + qid, q_id=int(qid)))
    ctsv, ptsv = os.path.join(OUT, "tiny_gqa_color_rows.tsv"), os.path.join(OUT, "tiny_gqa_rows.tsv")
    for path, rows in ((ctsv, color_rows), (ptsv, plain_rows)):
        with open(path, "w") as f:
            for k, v in rows:
                f.write(k + "\t" + v + "\n")
    ds = object.__new__(G.GQADataset)
    ds.args, ds.tokenizer, ds.labels = args, tok, list(range(9))
    ds.color_img_feat_tsv, ds.img_feat_tsv = TSVFile(ctsv, generate_lineidx=True), TSVFile(ptsv, generate_lineidx=True)
    # (the reference looks a colour row up with `if color_idx:` -- row 0 of the colour file is therefore never used; index 0 holds a spare copy)
    ds.qid2feat = {r[0]: i for i, r in enumerate(color_rows)}
    ds.imgid2feat = {r[0]: i for i, r in enumerate(plain_rows)}
    g["gqa_n"] = np.array(len(examples))
    for i, ex in enumerate(examples):
        ids, msk, seg, lab0, target, feat, qid, mpos = ds.tensorize_example(ex, cls_token=tok.cls_token, sep_token=tok.sep_token, cls_token_segment_id=0,
                                                                           pad_token_segment_id=0)
        k = "gqa%d_" % i
        cidx = ds.qid2feat.get(str(ex.q_id), None)
        g[k + "in"] = np.array(json.dumps({"question": gq[i][1], "pc": gq[i][2] if cidx else None, "label": gq[i][3], "n_regions": gq[i][4], "q_id": gq[i][0],
                                           "color_row_used": bool(cidx)}))
        g[k + "ids"], g[k + "mask"], g[k + "seg"] = ids.numpy(), msk.numpy(), seg.numpy()
        g[k + "label0"], g[k + "target"], g[k + "qid"], g[k + "mpos"] = lab0.numpy(), target.numpy(), qid.numpy(), np.array(mpos)
        g[k + "feat_rows"] = np.array(feat.shape[0])
    # ---- VCR: one row, four choices with object references
    colors, names = {"0": "red", "1_2": "blue"}, {"0": "person", "1_2": "dog and cat", "3": "table", "1": "dog"}
    vrow = ("77_0", json.dumps({"objects": [boxes(rng, 5), [colors, names]]}))
    vtsv = os.path.join(OUT, "tiny_vcr_rows.tsv")
    with open(vtsv, "w") as f:
        f.write(vrow[0] + "\t" + vrow[1] + "\n")
    question = ["why", "is", [0], "looking", "at", [2, 1], "?"]
    choices = [["because", [0], "wants", "the", "frisbee", "."], [[3], "is", "next", "to", [1], "."], ["she", "is", "holding", "a", "cup", "."],
               [[0], "and", [1, 2], "are", "sitting", "on", "the", "chair", "and", "the", "table", "and", "looking", "at", "the", "cup"] * 3]
    ex = InputInstance(guid="77", text_a=question, text_b=choices, label=2, img_key="77_0", q_id=5)
    vs = object.__new__(V.VCRDataset)
    vs.args, vs.tokenizer = args, tok
    vs.feat_tsv = TSVFile(vtsv, generate_lineidx=True)
    vs.imgid2feat = {"77_0": 0}
    out = vs.tensorize_example(ex, cls_token=tok.cls_token, sep_token=tok.sep_token, cls_token_segment_id=0, pad_token_segment_id=0)
    g["vcr_in"] = np.array(json.dumps({"question": question, "choices": choices, "colors": colors, "names": names, "n_regions": 5, "label": 2, "q_id": 5}))
    g["vcr_ids"] = np.stack([o[0].numpy() for o in out])
    g["vcr_mask"] = np.stack([o[1].numpy() for o in out])
    g["vcr_seg"] = np.stack([o[2].numpy() for o in out])
    g["vcr_qid"], g["vcr_label"] = np.array([o[4] for o in out]), np.array([o[5] for o in out])
    g["vcr_mpos"] = np.array(json.dumps([o[6] for o in out]))
    g["vcr_text_a"] = np.array(vs._vcr_textize(question, colors, names))
    g["vcr_texts_b"] = np.array([vs._vcr_textize(c, colors, names, colorful=True) for c in choices])
    g["args"] = np.array(json.dumps({"max_seq_length": args.max_seq_length, "max_img_seq_length": args.max_img_seq_length}))
    np.savez_compressed(os.path.join(OUT, "tiny_gqa_vcr_prompts.npz"), **g)
    for f in (ctsv, ptsv, vtsv):          # the rows were only inputs of the reference classes here; the arrays above are the fixture
        os.remove(f)
        if os.path.exists(f.replace(".tsv", ".lineidx")):
            os.remove(f.replace(".tsv", ".lineidx"))
    print("wrote", os.path.join(OUT, "tiny_gqa_vcr_prompts.npz"))


if __name__ == "__main__":
    main()
