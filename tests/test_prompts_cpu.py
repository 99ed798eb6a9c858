"""CPU: prompt assembly (cpt_amd/prompts.py) against the fixture the reference's own templates, tokenize() and
ZSLColorFinetuneDataset.__getitem__ produced (oracle/make_golden.py prompt_case;
/root/reference/Oscar/oscar/datasets/refcoco_zsl_cpt_dataset.py:18-54, 85-159, 211-302), rows read through the C decoder."""
import json
import os
import random

import numpy as np
import pytest

from cpt_amd import prompts


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "tiny_prompts.npz"))


@pytest.fixture(scope="module")
def tok(gold, tmp_path_factory):
    transformers = pytest.importorskip("transformers")
    vf = tmp_path_factory.mktemp("vocab") / "vocab.txt"
    vf.write_text("\n".join(str(w) for w in gold["vocab"]) + "\n")
    return transformers.BertTokenizer(str(vf), do_lower_case=True)


def test_templates_and_tokenize_match_reference(gold, tok):
    i = 0
    while "tok%d_in" % i in gold.files:
        cap, tb, nf, t = [str(x) for x in gold["tok%d_in" % i]]
        text_a = prompts.TEMPLATES[int(t)](cap, 0)
        assert text_a == str(gold["tok%d_text_a" % i])
        ids, msk, seg, lab = prompts.tokenize(tok, text_a, tb, int(nf))
        for got, key in ((ids, "ids"), (msk, "mask"), (seg, "seg"), (lab, "lab")):
            assert got.dtype == np.int64 and np.array_equal(got, gold["tok%d_%s" % (i, key)]), (i, key)
        # the cached-token form the row builder uses gives the same sequence
        ids2 = prompts.tokenize(tok, text_a, None, int(nf), tokens_b=tok.tokenize(tb))[0]
        assert np.array_equal(ids2, ids)
        i += 1
    assert i >= 5
    for t in (4, 5, 6):
        assert [prompts.TEMPLATES[t]("man in red shirt", [3, 10]), prompts.TEMPLATES[t]("man in red", [10])] == [str(x) for x in gold["tmpl%d" % t]]
    with pytest.raises(ValueError):
        prompts.tokenize(tok, "the dog", "man", 51)          # more regions than slots: refused, not silently mis-masked


@pytest.mark.parametrize("is_train", [False, True])
def test_rows_match_reference_getitem(gold, tok, golden_dir, is_train):
    from cpt_amd import io
    tsv = io.TSVFile(os.path.join(golden_dir, "tiny_prompt_rows.tsv"))
    anns, dets = json.loads(str(gold["anns"])), json.loads(str(gold["dets"]))
    random.seed(1234)
    build = prompts.PromptBuilder(tok, anns, dets, template=2, img_seq_len=50, is_train=is_train, n_items=tsv.num_rows(), rng=random)
    for i in range(tsv.num_rows()):
        name, payload = tsv.seek_raw(i)
        info, feats, fmask, counts = io.decode_row(payload, img_seq_len=50)
        q = build(name.decode() if isinstance(name, bytes) else name, info, counts)
        k = "%s%d_" % ("tr" if is_train else "ev", i)
        assert np.array_equal(q["input_ids"], gold[k + "ids"]), k
        assert np.array_equal(q["input_mask"], gold[k + "mask"]), k
        assert np.array_equal(q["segment_ids"], gold[k + "seg"]), k
        assert np.array_equal(q["mask_token_pos"], gold[k + "mpos"]), k
        assert list(q["gts"]) == gold[k + "gts"].tolist(), k
        # the kept proposals' decoded features are the reference's (zero padded to img_seq_len): same sums, and the image
        # half of the attention mask equals the decoder's region mask
        kept = feats[q["keep"]]
        assert np.allclose(kept.double().sum((1, 2)).numpy(), gold[k + "feat_sum"], rtol=0, atol=1e-9)
        assert np.array_equal(q["input_mask"][:, 70:], fmask[q["keep"]].numpy())


# ---- GQA / VCR few-shot drivers (round 6): fixture from the reference's own GQADataset / VCRDataset.tensorize_example (oracle/make_golden_gqa_vcr.py) ----
@pytest.fixture(scope="module")
def gold2(golden_dir):
    return np.load(os.path.join(golden_dir, "tiny_gqa_vcr_prompts.npz"))


@pytest.fixture(scope="module")
def tok2(gold2, tmp_path_factory):
    transformers = pytest.importorskip("transformers")
    vf = tmp_path_factory.mktemp("vocab2") / "vocab.txt"
    vf.write_text("\n".join(str(w) for w in gold2["vocab"]) + "\n")
    return transformers.BertTokenizer(str(vf), do_lower_case=True)


def test_gqa_prompts_match_reference_tensorize_example(gold2, tok2):
    a = json.loads(str(gold2["args"]))
    n = int(gold2["gqa_n"])
    assert n >= 4
    spliced = 0
    for i in range(n):
        k = "gqa%d_" % i
        d = json.loads(str(gold2[k + "in"]))
        q = prompts.assemble_gqa(tok2, d["question"], d["label"], 9, d["n_regions"], positions_and_colors=d["pc"] if d["pc"] else None, q_id=d["q_id"],
                                 max_seq_length=a["max_seq_length"], max_img_seq_length=a["max_img_seq_length"])
        spliced += d["pc"] is not None
        for got, key in ((q["input_ids"], "ids"), (q["input_mask"], "mask"), (q["segment_ids"], "seg"), (q["label_id"], "label0"), (q["q_id"], "qid")):
            assert got.dtype == np.int64 and np.array_equal(got, gold2[k + key]), (i, key)
        assert np.array_equal(q["target"], gold2[k + "target"]), i
        assert q["mask_token_pos"] == [int(x) for x in gold2[k + "mpos"]], i
        assert int(q["input_mask"][a["max_seq_length"]:].sum()) == min(d["n_regions"], a["max_img_seq_length"]) == int(gold2[k + "feat_rows"] if d["n_regions"] > a["max_img_seq_length"] else d["n_regions"])
    assert spliced >= 2          # (questions with colour words spliced in AND plain ones)


def test_vcr_prompts_match_reference_tensorize_example(gold2, tok2):
    a = json.loads(str(gold2["args"]))
    d = json.loads(str(gold2["vcr_in"]))
    assert prompts.vcr_textize(d["question"], d["colors"], d["names"]) == str(gold2["vcr_text_a"])
    q = prompts.assemble_vcr(tok2, d["question"], d["choices"], d["colors"], d["names"], d["n_regions"], label=d["label"], q_id=d["q_id"],
                             max_seq_length=a["max_seq_length"], max_img_seq_length=a["max_img_seq_length"])
    assert q["texts_b"] == [str(x) for x in gold2["vcr_texts_b"]]
    for got, key in ((q["input_ids"], "ids"), (q["input_mask"], "mask"), (q["segment_ids"], "seg")):
        assert got.dtype == np.int64 and np.array_equal(got, gold2["vcr_" + key]), key
    assert q["mask_token_pos"] == json.loads(str(gold2["vcr_mpos"]))
    assert [q["label"]] * len(d["choices"]) == [int(x) for x in gold2["vcr_label"]] and [q["q_id"]] * len(d["choices"]) == [int(x) for x in gold2["vcr_qid"]]
    # the long fourth choice was cut longest-first to max_seq_length - 3 tokens for the pair
    assert int((q["input_ids"][3] != 0).sum()) == a["max_seq_length"]
