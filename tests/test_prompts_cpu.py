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
