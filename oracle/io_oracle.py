"""CPU restatement of the reference's region-feature wire-format decode.

TEST INFRASTRUCTURE ONLY (tests/, bench cpu_baseline legs): the product path is cpt_amd/io.py on the C ABI of
include/cpt_io.h.  Pinned by tests/golden/tiny_rows.tsv + tiny_rows_expected.npz, which oracle/make_golden.py wrote
with the reference's own TSVFile (Oscar/oscar/utils/tsv_file.py) and decode_features
(Oscar/oscar/datasets/refcoco_zsl_cpt_dataset.py:161-180).
"""
import base64
import json

import numpy as np
import torch


def tsv_seek(tsv_path, lineidx_path, idx):
    """utils/tsv_file.py:47-56: byte offset from the .lineidx companion, one line, tab split, stripped."""
    with open(lineidx_path, "r") as fp:
        pos = [int(i.strip()) for i in fp.readlines()][idx]
    with open(tsv_path, "r") as fp:
        fp.seek(pos)
        return [s.strip() for s in fp.readline().split("\t")]


def decode_features(row_cols):
    """refcoco_zsl_cpt_dataset.py:161-180."""
    img_name, feat_str = row_cols
    feat_info = json.loads(feat_str)
    objs, caption, colors, rect_lists = feat_info["objects"]
    im_feats, od_labels = [], []
    for boxlist in objs:
        feats = [np.frombuffer(base64.b64decode(o["feature"]), np.float32) for o in boxlist]
        im_feats.append(torch.Tensor(np.stack(feats)))
        od_labels.append(" ".join([o["class"] for o in boxlist]))
    return img_name, od_labels, im_feats, caption, colors, rect_lists


def pad_regions(im_feats, img_seq_len, dim=2054):
    """refcoco_zsl_cpt_dataset.py:119-120 + the image part of input_mask built by tokenize()."""
    feats = [torch.cat([f, torch.zeros([img_seq_len - f.size(0), dim])], 0) for f in im_feats]
    mask = [[1] * f.size(0) + [0] * (img_seq_len - f.size(0)) for f in im_feats]
    return torch.stack(feats), torch.tensor(mask, dtype=torch.int64)
