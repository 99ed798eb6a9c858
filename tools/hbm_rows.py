"""The HBM-bound row kernels stand-alone (bench.py hbm_kernels): python tools/hbm_rows.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cpt_amd import config as cfgmod  # noqa: E402

if __name__ == "__main__":
    out = bench.hbm_kernels(cfgmod.oscar_base(), 64, torch.device("cuda:0"), iters=50)
    for k, v in out.items():
        print("%-110s %8.2f us %7.1f GB/s  %.3f" % (k[:110], v["us"], v["GB/s"], v["frac_of_8TB/s"]))
