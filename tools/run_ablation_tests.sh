#!/bin/bash
# The kernel-variant comparisons (pytest marker `ablation`) on the development build of the library (cpt_amd/libcpt_hip_abl.so, -DCPT_ABLATION):
# the product build skips them, its switches are compile-time constants.   usage (GPU box): tools/run_ablation_tests.sh [log]
CPT_AMD_ABLATION=1 python -m pytest tests -m "gpu and ablation" -q 2>&1 | tail -12 | tee ${1:-gpurun_out/ablation_tests.log}
