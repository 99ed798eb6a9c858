#!/bin/bash
# Round evidence on ONE box: hipBLASLt yardstick, K-loop counters against hipBLASLt, clocks / power / throttle accumulators under the step,
# then the round's bench lines, PMC passes and rocprofv3 kernel tables (tools/profile_round.sh).   usage: tools/evidence_round.sh r06
T=$1; O=gpurun_out; mkdir -p $O
tools/yardstick.bin > $O/${T}_yardstick.json 2> $O/${T}_yardstick.err
bash tools/kloop_diag.sh ${T}_kloop > $O/${T}_kloop.log 2>&1
cp $O/${T}_kloop/kloop_vs_hipblaslt.md $O/${T}_kloop_vs_hipblaslt.md 2>/dev/null
cp $O/${T}_kloop/summary.json $O/${T}_kloop_summary.json 2>/dev/null
bash tools/power_step_vs_chain.sh $T > /dev/null 2>&1
bash tools/throttle_step_vs_chain.sh $T > /dev/null 2>&1
bash tools/profile_round.sh $T > $O/${T}_profile_round.log 2>&1
tail -3 $O/${T}_profile_round.log
