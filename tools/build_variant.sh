#!/bin/bash
# Developer A/B build of libcpt_hip.so with extra compiler flags: tools/build_variant.sh NAME "-DCPT_WT=0 ..." -> tools/dbg/libcpt_NAME.so
# (git-ignored; run it with CPT_LIB_PATH=tools/dbg/libcpt_NAME.so, e.g. through tools/ab_libs.sh).
# ONLY="gemm_prod gemm_ffn" recompiles just those sources with the flags and links the main build's objects for the rest.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/cpt_variant_$name
rm -rf $obj; mkdir -p $obj $root/tools/dbg
pids=()
for f in $root/cpt_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $b "; then cp $root/cpt_amd/csrc/$b.o $obj/$b.o; continue; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -I$root/include $@ -c $f -o $obj/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $root/tools/dbg/libcpt_$name.so $obj/*.o
echo $root/tools/dbg/libcpt_$name.so
