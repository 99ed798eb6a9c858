"""Register / spill / LDS metadata of the kernels in a built object (build host).  usage: python tools/kmeta.py cpt_amd/csrc/gemm.o [substring ...]"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin/"
obj = sys.argv[1]
pats = sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(d, "x.o")])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    t = subprocess.check_output([LLVM + "llvm-readelf", "--notes", co], text=True)
for m in re.finditer(r"- \.agpr_count(.*?)\.wavefront_size", t, re.S):
    body = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", body).group(1)
    if pats and not any(p in name for p in pats):
        continue
    f = dict(re.findall(r"\.(vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\d+)", body))
    print("%-100s vgpr %s agpr %s sgpr %s spill %s scratch %s" % (name[:100], f.get("vgpr_count"), f.get("agpr_count"), f.get("sgpr_count"), f.get("vgpr_spill_count"), f.get("private_segment_fixed_size")))
