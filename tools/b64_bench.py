"""Stand-alone timing of the device base64 decoder (cpt_b64_decode_regions_device) at the bench shape and at 8x the regions.  usage (GPU box): python tools/b64_bench.py"""
import torch, sys
sys.path.insert(0,'/root/repo')
from cpt_amd import io
dev=torch.device('cuda:0')
chars=io.b64_chars(2054)
for Bs in (64, 512):
    txt = torch.randint(65, 91, (Bs, 50, chars), dtype=torch.uint8, device=dev); txt[:, :, chars-1] = 61
    mk = torch.ones(Bs, 50, dtype=torch.int64, device=dev); fo = torch.empty(Bs, 50, 2054, device=dev); derr = torch.zeros(1, dtype=torch.int64, device=dev)
    for _ in range(3): io.decode_text_device(txt, mk, fo, derr)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): io.decode_text_device(txt, mk, fo, derr)
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/20*1e-3; nb=Bs*50*(chars+2054*4)
    io.check_device_decode(derr, 50)
    print(Bs, 'us', round(t*1e6,2), 'frac', round(nb/t/8e12,3))
