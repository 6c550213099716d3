"""Developer aid: stage-by-stage comparison of the HIP forward with the bf16 emulation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.ref_bf16 import forward_bf16_emulated
from vit_amd.model import CNNViTHybrid
from vit_amd.synth import synth_state_dict, micro_doppler_batch
from vit_amd import ops

cfg = dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone="resnet18")
m = CNNViTHybrid(**cfg)
sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 1)
m.load_state_dict(sd)
dev = torch.device("cuda:0")
m = m.to(dev).eval()
x = micro_doppler_batch(4, seed=77)
taps = {}
emu = forward_bf16_emulated(sd, x, cfg, taps=taps)
eng = m.engine
P = eng.pack(); B = 4
b = eng.buffers(B, dev)
def cmp(name, got_nhwc, ref_nchw):
    g = got_nhwc.float().cpu().permute(0, 3, 1, 2)
    d = (g - ref_nchw).abs()
    nz = (d > 0).float().mean()
    print(f"{name:8s} max|d|={d.max():.5f} mean|d|={d.mean():.6f} frac!=0 {nz:.5f} mean|ref|={ref_nchw.abs().mean():.4f}")
with torch.no_grad():
    ops.input_pack(x.to(dev), b["img"], 3, 3)
    c1 = P["conv1"]
    ops.conv2d(b["img"], c1["w"], b["c1"], kh=7, kw=1, stride=2, pad=0, seg_len=32, scale=c1["scale"], bias=c1["bias"], act="relu")
    cmp("c1", b["c1"], taps["c1"])
    # feed the emulation's own pooled map to the first block to isolate a single conv
    pool = taps["pool"].permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    blk = P["blocks"][0]
    t = torch.empty_like(pool); o = torch.empty_like(pool)
    a = blk["c1"]; c = blk["c2"]
    ops.conv2d(pool, a["w"], t, kh=3, kw=3, stride=1, pad=1, scale=a["scale"], bias=a["bias"], act="relu")
    ops.conv2d(t, c["w"], o, kh=3, kw=3, stride=1, pad=1, scale=c["scale"], bias=c["bias"], residual=pool, act="relu")
    cmp("l1.0iso", o, taps["l1.0"])
    logits = m(x.to(dev)).cpu()
    print("logits vs emu", float((logits - emu).abs().max()))
    got_tok = b["x"].cpu()
    print("enc3 vs emu", float((got_tok - taps["enc3"]).abs().max()), float(taps["enc3"].abs().mean()))
