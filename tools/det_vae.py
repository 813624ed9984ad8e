"""Debug aid: is the VAE decode/encode bitwise reproducible (within a process and across processes)?"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vae_oracle as vorc
from reptext_amd.vae import AutoencoderKL

cfg = dict(vorc.FLUX_VAE_CFG, block_out_channels=(64, 128, 256, 256))
gpu = torch.device("cuda:0")
p = vorc.init_vae_params(cfg, seed=3)
vae = AutoencoderKL(**{k: v for k, v in cfg.items()}, device=gpu, dtype=torch.bfloat16)
vae.load_state_dict(p)
g = torch.Generator().manual_seed(1)
z = torch.randn(1, 16, 16, 16, generator=g).to(torch.bfloat16)
x = (torch.rand(1, 3, 128, 128, generator=g) * 2 - 1).to(torch.bfloat16)
h = lambda t: hashlib.md5(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:12]
outs, encs = [], []
for i in range(4):
    outs.append(vae.decode(z.to(gpu), return_dict=False)[0].clone())
    encs.append(vae.encode(x.to(gpu)).latent_dist.mean.clone())
print("decode", [h(o) for o in outs])
print("encode", [h(o) for o in encs])
for i in range(1, 4):
    d = (outs[i].float() - outs[0].float()).abs()
    print(i, "decode max diff", float(d.max()), "n diff", int((d > 0).sum()))
