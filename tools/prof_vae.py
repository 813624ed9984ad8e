"""VAE decode alone (1024^2), for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_vae.py`."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reptext_amd.config import flux_vae_config
from reptext_amd.vae import AutoencoderKL
dev = torch.device("cuda:0")
vae = AutoencoderKL(**flux_vae_config(), device=dev, dtype=torch.bfloat16).random_init_(seed=2)
z = torch.randn(1, 16, 128, 128, device=dev).to(torch.bfloat16)
for _ in range(4):
    img = vae.decode(z, return_dict=False)[0]
torch.cuda.synchronize()
print(img.shape)
