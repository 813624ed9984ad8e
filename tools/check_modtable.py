"""Debug aid: accuracy of the ModulationTable (all-steps adaLN GEMM) against the per-step GEMV path and the fp32 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import flux_oracle as orc
from reptext_amd import mmdit
from reptext_amd.transformer import FluxTransformer2DModel

cfg = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=4,
           joint_attention_dim=256, pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
gpu = torch.device("cuda:0")
tp = orc.init_mmdit_params(cfg, seed=11)
tr = FluxTransformer2DModel(**cfg, device=gpu, dtype=torch.bfloat16)
tr.load_state_dict(tp)
g = torch.Generator().manual_seed(5)
pooled = torch.randn(1, 64, generator=g).to(torch.bfloat16).float()
ts = [1.0, 0.622459]
guid = torch.full((1,), 3.5)
tab = tr.build_modulation_table(ts, guid.to(gpu), pooled.to(gpu, torch.bfloat16))
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for i, t in enumerate(ts):
    temb = orc.time_text_embed(tp, "time_text_embed", torch.full((1,), t) * 1000, guid * 1000, pooled)
    sm = tab.step(i)
    for blk in range(2):
        ref_i = orc.linear(tp, f"transformer_blocks.{blk}.norm1.linear", orc.silu(temb))
        ref_t = orc.linear(tp, f"transformer_blocks.{blk}.norm1_context.linear", orc.silu(temb))
        print(f"step {i} double {blk}: img {rel(sm.double[blk][0].cpu(), ref_i):.2e} txt {rel(sm.double[blk][1].cpu(), ref_t):.2e}")
        ref_s = orc.linear(tp, f"single_transformer_blocks.{blk}.norm.linear", orc.silu(temb))
        print(f"step {i} single {blk}: {rel(sm.single[blk].cpu(), ref_s):.2e}")
    ref_o = orc.linear(tp, "norm_out.linear", orc.silu(temb))
    print(f"step {i} out: {rel(sm.out.cpu(), ref_o):.2e}")
    sc = mmdit.EmbedScratch(1, 512, gpu)
    temb_g = tr._temb(sc, torch.full((1,), t, device=gpu), guid.to(gpu), pooled.to(gpu, torch.bfloat16))
    print(f"step {i} temb: {rel(temb_g.cpu(), temb):.2e}")
