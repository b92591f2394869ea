"""Where does the bf16 drift of the full-size C4 model (OTTER-Video-LLaMA7B) come from?  The same model, same weights (bf16-representable),
same batch, forward in fp32 parity mode (matches the host reference to 2e-5, tests/test_gpu_full_model_c4_c5.py) and in bf16 production mode;
per decoder layer: relative error of the residual stream leaving the layer (max over token rows of ||bf16 - fp32|| / ||fp32||), plus the
perceiver output and the logits.  Usage (GPU box): python tools/c4_drift_by_layer.py [config c4|c2] [T]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = "cuda:0"
model = bench.build_model(dev, seed=0, config=cfg, frozen_dtype=torch.float32)
with torch.no_grad():
    for p in model.parameters():
        p.copy_(p.to(torch.bfloat16).to(torch.float32))
model.eval()
vision_x, ids, mask, labels, _ = bench.synth_batch(model, 1, T, dev, seed=777, frames=8 if cfg == "c4" else 1)
layers = model.lang_encoder._get_decoder_layers()
cap = {}

def hook(i):
    def f(mod, args, out):
        h = out[0] if isinstance(out, (tuple, list)) else out
        cap.setdefault(i, []).append(h.detach().float().clone())
    return f

hs = [l.register_forward_hook(hook(i)) for i, l in enumerate(layers)]
hp = model.perceiver.register_forward_hook(lambda m, a, o: cap.setdefault("vis", []).append(o.detach().float().clone()))

def rowrel(a, b):
    a, b = a.reshape(-1, a.shape[-1]).double(), b.reshape(-1, b.shape[-1]).double()
    return float(((a - b).norm(dim=-1) / b.norm(dim=-1).clamp(min=1e-30)).max())

with torch.no_grad():
    o32 = model(vision_x=vision_x, lang_x=ids, attention_mask=mask, labels=labels)
for q in model.parameters():
    if not q.requires_grad:
        q.data = q.data.to(torch.bfloat16)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    o16 = model(vision_x=vision_x.to(torch.bfloat16), lang_x=ids, attention_mask=mask, labels=labels)
print("config %s, T = %d: perceiver output row-rel %.3e" % (cfg, T, rowrel(cap["vis"][1], cap["vis"][0])))
for i in range(len(layers)):
    a, b = cap[i][1], cap[i][0]
    print("layer %2d%s  stream row-rel %.3e   |stream| rms %.3e" % (i, " (gated)" if getattr(layers[i], "gated_cross_attn_layer", None) is not None else "        ",
                                                                   rowrel(a, b), float(b.pow(2).mean().sqrt())))
print("logits row-rel %.3e   loss %.6f vs %.6f" % (rowrel(o16.logits.float(), o32.logits.float()), float(o16.loss), float(o32.loss)))
