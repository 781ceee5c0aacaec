"""Diagnostic: per-tensor decoder-backward error (HIP vs fp64 oracle, fp32 oracle vs fp64) for one test configuration."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from oracle import recipe, tacotron2_oracle as O
from helpers import LSA, SMA, hp_for, maxabs, oracle_memories, tiny_hp, to_dev
from tacotron2_subword_amd import _lib as L, ops
from test_gpu_decoder_backward import hip_rnd

att = LSA if len(sys.argv) < 2 else sys.argv[1]
B, Tin, Tsub, T = [int(v) for v in (sys.argv[2:6] if len(sys.argv) >= 6 else (2, 300, 170, 3))]
hp = tiny_hp(att)
seed = 99173
P = recipe.make_weights(hp, seed=5)
x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T, seed=B))
mem, mem_sub = oracle_memories(P, hp, x)
rnd = hip_rnd(ops, L, hp, seed, B, T, Tin, Tsub)
kind = L.ATTN_SMA if att == SMA else L.ATTN_LSA
dec_keys = ["decoder." + k for k in L.decoder_param_keys(kind)]
g = torch.Generator().manual_seed(17)
R_mel, R_gate = torch.randn(B, hp["n_mel_channels"], T, generator=g), torch.randn(B, T, generator=g)

def run(dt):
    cv = lambda v: v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v
    Pg = {k: (cv(v).clone().requires_grad_(True) if k in dec_keys else cv(v)) for k, v in P.items()}
    r = {k: ([cv(a) for a in v] if isinstance(v, list) else cv(v)) for k, v in rnd.items()}
    m1, m2 = cv(mem).clone().requires_grad_(True), cv(mem_sub).clone().requires_grad_(True)
    mel, gate, al, alb = O.decoder_forward(m1, m2, cv(x[3]), x[1], x[2], Pg, hp, r)
    ((mel * cv(R_mel)).sum() + (gate * cv(R_gate)).sum()).backward()
    out = {k: Pg[k].grad for k in dec_keys}
    out["d_memory"], out["d_memory_sub"] = m1.grad, m2.grad
    return out

g32, g64 = run(torch.float32), run(torch.float64)
dims = L.dims_from_hparams(hp)
Pd = to_dev(P)
W = L.decoder_weights(Pd, dims.attention_kind)
memd, memsd = mem.cuda().contiguous(), mem_sub.cuda().contiguous()
dp = ops.decoder_forward(W, dims, memd, memsd, x[1].cuda(), x[2].cuda(), x[3].cuda().contiguous(), training=True, prenet_dropout=True, seed=seed)
G, dmem, dmems = ops.decoder_backward(W, Pd, dims, dp, memd, memsd, R_mel.transpose(1, 2).contiguous().cuda(), R_gate.cuda().contiguous(),
                                      training=True, prenet_dropout=True, seed=seed)
G = dict(G); G["d_memory"], G["d_memory_sub"] = dmem, dmems
rel = lambda a, ref: maxabs(a.double(), ref.double()) / max(float(ref.abs().max()), 1e-6)
print(f"{'tensor':70s} hip-vs-f64  f32oracle-vs-f64  |ref|max")
for k in g64:
    if "attention_layer" in k or "d_memory" in k:
        print(f"{k:70s} {rel(G[k], g64[k]):.2e}    {rel(g32[k], g64[k]):.2e}    {float(g64[k].abs().max()):.3e}")
