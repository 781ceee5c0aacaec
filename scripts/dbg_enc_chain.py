"""Dev: encoder BiLSTM persistent chain vs the per-step launches, error per direction and time step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron2_subword_amd import _lib as L, blocks, ops

B, T, E = int(sys.argv[1]) if len(sys.argv) > 1 else 5, int(sys.argv[2]) if len(sys.argv) > 2 else 4, 512
g = torch.Generator().manual_seed(1)
lstm = torch.nn.LSTM(E, E // 2, 1, batch_first=True, bidirectional=True).cuda()
x = torch.randn(B, T, E, generator=g).cuda()
res = {}
for chain in (True, False):
    L.set_chain(chain)
    xd = x.clone().requires_grad_(True)
    y = blocks.bilstm(xd, None, lstm)
    lstm.zero_grad()
    (y * torch.arange(y.numel(), device="cuda").view_as(y).remainder(7).float()).sum().backward()
    torch.cuda.synchronize()
    res[chain] = (y.detach().clone(), xd.grad.clone(), {k: p.grad.clone() for k, p in lstm.named_parameters()})
print("status", ops.chain_status_words())
y1, y0 = res[True][0], res[False][0]
for d in range(2):
    for t in range(T):
        e = (y1[:, t, d * 256:(d + 1) * 256] - y0[:, t, d * 256:(d + 1) * 256]).abs()
        print(f"dir {d} t {t}: max err {float(e.max()):.3e}  per-row max {[round(float(v), 5) for v in e.max(1).values[:8]]}  unit argmax {int(e.max(0).values.argmax())}")
print("dx rel", float((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max()))
for k in res[True][2]:
    print(k, float((res[True][2][k] - res[False][2][k]).abs().max() / res[False][2][k].abs().max()))

# hypotheses for step t=1 of the forward direction
import torch.nn.functional as F
W_ih, W_hh, b = lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0 + lstm.bias_hh_l0
H = 256
def step(xt, h, c, Wh):
    g = xt @ W_ih.t() + b + h @ Wh.t()
    i, f, gg, o = g[:, :H].sigmoid(), g[:, H:2 * H].sigmoid(), g[:, 2 * H:3 * H].tanh(), g[:, 3 * H:].sigmoid()
    c2 = f * c + i * gg
    return o * c2.tanh(), c2
with torch.no_grad():
    z = torch.zeros(B, H, device="cuda")
    h0, c0 = step(x[:, 0], z, z, W_hh)
    print("t0 vs launch", float((h0 - y0[:, 0, :H]).abs().max()))
    hyps = {"correct": (h0, W_hh), "h=0": (z, W_hh), "W^T-ish": (h0, W_hh.view(4, H, H).transpose(1, 2).reshape(4 * H, H))}
    perm = torch.arange(H, device="cuda").view(8, 2, 16).transpose(0, 1).reshape(-1)
    hyps["k perm A"] = (h0[:, perm], W_hh)
    hyps["k perm B"] = (h0, W_hh[:, perm])
    half = torch.cat([h0[:, :128], torch.zeros_like(h0[:, 128:])], 1)
    hyps["first half K only"] = (half, W_hh)
    for name, (hh, Wh) in hyps.items():
        h1, _ = step(x[:, 1], hh, c0, Wh)
        print(f"{name:20s}: vs chain {float((h1 - y1[:, 1, :H]).abs().max()):.3e}   vs launch {float((h1 - y0[:, 1, :H]).abs().max()):.3e}")
