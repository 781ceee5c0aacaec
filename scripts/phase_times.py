"""Dev tool: where one training iteration spends its time, phase by phase, on the GPU timeline (HIP events on the main
stream) and on the host (time at which the launching thread reached the same point).  host << gpu at a point = the GPU
is the bottleneck there; host ~ gpu = the launching thread is."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_subword_amd import _lib as L
from tacotron2_subword_amd.hparams import create_hparams
from tacotron2_subword_amd import train as T

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=4)
a = ap.parse_args()
L.set_precision(a.dtype)
hp = create_hparams()
model, optimizer, criterion = T.make_training_objects(hp)
model.train()
batch = T.synthetic_batch(hp, 64, 100, 60, 400, seed=1234)
x, y = model.parse_batch(batch)

marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e, time.perf_counter()))

fronts, dec_fwd = model._fronts, model.decoder.forward
def fronts_w(*args, **kw):
    out = fronts(*args, **kw)
    mark("encoders fwd")
    out[0].register_hook(lambda g: mark("decoder bwd"))
    return out
def dec_w(*args, **kw):
    out = dec_fwd(*args, **kw)
    mark("decoder fwd")
    out[0].register_hook(lambda g: mark("postnet bwd"))
    return out
model._fronts, model.decoder.forward = fronts_w, dec_w

def step(it):
    marks.clear()
    mark("start")
    model.zero_grad()
    y_pred = model(x)
    mark("postnet fwd")
    loss = criterion(y_pred, y, x, it)[0]
    mark("loss")
    loss.backward()
    mark("encoders bwd (backward() returned)")
    optimizer.step(max_norm=hp.grad_clip_thresh)
    mark("clip + Adam")

for it in range(a.iters):
    torch.cuda.synchronize()
    step(it)
    torch.cuda.synchronize()
t0e, t0h = marks[0][1], marks[0][2]
prev = 0.0
print(f"{'phase':40s} {'gpu ms':>8s} {'(+)':>7s} {'host ms':>8s}")
for name, e, th in marks[1:]:
    g = t0e.elapsed_time(e)
    print(f"{name:40s} {g:8.2f} {g - prev:7.2f} {(th - t0h) * 1e3:8.2f}")
    prev = g
