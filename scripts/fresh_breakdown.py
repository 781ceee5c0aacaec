"""Dev tool: CPU-side and wall time of the pieces of bench.py's fresh-batch loop."""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from tacotron2_subword_amd import _lib as L, data_utils as D, train as T
from tacotron2_subword_amd.hparams import create_hparams
L.set_precision("bf16")
hp = create_hparams(); hp.attention = bench.ATTENTION_NAMES["sma"]; hp.distributed_run = False
model, optimizer, criterion = T.make_training_objects(hp); model.train()
B, Tin, Tsub, Tn = 64, 100, 60, 400
hb = bench.host_batches(T, hp, B, Tin, Tsub, Tn, 4, seed=4321)
x, y = model.parse_batch(T.synthetic_batch(hp, B, Tin, Tsub, Tn, seed=1))
for i in range(4):
    xf, yf = model.parse_batch(D.batch_to_device(hb[i])); T.train_step(model, criterion, optimizer, xf, yf, hp, i)
torch.cuda.synchronize()
def loop(mode, n=8):
    acc = [0.0, 0.0, 0.0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        if mode == "resident": xf, yf = x, y; b = c = a
        else:
            dev = D.batch_to_device(hb[i % 4]); b = time.perf_counter()
            xf, yf = model.parse_batch(dev); c = time.perf_counter()
        T.train_step(model, criterion, optimizer, xf, yf, hp, i); e = time.perf_counter()
        acc[0] += b - a; acc[1] += c - b; acc[2] += e - c
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{mode:9s} wall {1e3*dt/n:6.2f} ms/step | cpu: to_device {1e3*acc[0]/n:5.2f} parse_batch {1e3*acc[1]/n:5.2f} train_step(enqueue) {1e3*acc[2]/n:5.2f}", flush=True)
for m in ("resident", "fresh", "resident", "fresh"):
    loop(m)
