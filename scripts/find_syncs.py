"""Dev tool: which calls of a training iteration synchronise the host with the GPU (torch's sync debug mode)."""
import os, sys, warnings, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from tacotron2_subword_amd import _lib as L, train as T
from tacotron2_subword_amd.hparams import create_hparams
L.set_precision("bf16")
hp = create_hparams(); hp.attention = bench.ATTENTION_NAMES["sma"]; hp.distributed_run = False
model, optimizer, criterion = T.make_training_objects(hp); model.train()
x, y = model.parse_batch(T.synthetic_batch(hp, 64, 100, 60, 400, seed=1))
for i in range(3):
    T.train_step(model, criterion, optimizer, x, y, hp, i)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    import traceback
    orig = warnings.showwarning
    T.train_step(model, criterion, optimizer, x, y, hp, 3)
torch.cuda.set_sync_debug_mode("default")
print("synchronising calls in one iteration:", len(w))
for m in w:
    print("  ", m.filename.replace(R + "/", ""), m.lineno, str(m.message)[:80])

# host time of the phases of an iteration (enqueue only: no synchronisation inside), GPU kept busy by the previous iteration
import time
acc = [0.0] * 5
N = 6
torch.cuda.synchronize()
t_all = time.perf_counter()
for i in range(N):
    a = time.perf_counter(); model.zero_grad()
    b = time.perf_counter(); y_pred = model(x)
    c = time.perf_counter(); loss = criterion(y_pred, y, x, i)[0]
    d = time.perf_counter(); loss.backward()
    e = time.perf_counter(); optimizer.step(max_norm=hp.grad_clip_thresh)
    f = time.perf_counter()
    for k, v in enumerate((b - a, c - b, d - c, e - d, f - e)): acc[k] += v
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / N
print("host ms per iteration: zero_grad %.2f forward %.2f loss %.2f backward %.2f step %.2f | wall %.2f" % tuple([1e3 * v / N for v in acc] + [1e3 * wall]))
