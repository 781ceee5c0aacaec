"""Dev tool: bench.py's fed loop (collate -> pinned stage -> H2D -> parse_batch -> training iteration) on its own, with
host-side stamps per phase and GPU-side step times; run it under `rocprofv3 --kernel-trace --memory-copy-trace` and feed the
trace to scripts/trace_gaps.py to see what the GPU waited for.   usage: fed_loop.py [steps] [mode: fed|resident]"""
import gc, os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from tacotron2_subword_amd import _lib as L, data_utils as D, train as T
from tacotron2_subword_amd.hparams import create_hparams
def cgroup_cpu():
    out = {}
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat"):
        try:
            out[f] = open(f).read().split("\n")
        except OSError as e:
            out[f] = str(e)
    return out
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
if len(sys.argv) > 4:
    torch.set_num_threads(int(sys.argv[4]))
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
mode = sys.argv[2] if len(sys.argv) > 2 else "fed"
L.set_precision("bf16")
hp = create_hparams(); hp.attention = bench.ATTENTION_NAMES["sma"]; hp.distributed_run = False
model, optimizer, criterion = T.make_training_objects(hp); model.train()
B, Tin, Tsub, Tn = 64, 100, 60, 400
hi = bench.host_items(T, hp, B, Tin, Tsub, Tn, 4, seed=4321)
x, y = model.parse_batch(T.synthetic_batch(hp, B, Tin, Tsub, Tn, seed=1))
stamps = []
def step(i, rec=False):
    a = time.perf_counter()
    if mode == "fed":
        st = D._RING.get(1); b = time.perf_counter()
        batch = D.collate_batch(hi[i % 4], st); c = time.perf_counter()
        dev = D.batch_to_device(batch); d = time.perf_counter()
        xf, yf = model.parse_batch(dev)
    else:
        b = c = d = a; xf, yf = x, y
    e = time.perf_counter()
    y_pred = model(xf); f = time.perf_counter()
    loss = criterion(y_pred, yf, xf, i)[0]; g = time.perf_counter()
    model.zero_grad(); loss.backward(); h = time.perf_counter()
    optimizer.step(max_norm=hp.grad_clip_thresh); k = time.perf_counter()
    if rec:
        stamps.append([1e3 * v for v in (b - a, c - b, d - c, e - d, f - e, g - f, h - g, k - h)])
for i in range(8):
    step(i)
torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[3] == "nogc":
    gc.disable()
cg0 = cgroup_cpu()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
t0 = time.perf_counter(); evs[0].record()
for i in range(steps):
    step(i, True); evs[i + 1].record()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
cg1 = cgroup_cpu()
print("cgroup before:", cg0); print("cgroup after: ", cg1)
gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
print(f"{mode}: {1e3 * dt / steps:.2f} ms/step wall; GPU-side step ms: median {sorted(gpu)[steps // 2]:.2f} max {max(gpu):.2f}")
names = ["stage.wait", "collate", "to_device", "parse_batch", "forward", "loss", "backward", "optim"]
print("step  gpu_ms | " + " ".join(f"{n:>11s}" for n in names))
for i, (gm, st) in enumerate(zip(gpu, stamps)):
    print(f"{i:4d} {gm:7.2f} | " + " ".join(f"{v:11.2f}" for v in st))
