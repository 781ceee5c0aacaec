"""Dev: WHEN does each gradient bucket become ready during backward (one GPU, RCCL process group of one rank)?  The bucketed
reductions of distributed.apply_gradient_allreduce start from autograd hooks; what they can overlap with is whatever backward
work follows a bucket's launch.  Prints launch time of every bucket relative to the start of backward, and the step's phases."""
import os, sys, socket, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from tacotron2_subword_amd import _lib as L, train as T
from tacotron2_subword_amd.hparams import create_hparams
L.set_precision("bf16")
hp = create_hparams(); hp.distributed_run = True
model, opt, crit = T.make_training_objects(hp); model.train()
x, y = model.parse_batch(T.synthetic_batch(hp, 64, 100, 60, 400, seed=1))
for i in range(3):
    T.train_step(model, crit, opt, x, y, hp, i)
arena = model._t2_arena
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
for rep in range(2):
    arena.launch_events, arena.exposed = [], []
    e_f0, e_b0, e_b1, e_s1 = ev(), ev(), ev(), ev()
    model.zero_grad(); e_f0.record()
    loss = crit(model(x), y, x, 0)[0]
    e_b0.record(); loss.backward(); e_b1.record()
    opt.step(max_norm=hp.grad_clip_thresh); e_s1.record()
    torch.cuda.synchronize()
    print(f"rep {rep}: forward {e_f0.elapsed_time(e_b0):.2f} ms, backward {e_b0.elapsed_time(e_b1):.2f} ms, optimizer {e_b1.elapsed_time(e_s1):.2f} ms; "
          f"exposed wait at the end of backward {sum(a.elapsed_time(b) for a, b in arena.exposed):.3f} ms")
    names = {id(p): k for k, p in model.named_parameters()}
    for i, n, e in arena.launch_events:
        first = names[id(arena.buckets[i][0])]; last = names[id(arena.buckets[i][-1])]
        print(f"   bucket {i:2d} ({4 * n / 2**20:6.1f} MB: {first} .. {last}) launched {e_b0.elapsed_time(e):7.2f} ms after backward began")
dist.destroy_process_group()
