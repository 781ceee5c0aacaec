"""Dev tool: cProfile of the host side of training iterations (where the 24 ms of enqueue time go)."""
import cProfile, os, pstats, sys, torch
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
pr = cProfile.Profile()
pr.enable()
for i in range(5):
    T.train_step(model, criterion, optimizer, x, y, hp, i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
