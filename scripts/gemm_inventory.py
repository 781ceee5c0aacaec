"""Dev: one training iteration with T2_GEMM_LOG=1 -> the products of an iteration (stderr), grouped."""
import os, sys, collections, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R)
    import torch
    from tacotron2_subword_amd import _lib as L, train as T
    from tacotron2_subword_amd.hparams import create_hparams
    L.set_precision("bf16")
    hp = create_hparams(); hp.distributed_run = False
    model, opt, crit = T.make_training_objects(hp); model.train()
    x, y = model.parse_batch(T.synthetic_batch(hp, 64, 100, 60, 400, seed=1))
    T.train_step(model, crit, opt, x, y, hp, 0); torch.cuda.synchronize()
    sys.stderr.write("=== ITERATION\n"); sys.stderr.flush()
    T.train_step(model, crit, opt, x, y, hp, 1); torch.cuda.synchronize()
    sys.exit(0)
env = dict(os.environ, T2_GEMM_LOG="1")
p = subprocess.run([sys.executable, __file__, "child"], env=env, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
lines = p.stderr.split("=== ITERATION\n")[-1].splitlines()
c = collections.Counter(l for l in lines if l.startswith("t2gemm"))
tot = 0
for l, n in sorted(c.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {l}")
print(len(c), "distinct,", sum(c.values()), "products per iteration")
