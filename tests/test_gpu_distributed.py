"""Two ranks on ONE GPU (gloo moves the CUDA buckets through the host): the data-parallel wrapper around the real
BERT_Tacotron2 — whose backward runs on two streams (model._fronts) plus the library's side stream — must leave
every rank with the mean of the per-rank gradients.  RCCL itself needs one device per rank and is exercised by
bench.py --gpus N on the multi-GPU node."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tacotron2_subword_amd.hparams import create_hparams
        from tacotron2_subword_amd.model import BERT_Tacotron2
        from tacotron2_subword_amd.loss_function import Tacotron2Loss
        from tacotron2_subword_amd.distributed import apply_gradient_allreduce
        from tacotron2_subword_amd import train as T
        torch.cuda.set_device(0)
        hp = create_hparams()
        B, Tin, Tsub, Tn = 4, 20, 17, 36

        def local_grads(model, r, step):
            x, y = model.parse_batch(T.synthetic_batch(hp, B, Tin, Tsub, Tn, seed=50 + r))
            model.zero_grad()
            loss = Tacotron2Loss()(model(x), y, x)[0]
            loss.backward()
            torch.cuda.synchronize()
            return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

        torch.manual_seed(1234)
        # eval mode + prenet dropout off: no random bits, so every process can recompute every rank's local gradient
        ref = BERT_Tacotron2(hp).cuda().eval()
        ref.decoder.prenet_dropout = False
        sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
        torch.manual_seed(999 + rank)                     # different init per rank: the wrap-time broadcast must fix it
        m = BERT_Tacotron2(hp).cuda().eval()
        m.decoder.prenet_dropout = False
        if rank == 0:
            m.load_state_dict(sd0)
        m = apply_gradient_allreduce(m)
        for k, v in m.state_dict().items():
            assert torch.equal(v, sd0[k]), k
        for step in range(2):                              # step 0: first-backward path; step 1: hook-launched buckets
            ref.load_state_dict(sd0)
            loc = [local_grads(ref, r, step) for r in range(world)]
            got = local_grads(m, rank, step)
            for k, g in got.items():
                want = sum(l[k] for l in loc) / world
                err = float((g - want).abs().max()) / max(float(want.abs().max()), 1e-6)
                assert err < 1e-5, (step, k, err)
            assert not any(k.startswith("decoder.decoder_rnn_bert") for k in got)
        q.put((rank, "ok", ""))
    except Exception:  # noqa
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_dp_world2_real_model_on_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, status, info in out:
        assert status == "ok", f"rank {rank}: {info}"
