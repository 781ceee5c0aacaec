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


def _worker_nccl(rank, world, port, q):
    """world_size-1 process group over RCCL (backend "nccl" IS RCCL on ROCm): the bucketed, hook-launched
    all_reduce(async_op=True) path of the wrapper on a real RCCL stream, in the bf16 mode bench.py runs, persistent
    chain kernels and side streams included.  With one rank the averaged gradient must equal the local one."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0",
                      T2_CHAIN_FORCE="1")    # the parent (pytest) may hold this GPU's persistent-kernel claim; it is idle while this runs
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from tacotron2_subword_amd import _lib as L
        from tacotron2_subword_amd.hparams import create_hparams
        from tacotron2_subword_amd.model import BERT_Tacotron2
        from tacotron2_subword_amd.loss_function import Tacotron2Loss
        from tacotron2_subword_amd.distributed import apply_gradient_allreduce
        from tacotron2_subword_amd import train as T
        reduce_tensor = T.reduce_tensor
        L.set_precision("bf16")
        hp = create_hparams()
        B, Tin, Tsub, Tn = 8, 24, 15, 40
        torch.manual_seed(1234)
        ref = BERT_Tacotron2(hp).cuda().eval()
        ref.decoder.prenet_dropout = False
        m = BERT_Tacotron2(hp).cuda().eval()
        m.decoder.prenet_dropout = False
        m.load_state_dict(ref.state_dict())
        m = apply_gradient_allreduce(m)

        def grads(model, seed):
            x, y = model.parse_batch(T.synthetic_batch(hp, B, Tin, Tsub, Tn, seed=seed))
            model.zero_grad()
            loss = Tacotron2Loss()(model(x), y, x)[0]
            loss.backward()
            torch.cuda.synchronize()
            return float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

        for step in range(3):                              # step 0: reduction at the end of backward; later: from the hooks
            l0, want = grads(ref, 70 + step)
            l1, got = grads(m, 70 + step)
            assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0))
            assert set(got) == set(want)
            for k in want:
                err = float((got[k] - want[k]).abs().max()) / max(float(want[k].abs().max()), 1e-6)
                assert err < 1e-5, (step, k, err)
        assert L.lib().t2_chain_claimed() == 1 and L.get_chain()               # the persistent chains really ran in this process
        r = reduce_tensor(torch.tensor([3.0], device="cuda"), world)          # train.py:23-27
        assert float(r) == 3.0
        q.put((rank, "ok", ""))
    except Exception:  # noqa
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_rccl_world1_real_model_bf16():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_nccl, args=(0, 1, _free_port(), q))
    p.start()
    rank, status, info = q.get(timeout=600)
    p.join(timeout=60)
    assert status == "ok", info


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
