#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s of the full training iteration (forward + loss + backward +
gradient clip + Adam; train.py:293-340 of the reference) at tensor batch B=64 per GPU, default
hparams (512-dim / 80-mel, dual-stream BERT_Tacotron2, StepwiseMonotonicAttention), synthetic
LJSpeech-shaped batches (100 phones, 60 sub-word tokens, 400 frames) — BASELINE.json configs[1];
with --gpus N: configs[2] (data parallel over RCCL, weak scaling, one process per GPU).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train|infer|gta] [--attention sma|lsa]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

--workload infer = configs[3] (autoregressive inference(), B=32, 1000 decoder steps, stop disabled; metric
decode_steps_per_sec), --workload gta = configs[4] (teacher-forced forward under no_grad, eval, B=128/GPU; replicas
only, no collective).  The default train line also carries a short "decode" measurement (N=1).

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself — fresh processes, before
anything touches a GPU, as the reference's distributed.py:181-200 does — and rank 0 prints the line.

Rank 0 prints ONE JSON line.  Extra objects (N = 1 unless noted):
  roofline      dominant decoder kernel: algorithmic bytes (or FLOPs) per launch / its average duration, measured with
                HIP events on the launch stream in one extra (untimed) profiled step
  fresh_batches the same loop fed like train.py:286-316: every step collates the next ragged batch out of its dataset items
                (data_utils.collate_batch into a page-locked stage), uploads it (batch_to_device) and runs parse_batch +
                the iteration; no host synchronisation anywhere in it
  fp32          the parity mode (exact fp32 GEMMs everywhere) on the same batch, a short leg
  bf16_error    max-abs difference of mel / gate / alignments between the two modes on the bench batch (eval forward)
  decode        BASELINE configs[3]: inference() at B=32, 1000 decoder steps, stop rule disabled
  gta           BASELINE configs[4]: teacher-forced forward, eval, no_grad, B=128
  cpu_baseline  the CPU oracle (torch-CPU restatement of the reference, oracle/) timed on this box's host cores on a
                bounded sample of the same workload
  rccl_world_size (every N) the size of the process group the ranks formed (backend nccl = RCCL), 1 without one
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ATTENTION_NAMES = {"sma": "StepwiseMonotonicAttention", "lsa": "LSA", "fa2": "ForwardAttentionV2", "gmm": "GMMAttention",
                   "dca": "DynamicConvolutionAttention"}

# peaks from /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF headline is 2:1 sparse)


def decoder_step_flops(hp, B, Tin, Tsub):
    """Algorithmic FLOPs of the per-step kernels: 2 * MACs."""
    P, E, Ha, Hd, A = hp.prenet_dim, hp.encoder_embedding_dim, hp.attention_rnn_dim, hp.decoder_rnn_dim, hp.attention_dim
    att_lstm_fwd = 2 * 2 * B * (E + Ha) * 4 * Ha + 2 * 2 * B * Ha * A       # recurrent half of both cells + query projection
    dec_lstm_fwd = 2 * B * Hd * 4 * Hd                                       # recurrent half (input half is a hoisted GEMM)
    return dict(att_lstm_fwd=att_lstm_fwd, dec_lstm_fwd=dec_lstm_fwd,
                att_lstm_bwd_gemm=2 * 2 * B * 4 * Ha * (E + Ha), dec_lstm_bwd_gemm=2 * B * 4 * Hd * Hd)


def decoder_step_bytes(hp, B, Tin, Tsub, wbytes=4):
    """Algorithmic bytes per launch: weights once + activations once.  wbytes = 4 (fp32 operands) or 2 (bf16
    shadows of the recurrent weights / activations; state, gates and pre-activations stay fp32)."""
    P, E, Ha, Hd, A = hp.prenet_dim, hp.encoder_embedding_dim, hp.attention_rnn_dim, hp.decoder_rnn_dim, hp.attention_dim
    att = 2 * (4 * Ha * (E + Ha) * wbytes + A * Ha * 4) + 2 * B * ((E + Ha) * wbytes + (4 * Ha + 4 * Ha + 3 * Ha + Ha // 8 * A) * 4)
    dec = 4 * Hd * Hd * wbytes + B * (Hd * wbytes + (4 * Hd + 4 * Hd + 3 * Hd) * 4)
    attn = B * (Tin + Tsub) * (E + A) * 4
    # chain_*: one persistent launch covers `steps` steps; its algorithmic bytes are the per-step figures above times the
    # steps (SURVEY.md section 8d counts what must move if nothing stays on chip between steps — the launch keeps the
    # weights in registers, so its HBM-side traffic is far below this figure; that is the point of it)
    pw_att = 2 * B * Ha * 16 * 4          # pointwise BPTT of both attention LSTMs: gates, cells, dh sources in; dg (fp32 + bf16) out
    pw_dec = B * Hd * 16 * 4
    bwd_att = 2 * (4 * Ha * (E + Ha) + B * 4 * Ha) * wbytes + 2 * 8 * B * (E + Ha) * 4
    bwd_dec = (4 * Hd * Hd + B * 4 * Hd) * wbytes + 8 * B * Hd * 4
    # (the persistent chain's figure leaves out the [Ha/8][B][A] query partials the launch-path kernel writes: they are that
    #  implementation's own exchange, not bytes the step must move)
    att_alg = att - 2 * B * (Ha // 8 * A) * 4
    return dict(att_lstm_fwd=att, dec_lstm_fwd=dec, attention_fwd=attn, attention_bwd=2 * attn,
                chain_a_fwd_per_step=att_alg + attn, chain_b_fwd_per_step=dec,
                chain_a_bwd_per_step=2 * attn + pw_att + bwd_att, chain_b_bwd_per_step=pw_dec + bwd_dec,
                att_lstm_bwd_gemm=2 * (4 * Ha * (E + Ha) + B * 4 * Ha) * wbytes + 2 * 8 * B * (E + Ha) * 4,
                dec_lstm_bwd_gemm=(4 * Hd * Hd + B * 4 * Hd) * wbytes + 8 * B * Hd * 4)


def cpu_baseline(B=64, Tin=100, Tsub=60, T=400, reps=1):
    """Reported baseline, not the target: the oracle's fp32 training iteration (forward + loss +
    backward + clip + Adam) on the host cores, bounded sample."""
    from oracle import recipe
    from oracle import tacotron2_oracle as O
    hp = O.default_hparams()
    P = recipe.make_weights(hp)
    params = []
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k and not k.startswith("decoder.decoder_rnn_bert"):
            v.requires_grad_(True)
            params.append(v)
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-6)
    x, y = recipe.parse_batch(recipe.make_batch(hp, B, Tin, Tsub, T))
    rnd = recipe.make_rnd(hp, B, Tin, Tsub, T)
    cores = torch.get_num_threads()
    best = None
    for _ in range(reps + 1):                      # first pass warms the allocator / thread pool
        t0 = time.perf_counter()
        opt.zero_grad()
        out = O.forward(P, hp, x, training=True, rnd=rnd, new_stats={})
        loss, _, _ = O.loss(out, y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return dict(value=B * T / best, unit="mel-frames/s", cores=cores, kind="port",
                sample=f"oracle training iteration (fwd+loss+bwd+clip+Adam), fp32, B={B} Tin={Tin} Tsub={Tsub} T={T}, "
                       f"best of {reps + 1} ({best:.2f} s)")


def decode_bench(model, hp, B, Tin, Tsub, steps, reps, seed=77):
    """Autoregressive inference() (BASELINE configs[3]): stop rule disabled, `steps` decoder steps per call."""
    from tacotron2_subword_amd import train as T
    b = T.synthetic_batch(hp, B, Tin, Tsub, 8, seed=seed)
    ids, sub, pcls, bcls = b[0].cuda(), b[6].cuda(), b[7].cuda(), b[8].cuda()
    was_training = model.training
    model.eval()
    old = model.decoder.gate_threshold, model.decoder.max_decoder_steps
    model.decoder.gate_threshold, model.decoder.max_decoder_steps = 2.0, steps
    try:
        with torch.no_grad():
            model.inference(ids, sub, pcls, bcls)            # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = model.inference(ids, sub, pcls, bcls)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
    finally:
        model.decoder.gate_threshold, model.decoder.max_decoder_steps = old
        model.train(was_training)
    assert out[0].shape[2] == steps
    return dict(batch=B, steps_per_call=steps, ms_per_call=round(1e3 * dt, 2), steps_per_sec=round(steps / dt, 1),
                frames_per_sec=round(B * steps / dt, 1), us_per_step=round(1e6 * dt / steps, 2))


def side_workload(a, rank, world, local):
    """--workload infer / gta: replicas only (no collective on the data path), one JSON line from rank 0."""
    from tacotron2_subword_amd import _lib as L
    from tacotron2_subword_amd.hparams import create_hparams
    from tacotron2_subword_amd import train as T
    L.set_precision(a.dtype)
    hp = create_hparams()
    hp.attention = ATTENTION_NAMES[a.attention]
    hp.distributed_run = False
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend=os.environ.get("T2_DIST_BACKEND", "nccl"), init_method="env://", world_size=world, rank=rank)
    model = T.load_model(hp)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if a.workload == "infer":
        B = a.batch or 32
        steps_per_call = a.frames or 1000
        model.eval()
        b = T.synthetic_batch(hp, B, a.tin, a.tsub, 8, seed=1234 + rank)
        ids, sub, pcls, bcls = b[0].cuda(), b[6].cuda(), b[7].cuda(), b[8].cuda()
        model.decoder.gate_threshold, model.decoder.max_decoder_steps = 2.0, steps_per_call
        run = lambda: model.inference(ids, sub, pcls, bcls)
        units, metric, unit = steps_per_call, "decode_steps_per_sec", "decoder-steps/s"
        desc = (f"BERT_Tacotron2.inference() B={B}/GPU, stop rule disabled, {steps_per_call} decoder steps per call "
                f"({a.tin} phones, {a.tsub} sub-word tokens); one bench step = one call")
    else:
        B = a.batch or 128
        Tn = a.frames or 400
        model.eval()
        batch = T.synthetic_batch(hp, B, a.tin, a.tsub, Tn, seed=1234 + rank)
        x, y = model.parse_batch(batch)
        run = lambda: model(x)
        units, metric, unit = B * Tn, "mel_frames_per_sec_gta", "mel-frames/s"
        desc = (f"GTA teacher-forced forward (no_grad, eval), B={B}/GPU, {a.tin} phones, {a.tsub} sub-word tokens, "
                f"{Tn} frames; one bench step = one forward")
    with torch.no_grad():
        for _ in range(a.warmup):
            run()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run()
        sync()
        dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
        torch.distributed.barrier()
    if rank != 0:
        return
    out = {"metric": metric, "value": round(world * units * a.steps / dt, 1), "unit": unit, "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 2), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": desc, "attention": hp.attention, "global_batch": world * B, "parallelism": f"replicas x{world}"}}
    if a.workload == "infer":
        out["frames_per_sec"] = round(world * B * units * a.steps / dt, 1)
        out["us_per_decoder_step"] = round(1e6 * dt / (a.steps * units), 2)
    print(json.dumps(out), flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh processes (one per GPU, env:// rendezvous
    on 127.0.0.1) BEFORE this process has touched a GPU — the job of the reference's distributed.py:181-200 — and exit
    with the worst return code.  Rank 0 inherits stdout and prints the JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p_ in procs:
        rc = max(rc, abs(p_.wait()))
    raise SystemExit(rc)


def host_items(T, hp, B, Tin, Tsub, Tn, n, seed):
    """n lists of B dataset items, each item what data_utils.BERTTacotron2Dataset.__getitem__ returns for one utterance
    (ragged lengths; the CLS vector as a stride-0 view over time, data_utils.py:76-79): what a loader step collates."""
    out = []
    for i in range(n):
        text, il, ilb, mel, gate, ol, sub, pcls, bcls, _ = T.synthetic_batch(hp, B, Tin, Tsub, Tn, seed=seed + 17 * i)
        items = []
        for b in range(B):
            nt, ns, nm = int(il[b]), int(ilb[b]), int(ol[b])
            stop = torch.zeros(nm, dtype=torch.float64)
            stop[-1] = 1.0
            items.append({"text": text[b, :nt].clone(), "mel_target": mel[b, :, :nm].t().contiguous().numpy(),
                          "bert_embedding": sub[b, :ns].clone(), "stop_token": stop.numpy(),
                          "bert_embedding_cls": pcls[b, :1].clone().expand(ns, -1), "phoneme_embedding_cls": pcls[b, :1].clone().expand(nt, -1)})
        out.append(items)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 64 train, 32 infer, 128 gta)")
    ap.add_argument("--tin", type=int, default=100)
    ap.add_argument("--tsub", type=int, default=60)
    ap.add_argument("--frames", type=int, default=0, help="frames per item (default 400; infer: decoder steps per call, default 1000)")
    ap.add_argument("--workload", choices=["train", "infer", "gta"], default="train")
    ap.add_argument("--attention", choices=["sma", "lsa", "fa2", "gmm", "dca"], default="sma",
                    help="sma = the reference's default hparams (StepwiseMonotonicAttention); lsa = LocationSensitiveAttention; "
                         "fa2 / gmm / dca = ForwardAttentionV2 / GMMAttention / DynamicConvolutionAttention")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="train workload: only the timed loop and the roofline (no fresh-batch, "
                                                             "fp32, decode, gta legs)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="bf16",
                    help="GEMM operand type: f32 = exact fp32 (parity path); bf16 = bf16 operands, fp32 accumulate/state")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a.gpus)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", rank))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = max(torch.cuda.device_count(), 1)
    if world > ndev:
        # several ranks on one GPU (a rehearsal): the persistent chain kernels need the whole device to themselves
        os.environ["T2_CHAIN"] = "0"
    torch.cuda.set_device(local % ndev)
    if a.workload != "train":
        return side_workload(a, rank, world, local)
    a.batch, a.frames = a.batch or 64, a.frames or 400

    from tacotron2_subword_amd import _lib as L
    from tacotron2_subword_amd import data_utils as D
    from tacotron2_subword_amd.hparams import create_hparams
    from tacotron2_subword_amd import train as T

    L.set_precision(a.dtype)
    from tacotron2_subword_amd.utils import cpu_quota, fit_cpu_threads
    host_threads = fit_cpu_threads()                                    # (load_model does this too; here so that it is reported)
    hp = create_hparams()
    hp.attention = ATTENTION_NAMES[a.attention]
    hp.distributed_run = world > 1
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("T2_DIST_BACKEND", "nccl")
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    model, optimizer, criterion = T.make_training_objects(hp)
    model.train()
    B, Tin, Tsub, Tn = a.batch, a.tin, a.tsub, a.frames
    batch = T.synthetic_batch(hp, B, Tin, Tsub, Tn, seed=1234 + rank)
    x, y = model.parse_batch(batch)                 # inputs resident in HBM before the timed region

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(a.warmup):
        T.train_step(model, criterion, optimizer, x, y, hp, it)
        it += 1
    arena = getattr(model, "_t2_arena", None)
    if arena is not None:
        arena.exposed = []                                              # (events only: nothing synchronises inside the loop)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = T.train_step(model, criterion, optimizer, x, y, hp, it)
        it += 1
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0                                 # this rank's own loop, before the closing barrier
    sync()
    dt = time.perf_counter() - t0
    ranks = None
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
        # per-rank picture for reading a scaling curve: each rank's own loop time and the part of the gradient reductions
        # its backward did not hide (time the launch stream spent waiting for them at the end of backward)
        exposed = sum(e0.elapsed_time(e1) for e0, e1 in arena.exposed) / max(1, len(arena.exposed)) if arena is not None and arena.exposed else 0.0
        arena.exposed = None
        mine = torch.tensor([1e3 * dt_local / a.steps, exposed], device="cuda", dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        per = [[round(float(v), 3) for v in t.tolist()] for t in allr]
        ranks = {"ms_per_step": [p_[0] for p_ in per], "ms_per_step_min": min(p_[0] for p_ in per), "ms_per_step_max": max(p_[0] for p_ in per),
                 "exposed_allreduce_ms": [p_[1] for p_ in per],
                 "gradient_bytes": int(arena.flat.numel() * 4) if arena is not None else 0, "buckets": len(arena.buckets) if arena is not None else 0,
                 "note": "ms_per_step: each rank's own timed loop (before the closing barrier); exposed_allreduce_ms: mean time per step its "
                         "launch stream waited for the bucketed RCCL reductions at the end of backward (HIP events) — what the overlap did not hide"}
    loss_val = float(loss.item())
    extras = world == 1 and not a.no_extras

    # the same loop fed from the host, as train.py:286-316 feeds it: every step collates the next model batch out of its B
    # dataset items (data_utils.collate_batch, straight into a page-locked stage of the ring), uploads it
    # (data_utils.batch_to_device: non-blocking H2D, CLS rows expanded on the device) and runs parse_batch + the
    # iteration.  Nothing in it synchronises: the host runs ahead of the GPU exactly as in the resident loop.
    fresh = None
    if extras:
        hi = host_items(T, hp, B, Tin, Tsub, Tn, 4, seed=4321)
        ring = D._RING

        def fed_step(i):
            batch = D.collate_batch(hi[i % 4], ring.get(1))
            xf, yf = model.parse_batch(D.batch_to_device(batch))
            T.train_step(model, criterion, optimizer, xf, yf, hp, it)
            return batch

        for i in range(8):                                              # steady state: every batch shape through every stage of the ring
            hb0 = fed_step(i)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        host_ms = []
        tf0 = time.perf_counter()
        evs[0].record()
        for i in range(a.steps):
            h0 = time.perf_counter()
            fed_step(i)
            evs[i + 1].record()
            host_ms.append(round(1e3 * (time.perf_counter() - h0), 2))
        torch.cuda.synchronize()
        dtf = time.perf_counter() - tf0
        step_ms = [round(evs[i].elapsed_time(evs[i + 1]), 2) for i in range(a.steps)]           # GPU-side: end of step i-1 to end of step i
        med = sorted(step_ms)[len(step_ms) // 2]
        fresh = dict(ms_per_step=round(1e3 * dtf / a.steps, 2), value=round(B * Tn * a.steps / dtf, 1), unit="mel-frames/s",
                     step_ms=step_ms, ms_per_step_median=med, max_over_median=round(max(step_ms) / med, 3), host_ms_per_step=host_ms,
                     host_batches=4, bytes_uploaded_per_batch=int(sum(v.numel() * v.element_size() if torch.is_tensor(v) else v.nbytes
                                                                      for k, v in dict.items(hb0) if v is not None and k != "align")
                                                                  + sum(v.numel() * v.element_size() for v in (hb0.cls_rows or {}).values())),
                     note="every step: data_utils.collate_batch of the next B ragged items into a page-locked stage + batch_to_device "
                          "(non-blocking H2D; CLS vectors travel as [B,768] rows and are expanded on the device) + parse_batch (host-side "
                          "maxima, no .item()) + the training iteration; PCIe-inclusive, never `value`.  step_ms = GPU-side time between "
                          "the ends of consecutive steps (HIP events), host_ms_per_step = host time to collate + enqueue one step")

    # one extra, untimed, profiled step: HIP events around every per-step decoder kernel launch
    roof, kernels = None, None
    if rank == 0:
        L.prof_enable(8 * Tn + 64)
    T.train_step(model, criterion, optimizer, x, y, hp, it)      # every rank takes the step (it contains collectives)
    if rank == 0:
        prof = L.prof_collect()
        torch.cuda.synchronize()
        bf = a.dtype == "bf16"
        fl, by = decoder_step_flops(hp, B, Tin, Tsub), decoder_step_bytes(hp, B, Tin, Tsub, 2 if bf else 4)
        for k in ("chain_a_fwd", "chain_b_fwd", "chain_a_bwd", "chain_b_bwd"):
            by[k] = by[k + "_per_step"] * Tn                         # one launch = all Tn steps
        kernels = {}
        for k, (ms, n) in prof.items():
            if n == 0:
                continue
            avg_us = 1e3 * ms / n
            e = dict(launches=n, avg_us=round(avg_us, 2), total_ms=round(ms, 2))
            if k.startswith("chain_"):
                e["steps_per_launch"] = Tn
                e["us_per_step"] = round(avg_us / Tn, 2)
            if k in fl:
                e["tflops"] = round(fl[k] / (avg_us * 1e-6) / 1e12, 2)
            if k in by:
                e["gbs"] = round(by[k] / (avg_us * 1e-6) / 1e9, 1)
            kernels[k] = e
        dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
        # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this round (separate --pmc passes, gfx950
        # corrections: profiles/README.md); static, not measured in this run
        traffic, traffic_source = None, None
        lsa = a.attention == "lsa"
        for name in (("r03_pmc_hbm_traffic_lsa.json",) if lsa else ("r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json")):
            tpath = os.path.join(ROOT, "profiles", name)
            tkey = {"att_lstm_fwd": "lstm_step_fwd_bf16_grid131072", "dec_lstm_fwd": "lstm_step_fwd_bf16_grid65536",
                    "attention_fwd": "attention_step_fwd_grid131072", "attention_bwd": "attention_step_bwd_grid131072",
                    "att_lstm_bwd_gemm": "lstm_bwd_gemm_bf16_grid393216", "dec_lstm_bwd_gemm": "lstm_bwd_gemm_bf16_grid131072",
                    "att_lstm_bwd_pointwise": "lstm_bwd_pointwise_q_grid65536", "dec_lstm_bwd_pointwise": "lstm_bwd_pointwise_grid65536",
                    "chain_a_fwd": "chain_fwd_lsa" if lsa else "chain_fwd_sma", "chain_b_fwd": "chain_fwd_lstm",
                    "chain_a_bwd": "chain_bwd_lsa" if lsa else "chain_bwd_sma", "chain_b_bwd": "chain_bwd_lstm"}.get(dom)
            if not bf or a.attention not in ("sma", "lsa") or (B, Tin, Tsub) != (64, 100, 60) or not tkey or not os.path.exists(tpath):
                continue
            ent = json.load(open(tpath)).get(tkey)
            if ent:
                traffic = ent.get("hbm_bytes_per_launch")
                if traffic is not None and ent.get("steps_per_launch") and ent["steps_per_launch"] != Tn:
                    traffic = traffic * Tn / ent["steps_per_launch"]
                traffic_source = f"profiles/{name} (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not measured in this run)"
                break
        # fp32 operands: the recurrent step GEMMs sit just above the fp32-MFMA ridge (AI ~ 26 FLOP/B vs 20) -> MFMA bound;
        # bf16 operands: 16x the matrix rate -> every per-step kernel is bound by operand delivery (HBM / L2)
        if dom in fl and not bf:
            roof = dict(kernel=dom, bound="mfma", achieved=kernels[dom]["tflops"], peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=round(kernels[dom]["tflops"] / PEAK_F32_MFMA_TFLOPS, 4), traffic=traffic, traffic_source=traffic_source,
                        note="fp32 MFMA (v_mfma_f32_32x32x2_f32) peak; algorithmic FLOPs per launch / avg HIP-event duration")
        else:
            roof = dict(kernel=dom, bound="hbm", achieved=kernels[dom].get("gbs"), peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=round(kernels[dom].get("gbs", 0.0) / PEAK_HBM_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                        note="algorithmic bytes per launch (DESIGN.md section 3; a persistent chain launch = its per-step figure x the "
                             "steps it covers) / avg HIP-event duration; the events bracket the launch, so a few us of dispatch gap are "
                             "included; traffic = HBM-side bytes per launch from the PMC passes in profiles/ — a persistent chain keeps "
                             "weights, state and memory on chip, so its traffic is far below its algorithmic bytes")
    if world > 1:
        torch.distributed.barrier()
    if rank != 0:
        return
    frames = world * B * Tn * a.steps
    out = {
        "metric": "mel_frames_per_sec_train", "value": round(frames / dt, 1), "unit": "mel-frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"BERT_Tacotron2 default hparams ({a.attention.upper()}), full training iteration fwd+loss+bwd+clip+Adam, "
                               f"B={B}/GPU, {Tin} phones, {Tsub} sub-word tokens, {Tn} frames, 80-mel",
                   "global_batch": world * B, "frames_per_item": Tn, "parallelism": f"dp{world}",
                   "hip_kernels": "embeddings, encoder conv/BN + BiLSTM, converters, decoder (prenets, attention LSTMs, SMA, "
                                  "decoder LSTM, projections), postnet conv/BN: forward and backward; gradient-norm clip + Adam",
                   "torch_ops": "loss reductions, cat/transpose copies",
                   "persistent_chains": bool(L.get_chain()),
                   "precision": ("bf16 operands / fp32 accumulate for every GEMM (hoisted LSTM input halves, per-step recurrent "
                                 "LSTM GEMMs via bf16 weight/activation shadows, convolutions, projections, weight gradients); "
                                 "LSTM gates/state, BatchNorm statistics, attention energies/recurrences, master weights, "
                                 "gradients and Adam fp32") if a.dtype == "bf16" else "fp32 everywhere (the parity path)",
                   "recurrent_steps_bf16": bool(a.dtype == "bf16" and B <= 128)},
        "rccl_world_size": torch.distributed.get_world_size() if world > 1 else 1, "dist_backend": backend,
        "host": {"cpus_visible": os.cpu_count(), "cpu_quota": round(cpu_quota(), 1), "torch_threads": host_threads},
        "ranks": ranks,
        "loss": round(loss_val, 5),
        "roofline": roof, "kernels": kernels,
    }
    if a.dtype == "bf16" and B > 128:
        out["config"]["note"] = "B > 128: the recurrent per-step GEMMs fall back to fp32 operands (bf16 steps cover B <= 128)"
    if fresh is not None:
        out["fresh_batches"] = fresh
    if extras and a.dtype == "bf16":
        # parity mode on the same batch + how far the bf16 mode's outputs sit from it (eval forward, dropout off)
        def eval_outputs():
            model.eval()
            model.decoder.prenet_dropout = False
            with torch.no_grad():
                o = model(x)
            model.decoder.prenet_dropout = True
            model.train()
            return [t.float().clone() for t in o]
        o16 = eval_outputs()
        L.set_precision("f32")
        o32 = eval_outputs()
        names = ("mel", "mel_postnet", "gate", "align", "align_bert")
        out["bf16_error"] = {n: round(float((p - q).abs().max()), 5) for n, p, q in zip(names, o16, o32)}
        out["bf16_error"]["note"] = "max-abs, bf16-operand mode vs fp32 mode (both HIP), eval forward of the bench batch after the timed steps"
        T.train_step(model, criterion, optimizer, x, y, hp, it)
        torch.cuda.synchronize()
        n32 = max(2, min(a.steps, 3))
        t32 = time.perf_counter()
        for _ in range(n32):
            T.train_step(model, criterion, optimizer, x, y, hp, it)
        torch.cuda.synchronize()
        d32 = (time.perf_counter() - t32) / n32
        out["fp32"] = dict(ms_per_step=round(1e3 * d32, 2), value=round(B * Tn / d32, 1), unit="mel-frames/s", steps=n32,
                           note="t2_set_precision(0): exact fp32 GEMMs everywhere — the mode the 1e-4 parity contract is tested in")
        L.set_precision(a.dtype)
    if extras:
        # "decode steps/sec" half of BASELINE.json's metric: configs[3] (B=32, 1000 decoder steps, stop disabled)
        out["decode"] = decode_bench(model, hp, 32, Tin, Tsub, steps=1000, reps=2)
        out["gta"] = gta_bench(model, hp, T, 128, Tin, Tsub, Tn, reps=3)
        if a.dtype == "bf16":
            out["gemm"] = gemm_bench(B, Tn)
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)


def gemm_bench(B, Tn, reps=10):
    """The matrix-core side of the iteration: the two largest products of the decoder (hoisted decoder-LSTM input product
    [B*T x 3072] . [4096 x 3072]^T and its weight gradient [4096 x B*T] . [B*T x 3072]) through the bf16-source GEMM, HIP
    events on the launch stream; `kernel` = bf16 operand copies made beforehand (matrix kernel + split-K reduce alone)."""
    import ctypes as C
    from tacotron2_subword_amd import _lib as L
    BT = B * Tn
    g = torch.Generator(device="cuda").manual_seed(5)
    ws = torch.empty(192 << 20, device="cuda")
    res = {}
    for name, M, N, K, ta, tb in (("dec_lstm_input NT", BT, 4096, 3072, False, True), ("dec_lstm_dW TN", 4096, 3072, BT, True, False)):
        A = torch.randn((K, M) if ta else (M, K), device="cuda", generator=g)
        Bm = torch.randn((N, K) if tb else (K, N), device="cuda", generator=g)
        out = torch.empty(M, N, device="cuda")
        a = L.GemmArgs()
        a.A, a.B, a.C, a.M, a.N, a.K = A.data_ptr(), Bm.data_ptr(), out.data_ptr(), M, N, K
        a.sam, a.sak = (1, A.stride(0)) if ta else (A.stride(0), 1)
        a.sbn, a.sbk = (Bm.stride(0), 1) if tb else (1, Bm.stride(0))
        a.ldc, a.batch, a.alpha, a.beta = N, 1, 1.0, 0.0
        a.ws, a.ws_bytes, a.splitk = ws.data_ptr(), ws.numel() * 4, 0
        mt, mk = C.c_float(), C.c_float()
        L.check(L.lib().t2_prof_gemm(C.byref(a), reps, C.byref(mt), C.byref(mk), L.stream()))
        fl = 2.0 * M * N * K
        res[name] = dict(M=M, N=N, K=K, kernel_us=round(1e3 * mk.value, 1), with_staging_us=round(1e3 * mt.value, 1),
                         tflops=round(fl / mk.value / 1e9, 1), tflops_with_staging=round(fl / mt.value / 1e9, 1))
    best = max(v["tflops"] for v in res.values())
    return dict(bound="mfma", achieved=best, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(best / PEAK_BF16_TFLOPS, 4), products=res,
                note="256x256-tile LDS-DMA kernel (csrc/gemm.hip), random operands; peak = dense bf16 MFMA rate of "
                     "MI355X_MICROARCH.md; `with_staging` adds the fp32 -> bf16 operand casts of a cold call")


def gta_bench(model, hp, T, B, Tin, Tsub, Tn, reps):
    """BASELINE configs[4]: teacher-forced forward, eval mode, no_grad, B=128 per GPU (GTA.py:57-59)."""
    was = model.training
    model.eval()
    x, _ = model.parse_batch(T.synthetic_batch(hp, B, Tin, Tsub, Tn, seed=99))
    with torch.no_grad():
        model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            model(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    model.train(was)
    return dict(batch=B, frames=Tn, ms_per_forward=round(1e3 * dt, 2), frames_per_sec=round(B * Tn / dt, 1))


if __name__ == "__main__":
    main()
