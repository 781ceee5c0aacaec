"""Data-parallel gradient averaging over RCCL (one process per GPU), replacing the reference's
hand-rolled wrapper (distributed.py:132-179) behind the same ``apply_gradient_allreduce(module)``.

Semantics kept: every state_dict tensor is broadcast from rank 0 at wrap time; after each
backward the gradients of all parameters that received one are summed over ranks and divided
by the world size; parameters that never receive a gradient (the dead decoder_rnn_bert,
model.py:197-199) keep ``grad is None``, so the optimizer skips them exactly as in the
reference; the module's class and state_dict keys do not change.

What differs (xGMI is point-to-point, 7 links per GPU; one ring is per-link bound):
  * gradients live in ONE pre-flattened fp32 arena (``param.grad`` become views into it on
    first use), so there is no flatten / copy-back (the reference makes two extra 207 MB
    copies per step, distributed.py:163-167);
  * the arena is reduced bucket by bucket, each launched from an autograd hook as soon as
    the bucket's last gradient has been accumulated, on RCCL's stream, overlapping the rest of
    backward (the reference issues a single all-reduce after the whole backward);
  * the 1/N scale is applied to the bucket right before its collective.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

_BUCKET_BYTES = 32 << 20


class GradArena:
    """Flat fp32 gradient storage + bucket plan (reverse parameter order ~ readiness order)."""

    def __init__(self, module, bucket_bytes=_BUCKET_BYTES):
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self.span, off = {}, 0
        for p in self.params:
            self.span[p] = (off, off + p.numel())
            off += p.numel()
        self.buckets, cur, cur_bytes = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {p: i for i, b in enumerate(self.buckets) for p in b}
        self.ranges = [(min(self.span[p][0] for p in b), max(self.span[p][1] for p in b)) for b in self.buckets]
        self.live = set()            # parameters that have produced a gradient at least once
        self.exposed = None          # a list: record (event, event) around the end-of-backward waits (measurement only)
        self.launch_events = None    # a list: record (bucket, event) when a bucket's reduction is launched (measurement only)
        self.reset()

    def view(self, p):
        lo, hi = self.span[p]
        return self.flat[lo:hi].view_as(p)

    def adopt(self, p):
        """Make p.grad a view of the arena (keeps the value autograd just accumulated)."""
        v = self.view(p)
        if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v
        self.live.add(p)

    def reset(self):
        # the parameters each bucket waits for this step: those known to be live NOW.  A parameter that produces its
        # first gradient later (a layer unfrozen mid-run) is not in the snapshot: it must not count a bucket down, and
        # its bucket is reduced by finish() at the end of that backward instead of from a hook.
        self.counted = set(self.live)
        self.late = False
        self.pending = [sum(1 for p in b if (not self.live) or p in self.live) for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.handles, self.queued = [], False

    def zero(self):
        self.flat.zero_()


def apply_gradient_allreduce(module):
    """Same contract as distributed.py:132-179; returns the same module object."""
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("apply_gradient_allreduce needs an initialised process group (train.init_distributed)")
    for t in module.state_dict().values():          # distributed.py:138-141
        if torch.is_tensor(t):
            dist.broadcast(t, 0)
    if getattr(module, "_t2_arena", None) is not None:
        return module                               # the reference wraps twice (train.py:81,219); hooks exist already
    world = dist.get_world_size()
    dec = getattr(module, "decoder", None)
    if dec is not None and hasattr(dec, "defer_weight_grads"):
        dec.defer_weight_grads = False              # the hooks below read gradients during backward: no side-stream tail
    arena = GradArena(module)
    module._t2_arena = arena
    module.needs_reduction = False

    def launch(i):
        lo, hi = arena.ranges[i]
        seg = arena.flat[lo:hi]
        # the model may run part of its backward on a side stream (model.BERT_Tacotron2._fronts): a bucket can hold
        # gradients accumulated on either stream, so the launching stream first waits for the other one
        side = getattr(module, "_t2_side", None)
        if side is not None and seg.is_cuda:
            cur = torch.cuda.current_stream()
            for st in (side, torch.cuda.default_stream()):
                if st != cur:
                    cur.wait_stream(st)
        if arena.launch_events is not None and seg.is_cuda:
            ev = torch.cuda.Event(enable_timing=True); ev.record()
            arena.launch_events.append((i, hi - lo, ev))
        seg.mul_(1.0 / world)
        arena.handles.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
        arena.launched[i] = True

    def finish():
        if not module.needs_reduction:
            return
        module.needs_reduction = False
        for i in range(len(arena.buckets)):          # first step / partially-dead buckets
            if not arena.launched[i] and any(p in arena.live for p in arena.buckets[i]):
                launch(i)
        # measurement (bench.py --gpus N): how long the launch stream sits in these waits = the part of the reductions that
        # backward did not hide; a pair of events per step, read after the timed loop
        rec = arena.exposed is not None and arena.flat.is_cuda
        if rec:
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
        for h in arena.handles:
            h.wait()
        if rec:
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            arena.exposed.append((e0, e1))
        arena.handles = []

    def make_hook(p):
        def hook(_):
            i = arena.bucket_of[p]
            newcomer = module.needs_reduction and arena.first_done and p not in arena.counted
            if newcomer and arena.launched[i]:
                # first gradient of a parameter nobody waited for, and its bucket is already being reduced: the copy into
                # the arena must not race with that collective, and the parameter gets a reduction of its own
                for h in arena.handles:
                    h.wait()
                arena.handles = []
                arena.adopt(p)
                seg = arena.view(p)
                seg.mul_(1.0 / world)
                arena.handles.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
                return
            arena.adopt(p)
            if not module.needs_reduction:
                return
            if newcomer:
                arena.late = True                    # its bucket has not started: finish() reduces it, no more launches from hooks
            else:
                arena.pending[i] -= 1
            if arena.pending[i] == 0 and not arena.launched[i] and arena.first_done and not arena.late:
                launch(i)
            if not arena.queued:
                arena.queued = True
                torch.autograd.Variable._execution_engine.queue_callback(finish)
        return hook

    for p in arena.params:
        p.register_post_accumulate_grad_hook(make_hook(p))

    # The decoder's persistent kernels need every CU of the device: a reduction launched by an earlier backward node (the
    # postnet's bucket) would otherwise hold some of them while the grid waits to become resident.  Work.wait() orders the
    # launch stream behind the collective without blocking the host; finish() waits on the same handles again (harmless).
    import weakref
    from . import ops as _ops
    mref = weakref.ref(module)

    def before_persistent():
        m = mref()
        if m is None:                                # the wrapped module is gone: retire this entry
            _ops.PRE_PERSISTENT.remove(before_persistent)
            return
        if m.needs_reduction:
            for h in m._t2_arena.handles:
                h.wait()

    _ops.PRE_PERSISTENT.append(before_persistent)

    arena.first_done = False

    def pre_forward(mod, inputs):                    # distributed.py:175-178
        if mod.needs_reduction is False and arena.live:
            arena.first_done = True                  # the set of live parameters is known after one backward
        mod.needs_reduction = True
        arena.reset()

    module.register_forward_pre_hook(pre_forward)

    def zero_grad(set_to_none=False):                # keeps param.grad pointing into the arena
        arena.zero()

    module.zero_grad = zero_grad
    return module
