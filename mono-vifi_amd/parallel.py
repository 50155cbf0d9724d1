"""Data-parallel plumbing: one process per GPU, gradients averaged by RCCL over xGMI.

The hot path shards by batch with no data-path collective; the one
exchange step of an optimisation step is the gradient all-reduce.  Instead of wrapping
each of the reference's seven sub-models in its own ``DistributedDataParallel``
(reference: train.py:205-208 -- two wrappers around the SAME module when
``fuse_model_type=shared_encoder``, ``find_unused_parameters=True`` for the dead ImageNet
``fc``, a buffer broadcast on every one of ~32 forwards per step), one reducer owns all
trainable parameters:

* parameters are de-duplicated by identity and laid out, in reverse registration order
  (the order backward produces them), in a few large flat fp32 buckets -- xGMI is
  point-to-point, so fewer / larger collectives win (bucket default 32 MB);
* ``zero_grad()`` sets every ``param.grad`` to None, so autograd's accumulation node *keeps* the gradient tensor
  backward produced instead of adding it into a zeroed buffer: no launch per parameter (round 3 kept
  ``param.grad`` aliased to a zeroed bucket -- ~200 tiny ``add_`` launches per ResNet18 step, ~1,100 per HRNet18
  step, plus the memsets);
* a post-accumulate-grad hook counts arrivals; when a bucket is complete its gradients are packed into the
  flat buffer with ONE multi-tensor copy, ``param.grad`` becomes the view into the bucket and the
  ``all_reduce`` is issued asynchronously (RCCL runs it on its own stream) and overlaps the rest of backward;
  ``finish()`` waits, and also reduces buckets whose parameters got no gradient this step (their slices are
  zero, so every rank issues the same collectives -- no ``find_unused_parameters`` graph walk).  With nothing to
  exchange (one process, no forced collectives) nothing is packed: the gradients stay where backward put them
  and a parameter without a gradient keeps ``grad = None`` (the optimiser skips it, as in the reference's
  single-process run);
* works unchanged on ``gloo`` (CPU) -- that is how tests/test_parallel.py covers N > 1;
* ``exchange="reduce_scatter"`` replaces each bucket's all-reduce by the two halves a ring
  all-reduce consists of, issued explicitly on the flat buffer: ``reduce_scatter_tensor``
  (each rank receives the averaged 1/N-th of the bucket it owns, in place) followed by
  ``all_gather_into_tensor``.  Same bytes on the links, same result
  (the reduction order per element is RCCL's in both forms); what it buys is a point between
  the halves where a rank holds exactly its shard -- the place a sharded optimiser step goes --
  and two smaller collectives per bucket for the scheduler to interleave with backward.

Every collective this package issues is counted in ``COMM_COUNTS`` (by kind), so that "one
collective per BatchNorm layer per grouped call" and "one exchange per bucket" are tested
numbers (tests/test_parallel.py, tests/test_trainer_gpu.py) and ``bench.py`` can report
collectives per step.
"""
from __future__ import annotations

import collections
import warnings

import torch
import torch.distributed as dist

# kind -> number of collectives issued since reset_comm_counts(); kinds: grad_all_reduce,
# grad_reduce_scatter, grad_all_gather, bn_all_gather, bn_all_reduce, broadcast, loss_all_reduce
COMM_COUNTS = collections.Counter()


import os as _os
_BUCKET_VIEWS = _os.environ.get("MVF_GRAD_BUCKET_VIEWS", "0") == "1"


def count_collective(kind, n=1):
    COMM_COUNTS[kind] += n


def reset_comm_counts():
    COMM_COUNTS.clear()


def comm_counts():
    return dict(COMM_COUNTS)


def init_distributed(opts, backend=None):
    """Initialise the default process group from the torchrun environment (or the
    reference's --local_rank / --world_size flags).  Returns (rank, world_size)."""
    import os
    if torch.cuda.is_available():
        # one process per GPU: make this rank's GPU the current device before any stream,
        # allocation or communicator is created (reference: train.py torch.cuda.set_device)
        torch.cuda.set_device(opts.local_rank % torch.cuda.device_count())
    if opts.world_size <= 1:
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("MVF_DIST_BACKEND") or (
            "nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", opts.local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, init_method="env://",
                                world_size=opts.world_size, rank=opts.global_rank, **kw)
    return dist.get_rank(), dist.get_world_size()


def unique_parameters(modules):
    """Trainable parameters of an iterable of modules, de-duplicated by identity (the
    reference appends the aliased encoder_mf parameters twice, train.py:198-200)."""
    seen, out = set(), []
    for m in modules:
        for p in m.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


def unique_modules(models):
    """name -> module with aliases removed (first name wins)."""
    seen, out = set(), {}
    for k, m in models.items():
        if id(m) not in seen:
            seen.add(id(m))
            out[k] = m
    return out


@torch.no_grad()
def broadcast_module_states(modules, src=0):
    """Make every rank start from rank ``src``'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    tensors, seen = [], set()
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            if id(t) not in seen and t.is_floating_point():
                seen.add(id(t))
                tensors.append(t)
    by_dev = {}
    for t in tensors:
        by_dev.setdefault((t.device, t.dtype), []).append(t)
    for group in by_dev.values():
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src)
        count_collective("broadcast")
        off = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class _Bucket:
    """One flat exchange buffer.  `buf` / `views` / `shard` are allocated the first time the bucket is exchanged (or its
    views are asked for): a run that exchanges nothing (one process, no forced collectives) holds no copy of the
    parameters' size."""
    __slots__ = ("_buf", "params", "_views", "pending", "handle", "launched", "_shard", "numel", "unused", "_rank", "_world")

    def __init__(self, params, numel, rank, world):
        self._buf, self.params, self._views = None, params, None
        self.pending, self.handle, self.launched = len(params), None, False
        self._shard = None      # reduce_scatter exchange: this rank's 1/N-th of `buf` (a view)
        self.numel, self._rank, self._world = numel, rank, world
        self.unused = []        # parameters that brought no gradient to the current step's exchange

    def _materialise(self):
        buf = torch.zeros(self.numel, dtype=torch.float32, device=self.params[0].device)
        views, off = [], 0
        for p in self.params:
            views.append(buf[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._buf, self._views = buf, views
        if self._world is not None:
            per = self.numel // self._world
            self._shard = buf[self._rank * per:(self._rank + 1) * per]

    @property
    def buf(self):
        if self._buf is None:
            self._materialise()
        return self._buf

    @property
    def views(self):
        if self._views is None:
            self._materialise()
        return self._views

    @property
    def shard(self):
        if self._buf is None:
            self._materialise()
        return self._shard


class BucketedGradReducer:
    """Flat-bucket gradient averaging overlapped with backward (see module docstring)."""

    EXCHANGES = ("all_reduce", "reduce_scatter")

    def __init__(self, params, world_size=None, bucket_mb=32.0, process_group=None,
                 always_reduce=False, exchange="all_reduce", overlap=True, tail_mb=4.0):
        if exchange not in self.EXCHANGES:
            raise ValueError(f"exchange must be one of {self.EXCHANGES}, got {exchange!r}")
        self.group = process_group
        # always_reduce: issue the collectives even for a group of one (RCCL smoke on one GPU)
        self.always_reduce = bool(always_reduce)
        self.exchange = exchange
        # overlap=False: nothing is issued from the hooks, finish() reduces every bucket after
        # backward (the measurement bench.py --no-overlap puts beside the overlapped step)
        self.overlap = bool(overlap)
        self.world = world_size if world_size is not None else (
            dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self._rank = dist.get_rank(process_group) if (dist.is_initialized() and self.world > 1) else 0
        self.params = list(params)
        self.buckets = []
        self._owner = {}
        cap = max(int(bucket_mb * (1 << 20) // 4), 1)
        groups, cur, cur_n = [], [], 0
        for p in reversed(self.params):          # backward order
            if cur and (cur_n + p.numel() > cap or p.device != cur[0].device):
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        # The bucket that fills LAST goes out when backward is over: nothing is left to hide its exchange
        # behind (measured on one GPU: issued 0.04 ms before the end of backward, the others 14-40 ms).  Keep
        # that exposed tail short: the parameters arriving last get a small bucket of their own.
        tail_cap = max(int(tail_mb * (1 << 20) // 4), 1)
        last = groups[-1] if groups else []
        if tail_mb > 0 and len(last) > 1 and sum(p.numel() for p in last) > 2 * tail_cap:
            tail, n = [], 0
            while len(last) > 1 and n + last[-1].numel() <= tail_cap:
                n += last[-1].numel()
                tail.insert(0, last.pop())
            if tail:
                groups.append(tail)
        for g in groups:
            self._seal(g)
        # where the exchanges of the current step were issued from: a post-accumulate-grad hook (= during
        # backward, the overlapped case) or finish() (= after it); reset by zero_grad()
        self.issued_from_hook = 0
        self.issued_from_finish = 0
        self._warned_exposed = False
        # timeline=True (GPU): an event on the compute stream at every exchange issued from a hook and one
        # when finish() is reached (= the end of backward's enqueued work) -- `timeline_ms()` then says how much
        # of the backward pass was still ahead when each bucket went out (evidence of the overlap; off by default)
        self.timeline = False
        self._ev_issue, self._ev_end = [], None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    @property
    def exchanging(self):
        """Whether a step issues collectives at all (N > 1, or a group of one forced through them)."""
        return self.world > 1 or self.always_reduce

    def _seal(self, params):
        n = sum(p.numel() for p in params)
        # reduce_scatter exchange: every rank owns an equal slice, so the flat buffer is padded
        # to a multiple of the group size (the padding stays zero: nothing ever writes it)
        n_pad = -(-n // self.world) * self.world if self.exchange == "reduce_scatter" else n
        for p in params:
            p.grad = None
        b = _Bucket(params, n_pad, self._rank, self.world if self.exchange == "reduce_scatter" else None)
        for p in params:
            self._owner[id(p)] = b
        self.buckets.append(b)

    # ---------------------------------------------------------------- step protocol
    def zero_grad(self):
        """Replaces optimizer.zero_grad(set_to_none=True): every gradient starts the step as None, so the
        accumulation node keeps the tensor backward hands it (no add into a zeroed buffer, no memset)."""
        self.issued_from_hook = self.issued_from_finish = 0
        self._ev_issue, self._ev_end = [], None
        for b in self.buckets:
            b.pending, b.handle, b.launched = len(b.params), None, False
            b.unused = []
            if _BUCKET_VIEWS:       # round-3 form, kept as a developer knob for same-box A/B timing
                b.buf.zero_()
                for p, v in zip(b.params, b.views):
                    p.grad = v
                continue
            for p in b.params:
                p.grad = None

    def _pack(self, b):
        """Gradients of a bucket -> its flat buffer (one multi-tensor copy; a parameter that got no gradient
        contributes zeros), and ``param.grad`` -> the view, so that what the collective averages in place is
        what clipping and the optimiser read.  Parameters without a gradient are remembered: `finish()` hands
        them back as ``grad = None`` (see there)."""
        src, dst = [], []
        for p, v in zip(b.params, b.views):
            g = p.grad
            if g is None:
                v.zero_()
                b.unused.append(p)
            elif g.data_ptr() != v.data_ptr():
                src.append(g if g.dtype == v.dtype else g.to(v.dtype))
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v in zip(b.params, b.views):
            p.grad = v

    def _chained(self):
        """True when two collectives issued back to back on the group run in issue order
        (RCCL: one communicator stream).  gloo's worker threads give no such guarantee."""
        return dist.get_backend(self.group) == "nccl"

    def _launch(self, b):
        b.launched = True
        if not self.exchanging:
            return
        with torch.no_grad():
            self._pack(b)
        b.buf.div_(self.world)
        if self.exchange == "all_reduce":
            b.handle = dist.all_reduce(b.buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            count_collective("grad_all_reduce")
            return
        # in place: rank r's output is slice r of its own input (RCCL's in-place form)
        h = dist.reduce_scatter_tensor(b.shard, b.buf, op=dist.ReduceOp.SUM, group=self.group,
                                       async_op=True)
        count_collective("grad_reduce_scatter")
        if self._chained():
            h = dist.all_gather_into_tensor(b.buf, b.shard, group=self.group, async_op=True)
            count_collective("grad_all_gather")
        b.handle = h

    def _on_grad(self, p):
        b = self._owner[id(p)]
        b.pending -= 1
        if b.pending == 0 and not b.launched and self.overlap:
            if self.timeline and b.params[0].is_cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                self._ev_issue.append(ev)
            self._launch(b)
            self.issued_from_hook += 1

    def finish(self):
        """Call after ``loss.backward()``: reduces the buckets that never filled (unused
        parameters) and waits for every collective.
        A parameter that received no gradient in this step leaves with ``grad = None`` -- as it does in a run
        without an exchange, and as the reference's ``DistributedDataParallel(find_unused_parameters=True)``
        (train.py:208) leaves a globally unused one -- so AdamW neither decays it nor ages its moments at N = 8 while
        it does not at N = 1.  Data-parallel ranks run the same graph, so the set is the same on every
        rank; its zero slot still takes part in the collective (the bucket layout is static).
        CONTRACT: the set of parameters that receive a gradient must be the same on every rank.  The hooks
        issue the bucket collectives in the order the buckets FILL, and a communicator matches collectives by issue
        order: a parameter used on some ranks only (a data-dependent branch) changes that order between ranks and the
        exchange mismatches -- under gloo it fails loudly ("collective mismatch"), it is not silently wrong.  The
        reference's DDP tolerates it by walking the graph before backward; none of its models needs it
        (train.py:698-886 runs the same modules on every rank)."""
        if self.timeline and self.buckets and self.buckets[0].params[0].is_cuda:
            self._ev_end = torch.cuda.Event(enable_timing=True)
            self._ev_end.record()
        for i, b in enumerate(self.buckets):
            if not b.launched:
                if self.overlap and self.exchanging and b.pending > 0 and not self._warned_exposed:
                    # a bucket that holds a parameter without a gradient never fills during backward: its exchange
                    # goes out here, fully exposed (nothing left to hide it behind).  None of the shipped networks
                    # has such a parameter; say so ONCE for whoever adds one
                    self._warned_exposed = True
                    warnings.warn(
                        f"BucketedGradReducer: bucket {i} of {len(self.buckets)} ({b.numel * 4 / 2 ** 20:.1f} MB, "
                        f"{len(b.params)} parameters) is exchanged AFTER backward because {b.pending} of its parameters "
                        "received no gradient this step; its collective is not overlapped.  Give unused parameters a "
                        "bucket of their own (or drop them from the optimiser) to keep the exchange hidden.",
                        RuntimeWarning, stacklevel=2)
                self._launch(b)
                self.issued_from_finish += 1
        gather_late = self.exchange == "reduce_scatter" and (self.world > 1 or self.always_reduce) \
            and not self._chained()
        for b in self.buckets:
            if b.handle is not None:
                b.handle.wait()
                b.handle = None
                if gather_late:
                    dist.all_gather_into_tensor(b.buf, b.shard, group=self.group)
                    count_collective("grad_all_gather")
        for b in self.buckets:
            for p in b.unused:
                p.grad = None
            b.unused = []

    def timeline_ms(self):
        """timeline=True: per exchange issued from a hook, the GPU time between its issue point and the end
        of the backward pass on the compute stream (ms; synchronises)."""
        if self._ev_end is None:
            return []
        self._ev_end.synchronize()
        return [round(ev.elapsed_time(self._ev_end), 3) for ev in self._ev_issue]

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    @property
    def num_buckets(self):
        return len(self.buckets)

    @property
    def total_bytes(self):
        return sum(b.numel for b in self.buckets) * 4

    def bucket_view(self, p):
        """The slice of its bucket a parameter's gradient is packed into (and, after the exchange, is)."""
        b = self._owner[id(p)]
        return b.views[[id(q) for q in b.params].index(id(p))]
