"""Grouped invocation of a BatchNorm network.

One optimisation step of the reference calls the depth encoder 8 times and the pose
encoder 6 times on mutually independent inputs (reference: train.py:724-731, 745-747,
788-797, 830-868), each call normalising with *its own* batch statistics and -- under
SyncBatchNorm -- issuing its own collectives per layer (280 per direction per step).

Here the G independent inputs are interleaved along the batch dimension
(sample ``n = b*G + g``), so that a contiguous activation ``[B*G, C, H, W]`` is, without any
copy, also the tensor ``[B, G*C, H, W]``: batch normalisation over that folded view with the
affine parameters repeated G times computes exactly the per-call statistics of every call
in ONE fused batch-norm launch, and under SyncBatchNorm one collective per layer carries
the statistics of all G calls.  Convolutions, pooling and activations are per-sample and do
not care about the interleaving.  Running statistics are advanced as the G sequential
momentum updates the per-call form would perform, in group order.

State-dict keys and shapes are those of ``nn.BatchNorm2d`` (checkpoints interchange with the
reference's ``.pth`` files).
"""
from __future__ import annotations

import contextlib
import ctypes
import os

import torch
import torch.distributed as dist
import torch.nn as nn
from ..consts import const_tensor


def _count(kind):
    from ..parallel import count_collective
    count_collective(kind)


_BN_PLAN = os.environ.get("MVF_BN_PLAN", "1") != "0"      # developer knob: per-layer launches instead of the layer plan


def _native_vectors(*tensors):
    """The one-launch forms of the per-layer vector glue (`mvf_bn_tile / _fold_running / _untile`) apply to
    contiguous fp32 vectors on the HIP device; anything else (the CPU gloo tests, reduced precision) keeps the
    tensor-op formulation."""
    return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors)


def _nat():
    from .. import _native as nat
    return nat, torch.cuda.current_stream().cuda_stream


def merge_groups(tensors):
    """G tensors [B, ...] -> [B*G, ...] interleaved (sample n = b*G + g)."""
    return torch.stack(list(tensors), 1).flatten(0, 1)


def split_groups(t, groups):
    """[B*G, ...] interleaved -> tuple of G views [B, ...] (one autograd node: the backward is
    a single stack of the group gradients)."""
    return torch.unbind(t.view(t.shape[0] // groups, groups, *t.shape[1:]), 1)


def _alias(y, shape):
    """``y`` under another shape WITHOUT a view relation autograd knows of.  The output of the
    folded batch norm is followed by in-place ReLUs (torchvision's / the reference's blocks):
    on a differentiable ``view`` of the output that turns every layer's backward into
    AsStridedBackward + CopySlices -- five extra passes over the activation (measured: 4.2 ms of
    device copies + 0.8 ms of fills per ResNet18 step)."""
    return y.new_empty(0).set_(y.untyped_storage(), y.storage_offset(), shape)


class _FoldedBN(torch.autograd.Function):
    """Training-mode batch norm of ``x [B*G, C, ...]`` over its folded view ``[B, G*C, ...]``
    (ATen's own dispatch: MIOpen on a HIP device), returned in the shape of ``x``.

    The per-channel vectors are tiled ``groups`` times HERE (one stack + one repeat for weight,
    bias and both running statistics; the adjoint is one stack + one sum) instead of four
    ``repeat`` nodes and their four reductions per layer: an HRNet18 step has 325 of these
    layers and was bound by the host enqueueing ~22 tiny launches around each of them.
    Returns (y, tiled running mean, tiled running var) -- the tiled statistics hold, per group,
    one momentum update from the common starting value (folded by the caller)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, groups, momentum, eps, tiled=None):
        N, C = x.shape[0], x.shape[1]
        xv = x.view(N // groups, groups * C, *x.shape[2:])
        if tiled is not None:            # filled for every layer of the network when the grouped call began (_BnPlan)
            wG, bG, rmG, rvG = tiled.unbind(0)
        elif _native_vectors(weight, bias, running_mean, running_var):
            nat, st = _nat()
            tiled = torch.empty((4, groups * C), dtype=torch.float32, device=x.device)
            nat.check(nat.lib().mvf_bn_tile(nat.ptr(weight), nat.ptr(bias), nat.ptr(running_mean), nat.ptr(running_var),
                                            nat.ptr(tiled), C, groups, st), "bn_tile")
            wG, bG, rmG, rvG = tiled.unbind(0)
        else:
            wG, bG, rmG, rvG = torch.stack([weight, bias, running_mean, running_var]).repeat(1, groups).unbind(0)
        y, save_mean, save_var, reserve, impl = torch._batch_norm_impl_index(
            xv, wG, bG, rmG, rvG, True, momentum, eps, True)
        ctx.save_for_backward(xv, wG, rmG, rvG, save_mean, save_var, reserve)
        ctx.impl, ctx.eps, ctx.groups = impl, eps, groups
        ctx.mark_non_differentiable(rmG, rvG)
        # without this the engine hands backward() a zero tensor for each of the two tiled statistics:
        # two fill launches per layer (650 per step for an HRNet18 grouped call)
        ctx.set_materialize_grads(False)
        return _alias(y, x.shape), rmG, rvG

    @staticmethod
    def backward(ctx, gy, _grm, _grv):
        if gy is None:
            return (None,) * 9
        xv, wG, rm, rv, save_mean, save_var, reserve = ctx.saved_tensors
        gx, gw, gb = torch.ops.aten._batch_norm_impl_index_backward(
            ctx.impl, xv, gy.contiguous().view_as(xv), wG, rm, rv, save_mean, save_var, True, ctx.eps,
            [True, True, True], reserve)
        G = ctx.groups
        if _native_vectors(gw, gb):
            nat, st = _nat()
            Cn = gw.numel() // G
            gwb = torch.empty((2, Cn), dtype=torch.float32, device=gw.device)
            nat.check(nat.lib().mvf_bn_untile(nat.ptr(gw), nat.ptr(gb), nat.ptr(gwb), Cn, G, st), "bn_untile")
        else:
            gwb = torch.stack([gw, gb]).view(2, G, gw.numel() // G).sum(1)      # adjoint of the tiling
        return gx.view_as(gy), gwb[0], gwb[1], None, None, None, None, None, None


class _AllReduceSyncBN(torch.autograd.Function):
    """Synchronised batch norm with plain tensor ops and ONE all-reduce per direction
    (device-agnostic: this is what the CPU gloo tests run; on a HIP device ATen's fused
    batch-norm kernels are used instead, see _FusedSyncBN)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        C = x.shape[1]
        red = [0] + list(range(2, x.dim()))
        n_local = x.numel() // C
        xd = x.double()              # E[x^2] - mean^2 cancels: accumulate the moments in fp64
        stats = torch.cat([xd.sum(red), (xd * xd).sum(red), xd.new_full((1,), float(n_local))])
        dist.all_reduce(stats, group=group)
        _count("bn_all_reduce_fwd")
        n = stats[-1]
        mean = stats[:C] / n
        var = (stats[C:2 * C] / n - mean * mean).clamp_min(0.0)
        invstd = torch.rsqrt(var + eps).to(x.dtype)
        mean, var, n = mean.to(x.dtype), var.to(x.dtype), n.to(x.dtype)
        shape = [1, C] + [1] * (x.dim() - 2)
        xhat = (x - mean.view(shape)) * invstd.view(shape)
        ctx.save_for_backward(xhat, weight, invstd, n)
        ctx.group = group
        ctx.mark_non_differentiable(mean, var, n)
        return xhat * weight.view(shape) + bias.view(shape), mean, var, n

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gn):
        xhat, weight, invstd, n = ctx.saved_tensors
        C = xhat.shape[1]
        red = [0] + list(range(2, xhat.dim()))
        shape = [1, C] + [1] * (xhat.dim() - 2)
        s_gy, s_gyx = gy.sum(red), (gy * xhat).sum(red)
        both = torch.cat([s_gy, s_gyx])
        dist.all_reduce(both, group=ctx.group)
        _count("bn_all_reduce")
        m_gy, m_gyx = both[:C] / n, both[C:] / n
        gx = (gy - m_gy.view(shape) - xhat * m_gyx.view(shape)) * (weight * invstd).view(shape)
        return gx, s_gyx, s_gy, None, None


class _FusedSyncBN(torch.autograd.Function):
    """Synchronised batch norm on a HIP device from ATen's fused building blocks
    (torch.batch_norm_stats / batch_norm_gather_stats_with_counts / batch_norm_elemt and the
    two backward kernels): ONE all_gather of (mean, invstd) forward and ONE all_reduce
    of (sum dy, sum dy*xmu) backward per layer, both on RCCL.  Same arithmetic as
    nn.SyncBatchNorm, without its dependency on torch's private ``nn.modules._functions`` and
    without its per-layer host synchronisation (it filters empty ranks with a boolean mask;
    every rank here always holds the same non-empty batch, `drop_last=True`).

    ``x`` is the folded view ``[B, G*C, ...]`` of ``groups`` interleaved calls; ``weight`` / ``bias`` are the layer's
    ``[C]`` parameters.  Round 5: the vector glue of a layer was ten tiny launches each way (the element count as a
    fresh tensor, its gather and conversion, a differentiable ``repeat`` of weight and bias and the two reductions of
    its adjoint) -- 428 launches and +3 % per ResNet18 step on one GPU.  Now: the counts are a cached constant (every
    rank holds the same batch, so they need no gather), weight and bias are tiled by `mvf_bn_tile` (or come tiled
    from the layer plan) and their gradients untiled by `mvf_bn_untile`."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, group, world, out_shape, groups,
                tiled_wb):
        x = x.contiguous()
        GC = x.shape[1]
        if tiled_wb is not None:
            wG, bG = tiled_wb
        elif groups == 1:
            wG, bG = weight, bias
        else:
            wG, bG = weight.repeat(groups), bias.repeat(groups)        # (inside forward(): plain copies, no autograd nodes)
        mean, invstd = torch.batch_norm_stats(x, eps)
        combined = torch.cat([mean, invstd])
        if dist.get_backend(group) == "gloo":          # no all_gather_into_tensor on gloo
            parts = [torch.empty_like(combined) for _ in range(world)]
            dist.all_gather(parts, combined, group)
            allc = torch.stack(parts, 0)
        else:
            flat = torch.empty(world * combined.numel(), dtype=combined.dtype, device=combined.device)
            dist.all_gather_into_tensor(flat, combined, group)
            allc = flat.view(world, combined.numel())
        _count("bn_all_gather")
        mean_all, invstd_all = allc[:, :GC], allc[:, GC:]
        n = x.numel() // GC
        counts = const_tensor((float(n),) * world, x.device, mean.dtype)
        mean, invstd = torch.batch_norm_gather_stats_with_counts(
            x, mean_all.contiguous(), invstd_all.contiguous(), running_mean, running_var, momentum, eps,
            counts)
        ctx.save_for_backward(x, wG, mean, invstd, const_tensor((float(n),) * world, x.device, torch.int32))
        ctx.group, ctx.groups = group, groups
        return _alias(torch.batch_norm_elemt(x, wG, bG, mean, invstd, eps), out_shape)

    @staticmethod
    def backward(ctx, gy):
        x, wG, mean, invstd, counts = ctx.saved_tensors
        gy = gy.contiguous().view_as(x)
        sum_dy, sum_dy_xmu, gw, gb = torch.batch_norm_backward_reduce(
            gy, x, mean, invstd, wG, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
            ctx.needs_input_grad[2])
        gx = None
        if ctx.needs_input_grad[0]:
            C = sum_dy.shape[0]
            both = torch.cat([sum_dy, sum_dy_xmu])
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=ctx.group)
            _count("bn_all_reduce")
            sum_dy, sum_dy_xmu = torch.split(both, C)
            gx = torch.batch_norm_backward_elemt(gy, x, mean, invstd, wG, sum_dy, sum_dy_xmu, counts)
        G = ctx.groups
        if G > 1 and gw is not None and gb is not None:
            if _native_vectors(gw, gb):
                nat, st = _nat()
                Cn = gw.numel() // G
                gwb = torch.empty((2, Cn), dtype=torch.float32, device=gw.device)
                nat.check(nat.lib().mvf_bn_untile(nat.ptr(gw), nat.ptr(gb), nat.ptr(gwb), Cn, G, st), "bn_untile")
                gw, gb = gwb[0], gwb[1]
            else:
                gw, gb = gw.view(G, -1).sum(0), gb.view(G, -1).sum(0)
        elif G > 1:
            gw = gw.view(G, -1).sum(0) if gw is not None else None
            gb = gb.view(G, -1).sum(0) if gb is not None else None
        return gx, gw, gb, None, None, None, None, None, None, None, None, None


class GroupedBatchNorm2d(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` whose input may hold ``groups`` interleaved independent calls."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.groups = 1
        self._plan, self._tiled, self._called = None, None, False     # see _BnPlan
        self._plan_off = False       # the layer ran twice inside one grouped() call once: it stays out of the plan
        self.sync = False            # synchronise statistics across ranks (SyncBatchNorm semantics)
        self.force_sync = False      # take the synchronised branch even in a group of one (tests)
        self.process_group = None

    def _sync_bn(self, xv, w, b, rm, rv, world, out_shape, G=1, tiled_wb=None):
        """SyncBatchNorm over the folded view; updates rm/rv in place.  On a HIP device `w`, `b` are the layer's [C]
        parameters (tiled inside `_FusedSyncBN`: `tiled_wb` = the plan's tiled copies); on the CPU they come tiled."""
        if xv.is_cuda:
            group = self.process_group or dist.group.WORLD
            return _FusedSyncBN.apply(xv, w, b, rm, rv, self.eps, self.momentum, group, world, out_shape, G, tiled_wb)
        y, mean, var, n = _AllReduceSyncBN.apply(xv, w, b, self.eps, self.process_group)
        with torch.no_grad():
            rm.mul_(1 - self.momentum).add_(mean, alpha=self.momentum)
            rv.mul_(1 - self.momentum).add_(var * (n / (n - 1)), alpha=self.momentum)
        return y.view(out_shape)

    def forward(self, x):
        G = self.groups
        world = dist.get_world_size(self.process_group) if (self.sync and dist.is_initialized()) else 1
        sync = self.sync and (world > 1 or self.force_sync) and self.training
        if not self.training or (G == 1 and not sync):
            return super().forward(x)       # eval: running statistics are the same for every call
        if self.momentum is None or not self.track_running_stats:
            raise RuntimeError("GroupedBatchNorm2d needs momentum-tracked running statistics")
        N, C = x.shape[0], x.shape[1]
        if N % G:
            raise RuntimeError(f"batch {N} is not a multiple of the group count {G}")
        if not x.is_contiguous():
            x = x.contiguous()
        plan = self._plan                   # set by grouped() while a call with a prepared plan is running
        tiled = None
        if plan is not None and plan.tiled_for(self, G):
            if self._called:
                # second run inside one grouped() call (a block shared by two branches): the prepared buffer holds the
                # first run's update.  Fold that update now, run this and every later call of the layer on the per-layer
                # path, and keep the layer out of the network's plan from here on (no RuntimeError).
                self._fold_running(self._tiled[2], self._tiled[3], G, C)
                self._called, self._plan_off = False, True
            else:
                tiled = self._tiled
                self._called = True
        if sync:
            xv = x.view(N // G, G * C, *x.shape[2:])
            if tiled is not None:
                rm, rv = tiled[2], tiled[3]
            else:
                rm, rv = self.running_mean.repeat(G), self.running_var.repeat(G)
            if xv.is_cuda:
                # the fused function tiles weight / bias itself (or takes the plan's copies) and untiles their gradients
                y = self._sync_bn(xv, self.weight, self.bias, rm, rv, world, x.shape, G,
                                  (tiled[0], tiled[1]) if tiled is not None else None)
            else:
                # (CPU: weight and bias reach the synchronised function through a differentiable repeat)
                y = self._sync_bn(xv, self.weight.repeat(G), self.bias.repeat(G), rm, rv, world, x.shape)
        else:
            y, rm, rv = _FoldedBN.apply(x, self.weight, self.bias, self.running_mean, self.running_var, G,
                                        self.momentum, self.eps, tiled)
        if tiled is None:                   # (with a plan, every layer's fold runs in ONE launch when the call ends)
            self._fold_running(rm, rv, G, C)
        return y

    @staticmethod
    def fold_coefficients(m, G):
        """(c_g, beta) of `_fold_running` for momentum m and G groups."""
        c = [(1 - m) ** (G - 1 - g) for g in range(G)]
        return c, (1 - m) ** G - (1 - m) * sum(c)

    @torch.no_grad()
    def _fold_running(self, rm, rv, G, C):
        """rm/rv hold, per group, ONE momentum update from the common starting value:
        r_g = (1-m) r + m s_g.  The per-call form would apply the G updates in sequence:
        r <- (1-m)^G r + sum_g m (1-m)^(G-1-g) s_g.
        With upd_g = r_g this is r <- [(1-m)^G - (1-m) sum_g c_g] r + sum_g c_g upd_g,
        c_g = (1-m)^(G-1-g): ONE matrix-vector launch per statistic (it was eight element-wise
        launches)."""
        c, beta = self.fold_coefficients(self.momentum, G)
        if G <= 32 and _native_vectors(self.running_mean, self.running_var, rm, rv) and \
                self.num_batches_tracked.is_cuda and self.num_batches_tracked.dtype == torch.int64:
            nat, st = _nat()
            nat.check(nat.lib().mvf_bn_fold_running(nat.ptr(self.running_mean), nat.ptr(self.running_var), nat.ptr(rm),
                                                    nat.ptr(rv), (ctypes.c_float * G)(*c), float(beta), C, G,
                                                    nat.ptr(self.num_batches_tracked), st), "bn_fold_running")
            return
        coef = const_tensor(c, rm.device, rm.dtype)
        for run, upd in ((self.running_mean, rm), (self.running_var, rv)):
            run.addmv_(upd.view(G, C).t(), coef, beta=beta, alpha=1.0)
        self.num_batches_tracked += G


def convert_grouped_batchnorm(module, sync=False, process_group=None, force_sync=False):
    """Replace every ``nn.BatchNorm2d`` (or SyncBatchNorm) under ``module`` by a
    ``GroupedBatchNorm2d`` sharing its parameters and buffers; returns the module."""
    out = module
    if isinstance(module, (nn.BatchNorm2d, nn.SyncBatchNorm)) and not isinstance(module, GroupedBatchNorm2d):
        out = GroupedBatchNorm2d(module.num_features, module.eps, module.momentum, module.affine,
                                 module.track_running_stats)
        if module.affine:
            out.weight, out.bias = module.weight, module.bias
        out.running_mean, out.running_var = module.running_mean, module.running_var
        out.num_batches_tracked = module.num_batches_tracked
        out.training = module.training
    if isinstance(out, GroupedBatchNorm2d):
        out.sync, out.process_group, out.force_sync = bool(sync), process_group, bool(force_sync)
    for name, child in module.named_children():
        new = convert_grouped_batchnorm(child, sync, process_group, force_sync)
        if new is not child:
            out.add_module(name, new)
    return out


class _BnPlan:
    """The vector glue of ALL grouped batch-norm layers of a network as one launch per grouped call each way.

    Per layer a persistent ``_tiled [4, G*C]`` buffer (weight | bias | running_mean | running_var repeated G times)
    and one row of a device table (`mvf_bn_tile_many` / `mvf_bn_fold_many`, include/mvf_hotpath.h).  ``begin()`` fills
    every buffer from the current parameters and statistics; the layers' forwards hand their buffer to the
    batch-norm call (which advances rows 2, 3 in place, one momentum update per group); ``end()`` folds those rows into
    the running statistics and advances the step counters -- for the layers that ran; if some did not (a network
    with a conditional branch), the per-layer fold is used for the ones that did.  The table is rebuilt only when a
    parameter or buffer moved (``.to()``, a replaced tensor).  Plans live on the network module they belong to."""

    def __init__(self, bns, groups):
        self.bns, self.G = bns, groups
        self.key = self.table = None
        self.max_c = max(m.num_features for m in bns)
        self.live = False

    @classmethod
    def of(cls, module, bns, groups):
        """The plan of (module, groups), or None where it does not apply (CPU, mixed momenta / devices, G > 32)."""
        bns = [m for m in bns if not m._plan_off]
        if groups < 2 or groups > 32 or not bns or not _BN_PLAN:
            return None
        m0 = bns[0]
        if not all(m.training and m.affine and m.track_running_stats and m.momentum == m0.momentum and
                   m.momentum is not None and m.weight.device == m0.weight.device and
                   _native_vectors(m.weight, m.bias, m.running_mean, m.running_var) and
                   m.num_batches_tracked.is_cuda and m.num_batches_tracked.dtype == torch.int64 for m in bns):
            return None
        plans = module.__dict__.setdefault("_mvf_bn_plans", {})
        plan = plans.get(groups)
        if plan is None or plan.bns != bns:
            plan = plans[groups] = cls(bns, groups)
        return plan

    def _ensure_table(self):
        G = self.G
        for m in self.bns:
            if m._tiled is None or m._tiled.shape[1] != G * m.num_features or m._tiled.device != m.weight.device:
                m._tiled = torch.empty((4, G * m.num_features), dtype=torch.float32, device=m.weight.device)
        key = tuple(p for m in self.bns for p in (m.weight.data_ptr(), m.bias.data_ptr(), m.running_mean.data_ptr(),
                                                  m.running_var.data_ptr(), m._tiled.data_ptr(),
                                                  m.num_batches_tracked.data_ptr()))
        if key != self.key:
            rows = []
            for i, m in enumerate(self.bns):
                rows.append(list(key[6 * i:6 * i + 6]) + [m.num_features, 0])
            host = torch.tensor(rows, dtype=torch.int64).pin_memory()
            self.table = host.to(self.bns[0].weight.device, non_blocking=True)
            self._host = host           # keep the pinned source alive until the copy has run
            self.key = key

    def begin(self):
        self._ensure_table()
        nat, st = _nat()
        nat.check(nat.lib().mvf_bn_tile_many(nat.ptr(self.table), len(self.bns), self.max_c, self.G, st), "bn_tile_many")
        # the launch rewrote every layer's buffer through a raw pointer: tell autograd, so that a graph of an EARLIER
        # grouped() call that still holds views of it (weights saved for backward) fails loudly instead of
        # differentiating against the new values
        torch.autograd.graph.increment_version([m._tiled for m in self.bns])
        for m in self.bns:
            m._plan, m._called = self, False
        self.live = True

    def tiled_for(self, m, G):
        return self.live and G == self.G and m._tiled is not None and not m._plan_off

    def end(self):
        self.live = False
        called = [m for m in self.bns if m._called]
        for m in self.bns:
            m._plan = None
        if not called:
            return
        G = self.G
        if len(called) == len(self.bns):
            c, beta = GroupedBatchNorm2d.fold_coefficients(self.bns[0].momentum, G)
            nat, st = _nat()
            nat.check(nat.lib().mvf_bn_fold_many(nat.ptr(self.table), len(self.bns), self.max_c, (ctypes.c_float * G)(*c),
                                                 float(beta), G, st), "bn_fold_many")
            return
        for m in called:
            m._fold_running(m._tiled[2], m._tiled[3], G, m.num_features)


@contextlib.contextmanager
def grouped(module, groups):
    """Within the block, the BatchNorm layers of ``module`` treat their input as ``groups``
    interleaved independent calls."""
    bns = [m for m in module.modules() if isinstance(m, GroupedBatchNorm2d)]
    if groups > 1 and not bns and any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in module.modules()):
        raise RuntimeError("grouped(): convert the module with convert_grouped_batchnorm() first")
    for m in bns:
        m.groups = groups
    plan = _BnPlan.of(module, bns, groups)
    if plan is not None:
        plan.begin()
    try:
        yield module
    finally:
        for m in bns:
            m.groups = 1
        if plan is not None:
            plan.end()
