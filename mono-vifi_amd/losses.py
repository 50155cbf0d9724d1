"""The three ``Trainer`` loss methods on the hot path, with the reference's signatures.

reference: train.py:956-1051 (``generate_images_pred``, ``compute_reprojection_loss``,
``compute_losses_base``).  They are written as a mixin so that ``Trainer`` (trainer.py)
inherits them and tests can call them unbound on a light-weight object carrying ``opt``
exactly as the golden-vector capture does with the reference's own methods.

Two ways through the path:

* **staged** (drop-in): ``generate_images_pred`` materialises the warped image
  (``mvf_warp_fwd``), ``compute_losses_base`` consumes it (``mvf_photo_fwd``) -- same
  call sequence and intermediate tensors as the reference.
* **fused** (``compute_unit``): one forward and one backward kernel per unit with the
  warped images held in LDS only (``mvf_unit_fwd`` / ``mvf_unit_bwd``); this is what
  ``Trainer.process_batch`` uses.
"""
from __future__ import annotations

import torch

from . import ops


class HotPathLosses:
    """Mixin; expects ``self.opt`` with min_depth, max_depth, no_ssim, avg_reprojection,
    disable_automasking, disparity_smoothness (reference: options.py:84-105,190-200)."""

    # ------------------------------------------------------------------ helpers
    def _loss_flags(self):
        o = self.opt
        return ops._flags(o.no_ssim, o.avg_reprojection, o.disable_automasking)

    def _tie_break_noise(self, disp, num_src):
        """The draw of reference train.py:1023-1024 (before its 1e-5 scale).  Tests
        inject a fixed tensor through ``self.tie_break_noise``."""
        if self.opt.disable_automasking:
            return None
        B, _, H, W = disp.shape
        n_id = 1 if self.opt.avg_reprojection else num_src
        fixed = getattr(self, "tie_break_noise", None)
        if fixed is not None:
            if tuple(fixed.shape) != (B, n_id, H, W):
                raise RuntimeError(f"tie_break_noise must be {(B, n_id, H, W)}, got {tuple(fixed.shape)}")
            return fixed
        return torch.randn((B, n_id, H, W), device=disp.device)

    # ------------------------------------------------------------------ reference API
    def generate_images_pred(self, disp_tgt, pose_tgt_src, img_src, K, inv_K):
        """Warp ``img_src`` into the target view; reference: train.py:956-971."""
        disp = disp_tgt[("disp", 0)]
        return ops.Warp.apply(disp, pose_tgt_src, img_src, K, inv_K,
                              self.opt.min_depth, self.opt.max_depth, 1e-7)

    def compute_reprojection_loss(self, pred, target):
        """0.85*SSIM + 0.15*L1 map [B,1,H,W]; reference: train.py:973-985."""
        return ops.Reprojection.apply(pred, target, bool(self.opt.no_ssim))

    def compute_losses_base(self, disp_tgt, img_tgt, imgs_src_tgt, imgs_src, mask_rec=None):
        """Min-reprojection + auto-mask + smoothness; reference: train.py:987-1051.
        Returns (loss, auto_mask | None)."""
        disp = disp_tgt[("disp", 0)]
        S = len(imgs_src_tgt)
        o = self.opt
        if mask_rec is not None and o.disable_automasking and (o.avg_reprojection or S == 1) \
                and disp.shape[0] > 1:
            # the reference's in-place `to_optimise *= mask_rec[:,0]` cannot broadcast a
            # [B,1,H,W] map against [B,H,W] (train.py:1030-1036)
            raise RuntimeError("output with shape [B, 1, H, W] doesn't match the broadcast shape")
        noise = self._tie_break_noise(disp, S)
        srcs = list(imgs_src) if not o.disable_automasking else []
        loss, auto_mask, _, _ = ops.LossesBase.apply(
            disp, img_tgt, mask_rec, noise, S, self._loss_flags(), float(o.disparity_smoothness),
            *imgs_src_tgt, *srcs)
        return loss, (None if o.disable_automasking else auto_mask)

    # ------------------------------------------------------------------ fused units
    def compute_unit(self, disp_tgt, img_tgt, poses, imgs_src, K, inv_K, mask_rec=None,
                     want_auto_mask=False):
        """``len(poses)`` x generate_images_pred + compute_losses_base in one fused
        forward kernel (and one fused backward).  ``poses``: list of [B,4,4] or a stacked
        [S,B,4,4] tensor.  Returns (loss, auto_mask | None)."""
        disp = disp_tgt[("disp", 0)]
        o = self.opt
        T = poses if torch.is_tensor(poses) else torch.stack(list(poses), 0)
        S = T.shape[0]
        # tie-break draw (train.py:1023-1024): generated inside the forward+backward tile kernel
        # unless a test injected a fixed tensor or the route with separate kernels runs
        in_kernel = (getattr(o, "inkernel_noise", True) and getattr(self, "tie_break_noise", None) is None
                     and ops.unit_uses_fwdbwd(S, disp, T))
        noise = None if in_kernel else self._tie_break_noise(disp, S)
        cfg = (S, self._loss_flags(), float(o.disparity_smoothness), o.min_depth, o.max_depth,
               1e-7, bool(want_auto_mask), False, None, disp_tgt.get(("disp_mean_partials", 0)))
        loss, auto_mask, _, _, _ = ops.Unit.apply(disp, img_tgt, T, K, inv_K, mask_rec, noise, cfg,
                                                  *imgs_src)
        return loss, (auto_mask if want_auto_mask and not o.disable_automasking else None)

    def compute_units(self, units, want_ident=False, want_auto_mask=False, want_sum=False, sum_in=None):
        """Several mutually independent units of one shape as ONE launch (reference: the three
        calls of each group in process_batch, train.py:747-760 / 795-810 / 837-882).

        ``units``: list of dicts with keys disp_tgt, img_tgt, poses, imgs_src, K, inv_K and
        optionally mask_rec, ident (the identity maps another unit with the same target and
        sources returned).  Returns (losses [n], idents list | None, auto_masks list | None); with
        ``want_sum`` the first element is the 0-dim SUM of the n losses instead (what process_batch adds to
        loss_base per group, train.py:760 / 812 / 882), written by the launch itself -- no reduction launch
        forward, no expand + copy of its gradient backward; ``sum_in`` (0-dim fp32 tensor on the device, with
        ``want_sum``): the running total the sum is added to, inside the same finishing kernel.  ``want_ident``:
        one flag for all units or one per unit.
        Falls back to one `compute_unit` per entry -- and then returns `idents = None`: no identity
        maps are handed over, the partner units re-evaluate them -- when the forward+backward kernel
        cannot take the group as one launch: `--batch_units False`, more than MAX_UNITS entries, S > 2,
        no gradient wanted, or entries of different shapes.  An injected `tie_break_noise` tensor does
        NOT force the fallback: the batched launch takes the noise as a tensor per unit."""
        o = self.opt
        n = len(units)
        prepared = []
        for un in units:
            disp = un["disp_tgt"][("disp", 0)]
            poses = un["poses"]
            T = poses if torch.is_tensor(poses) else torch.stack(list(poses), 0)
            prepared.append((disp, T))
        S = prepared[0][1].shape[0]
        batched = (n <= ops.nat.MAX_UNITS and getattr(o, "batch_units", True) and
                   all(ops.unit_uses_fwdbwd(S, d, T) and T.shape[0] == S and d.shape == prepared[0][0].shape
                       for d, T in prepared))
        if not batched:
            out = [self.compute_unit(un["disp_tgt"], un["img_tgt"], un["poses"], un["imgs_src"], un["K"],
                                     un["inv_K"], un.get("mask_rec"), want_auto_mask) for un in units]
            per_unit = torch.stack([l for l, _ in out])
            if want_sum:
                per_unit = per_unit.sum() if sum_in is None else sum_in + per_unit.sum()
            return per_unit, None, ([m for _, m in out] if want_auto_mask else None)
        in_kernel = getattr(o, "inkernel_noise", True) and getattr(self, "tie_break_noise", None) is None
        flat, mean_parts, sinks, tokens = [], [], [], []
        defer = bool(getattr(o, "defer_unit_grads", True))
        for un, (disp, T) in zip(units, prepared):
            noise = None if in_kernel else self._tie_break_noise(disp, S)
            # a disparity that is a view of its head's output (ops.disp_head): the unit reads it in place and leaves
            # its raw gradient to the head's adjoint kernel (ops.HeadSink) -- no scaled copy, no re-interleaving stack
            sink = un["disp_tgt"].get(("disp_head_sink", 0)) if defer else None
            where = sink.claim(disp) if (sink is not None and disp.requires_grad) else None
            if where is not None:
                sinks.append((sink, where[0], where[1]))
                if not any(t is sink.token for t in tokens):
                    tokens.append(sink.token)
                disp = disp.detach()
            else:
                sinks.append(None)
            flat += [disp, un["img_tgt"], T, un["K"], un["inv_K"], un.get("mask_rec"), noise,
                     un.get("ident"), *un["imgs_src"]]
            mean_parts.append(un["disp_tgt"].get(("disp_mean_partials", 0)))
        wid = list(want_ident) if isinstance(want_ident, (list, tuple)) else [bool(want_ident)] * n
        use_sum_in = bool(want_sum) and sum_in is not None
        if use_sum_in:
            flat.append(sum_in.float().reshape(()))
        cfg = dict(n=n, S=S, flags=self._loss_flags(), smoothness=float(o.disparity_smoothness),
                   min_depth=o.min_depth, max_depth=o.max_depth, eps=1e-7,
                   want_mask=bool(want_auto_mask), want_idx=False, want_ident=wid, want_sum=bool(want_sum),
                   sum_in=use_sum_in, sinks=sinks if tokens else None, n_tokens=len(tokens),
                   mean_parts=mean_parts if any(m is not None for m in mean_parts) else None)
        res = ops.Units.apply(cfg, *flat, *tokens)
        losses, per = (res[-1] if want_sum else res[0]), res[2:]
        idents = [per[4 * u + 3] if wid[u] else None for u in range(n)] \
            if any(wid) and not o.disable_automasking else None
        masks = [per[4 * u + 0] for u in range(n)] if want_auto_mask and not o.disable_automasking else None
        return losses, idents, masks
