"""-m gpu: several hot-path units in ONE launch (mvf_units_fwdbwd) against the same units launched
one at a time, against the oracle, and the identity-map hand-over between the two units of a
(single-frame, multi-frame) pair (reference: train.py:747-760, 795-810, 837-882).

Bars: batched == one-at-a-time bit for bit (argmin, loss, gradients: the folds inside the launch
run in a fixed order whoever performs them); identity maps == the oracle's bit for bit; a unit fed
with handed-over identity maps == the same unit evaluating them itself, bit for bit."""
import numpy as np
import pytest
import torch

from conftest import assert_grad_close, rel_err
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from mono_vifi_amd import _native
    _native.lib()
    return torch.device("cuda:0")


def T(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.requires_grad_(True) if grad else t


def N(t):
    return t.detach().cpu().numpy()


def _inputs(seed, B, H, W, flags, with_mask):
    from mono_vifi_amd import synthetic
    inp = synthetic.unit_inputs(seed, B, H, W, pose_scale=0.03, with_mask=True,
                                disp_mode="smooth" if H * W > 10000 else "noise")
    inp["T"] = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1))
                         for k in range(2)], 0)
    inp["noise_used"] = np.ascontiguousarray(inp["noise"][:, :1] if flags & 2 else inp["noise"])
    if not with_mask:
        inp["mask_rec"] = None
    return inp


def _cfg(n, flags, **kw):
    return dict(n=n, S=2, flags=flags, smoothness=1e-3, min_depth=0.1, max_depth=100.0, eps=1e-7,
                want_mask=True, want_idx=True, **kw)


def _flat(inp, dev, flags, disp=None, ident=None):
    d = disp if disp is not None else T(inp["disp"], dev, True)
    Tt = T(inp["T"], dev, True)
    noise = None if flags & 4 else T(inp["noise_used"], dev)
    mask = T(inp["mask_rec"], dev) if inp["mask_rec"] is not None else None
    return d, Tt, [d, T(inp["tgt"], dev), Tt, T(inp["K"], dev), T(inp["inv_K"], dev), mask, noise, ident,
                   T(inp["src"][0], dev), T(inp["src"][1], dev)]


@pytest.mark.parametrize("shape", [(2, 33, 70), (1, 64, 200), (12, 192, 640)])
@pytest.mark.parametrize("flags", [0, 1, 2, 4])
def test_batched_units_equal_single_launches(dev, shape, flags):
    """Three different units in one launch -- one of them reading its disparity out of an
    interleaved [B*G,1,H,W] tensor (image stride G*H*W, as a grouped decoder call leaves it) --
    against the same three launched one at a time; the first also against the oracle."""
    from mono_vifi_amd import ops
    B, H, W = shape
    use_mask = not (flags & 4 and flags & 2)
    inps = [_inputs(4100 + 13 * u + H + flags, B, H, W, flags, use_mask) for u in range(3)]
    # unit 1 takes a strided view
    G = 3
    inter = torch.zeros((B * G, 1, H, W), device=dev)
    inter.view(B, G, 1, H, W)[:, 1] = T(inps[1]["disp"], dev)
    inter.requires_grad_(True)
    view = torch.unbind(inter.view(B, G, 1, H, W), 1)[1]
    assert not view.is_contiguous() or B == 1

    flat, leaves = [], []
    for u, inp in enumerate(inps):
        d, Tt, f = _flat(inp, dev, flags, disp=view if u == 1 else None)
        flat += f
        leaves.append((d if u != 1 else inter, Tt))
    res = ops.Units.apply(_cfg(3, flags), *flat)
    losses, per = res[0], res[2:]
    losses.sum().backward()
    got = []
    for u in range(3):
        gd = leaves[u][0].grad
        if u == 1:
            gd = gd.view(B, G, 1, H, W)[:, 1]
        got.append((float(losses[u].detach()), N(per[4 * u + 1]), N(gd), N(leaves[u][1].grad), N(per[4 * u + 2]),
                    N(per[4 * u + 0])))
    if B > 1:       # nothing leaked into the other groups of the interleaved tensor
        other = inter.grad.view(B, G, 1, H, W)[:, [0, 2]]
        assert float(other.abs().max()) == 0.0

    for u, inp in enumerate(inps):
        d, Tt, f = _flat(inp, dev, flags)
        r1 = ops.Units.apply(_cfg(1, flags), *f)
        r1[0].sum().backward()
        one = (float(r1[0][0].detach()), N(r1[2 + 1]), N(d.grad), N(Tt.grad), N(r1[2 + 2]), N(r1[2 + 0]))
        assert got[u][0] == one[0], f"unit {u}: loss {got[u][0]} != {one[0]}"
        assert np.array_equal(got[u][1], one[1]), f"unit {u}: argmin"
        assert np.array_equal(got[u][2], one[2]), f"unit {u}: grad_disp"
        assert np.array_equal(got[u][3], one[3]), f"unit {u}: grad_T"
        assert np.array_equal(got[u][4], one[4]), f"unit {u}: sampling indices"
        assert np.array_equal(got[u][5], one[5]), f"unit {u}: auto mask"

    inp = inps[0]
    ref = O.unit(inp["disp"], inp["tgt"], inp["src"], inp["T"], inp["K"], inp["inv_K"], inp["noise_used"],
                 inp["mask_rec"], flags, want_grads=True)
    am = got[0][1].astype(np.int32)
    am[am == 255] = -1
    assert np.array_equal(am, ref["idx"])
    assert abs(got[0][0] - ref["loss"]) <= 1e-5 * abs(ref["loss"])
    assert_grad_close(got[0][2], ref["grad_disp"], TOL, "grad_disp vs oracle")
    assert rel_err(got[0][3], ref["grad_T"]) <= TOL
    for k in range(2):
        assert np.array_equal(got[0][4][k, ..., 0], ref["x0"][k]) and np.array_equal(got[0][4][k, ..., 1], ref["y0"][k])


@pytest.mark.parametrize("shape", [(2, 33, 70), (3, 17, 65), (12, 192, 640)])
@pytest.mark.parametrize("flags", [0, 1, 2])
def test_identity_maps_handed_over(dev, shape, flags):
    """Unit A (disparity 1) writes its identity maps; unit B (disparity 2; same target, sources,
    poses -- the multi-frame unit of the same target, train.py:795-797) takes them.  The maps are
    the oracle's identity-reprojection maps bit for bit, and B's results are bit-equal to B
    evaluating the identity pair itself and match the oracle."""
    from mono_vifi_amd import ops, synthetic
    B, H, W = shape
    inp = _inputs(5200 + H + flags, B, H, W, flags, True)
    other = synthetic.unit_inputs(77, B, H, W, disp_mode="smooth" if H * W > 10000 else "noise")
    inp_b = dict(inp)
    inp_b["disp"] = other["disp"]
    inp_b["noise_used"] = np.ascontiguousarray(other["noise"][:, :1] if flags & 2 else other["noise"])

    _, _, fa = _flat(inp, dev, flags)
    ra = ops.Units.apply(_cfg(1, flags, want_ident=True), *fa)
    ident = ra[2 + 3]
    assert tuple(ident.shape) == (B, H, W, 2)
    ref_a = O.unit(inp["disp"], inp["tgt"], inp["src"], inp["T"], inp["K"], inp["inv_K"], inp["noise_used"],
                   inp["mask_rec"], flags)
    idl = np.moveaxis(ref_a["idl"], 1, -1)          # [B,H,W,S]
    assert np.array_equal(N(ident), idl), "identity maps differ from the oracle's"

    out = []
    for handed in (ident, None):
        d, Tt, fb = _flat(inp_b, dev, flags, ident=handed)
        rb = ops.Units.apply(_cfg(1, flags), *fb)
        rb[0].sum().backward()
        out.append((float(rb[0][0].detach()), N(rb[2 + 1]), N(d.grad), N(Tt.grad), N(rb[2 + 0])))
    for a, b, what in zip(out[0], out[1], ("loss", "argmin", "grad_disp", "grad_T", "auto_mask")):
        assert np.array_equal(a, b), f"{what}: handed-over identity maps change the result"
    ref_b = O.unit(inp_b["disp"], inp["tgt"], inp["src"], inp["T"], inp["K"], inp["inv_K"], inp_b["noise_used"],
                   inp["mask_rec"], flags, want_grads=True)
    am = out[0][1].astype(np.int32)
    am[am == 255] = -1
    assert np.array_equal(am, ref_b["idx"])
    assert abs(out[0][0] - ref_b["loss"]) <= 1e-5 * abs(ref_b["loss"])
    assert_grad_close(out[0][2], ref_b["grad_disp"], TOL, "grad_disp vs oracle (handed-over identity maps)")
    assert rel_err(out[0][3], ref_b["grad_T"]) <= TOL


def test_per_unit_upstream_gradients_and_repeatability(dev):
    """backward() with a different upstream gradient per unit of a launch; two identical launches
    give identical bits (the in-kernel folds do not depend on which workgroup arrives last); the
    ticket counters are left zeroed."""
    from mono_vifi_amd import ops
    B, H, W = 3, 40, 100
    inps = [_inputs(6100 + u, B, H, W, 0, u == 2) for u in range(3)]
    wts = [0.5, 2.0, -1.25]

    def run():
        flat, leaves = [], []
        for inp in inps:
            d, Tt, f = _flat(inp, dev, 0)
            flat += f
            leaves.append((d, Tt))
        res = ops.Units.apply(_cfg(3, 0), *flat)
        sum(w * res[0][u] for u, w in enumerate(wts)).backward()
        return [float(v) for v in res[0].detach()], [(N(d.grad), N(t.grad)) for d, t in leaves]

    l1, g1 = run()
    l2, g2 = run()
    assert l1 == l2
    for (a, b), (c, d) in zip(g1, g2):
        assert np.array_equal(a, c) and np.array_equal(b, d)
    for u, inp in enumerate(inps):
        ref = O.unit(inp["disp"], inp["tgt"], inp["src"], inp["T"], inp["K"], inp["inv_K"], inp["noise_used"],
                     inp["mask_rec"], 0, want_grads=True, gloss=wts[u])
        assert abs(l1[u] - ref["loss"]) <= 1e-5 * abs(ref["loss"])
        assert_grad_close(g1[u][0], ref["grad_disp"], TOL, f"unit {u} grad_disp (upstream {wts[u]})")
        assert rel_err(g1[u][1], ref["grad_T"]) <= TOL
    for t in ops._TICKETS.values():
        assert int(t.abs().sum()) == 0, "ticket counters not left zeroed"


def test_trainer_losses_batched_equal_unbatched(dev):
    """HotPathLosses.compute_units (one launch, identity maps handed over) against one
    compute_unit per entry: losses and gradients bit-equal for an injected noise tensor."""
    from types import SimpleNamespace
    from mono_vifi_amd.losses import HotPathLosses
    B, H, W = 2, 64, 96
    inps = [_inputs(7000 + u, B, H, W, 0, False) for u in range(3)]

    class L(HotPathLosses):
        pass
    res = {}
    for batched in (True, False):
        s = L()
        s.opt = SimpleNamespace(min_depth=0.1, max_depth=100.0, no_ssim=False, avg_reprojection=False,
                                disable_automasking=False, disparity_smoothness=1e-3, batch_units=batched)
        s.tie_break_noise = T(inps[0]["noise_used"], dev)
        units, leaves = [], []
        for inp in inps:
            d, Tt = T(inp["disp"], dev, True), T(inp["T"], dev, True)
            units.append(dict(disp_tgt={("disp", 0): d}, img_tgt=T(inp["tgt"], dev), poses=Tt,
                              imgs_src=[T(inp["src"][0], dev), T(inp["src"][1], dev)], K=T(inp["K"], dev),
                              inv_K=T(inp["inv_K"], dev)))
            leaves.append((d, Tt))
        losses, idents, _ = s.compute_units(units, want_ident=batched)
        assert (idents is not None) == batched
        losses.sum().backward()
        res[batched] = ([float(v) for v in losses.detach()], [(N(d.grad), N(t.grad)) for d, t in leaves])
    assert res[True][0] == res[False][0]
    for (a, b), (c, d) in zip(res[True][1], res[False][1]):
        assert np.array_equal(a, c) and np.array_equal(b, d)


@pytest.mark.parametrize("which", ["tgt", "src", "mask", "noise", "K"])
def test_mismatched_planes_raise_instead_of_reading_out_of_bounds(dev, which):
    """Raw pointers and strides go to the kernel, so every plane is shape-checked on the host: a
    target / source / mask / noise / intrinsics tensor of another resolution or batch size is a RuntimeError,
    not an out-of-bounds device read."""
    from mono_vifi_amd import ops
    B, H, W = 2, 32, 64
    inp = _inputs(77, B, H, W, 0, True)
    _, _, flat = _flat(inp, dev, 0)
    small = _inputs(78, B, H // 2, W, 0, True)
    if which == "tgt":
        flat[1] = T(small["tgt"], dev)
    elif which == "src":
        flat[9] = T(small["src"][1], dev)
    elif which == "mask":
        flat[5] = T(small["mask_rec"], dev)
    elif which == "noise":
        flat[6] = T(inp["noise_used"][:, :1], dev)          # one candidate's noise for two sources
    else:
        flat[3] = T(inp["K"][:1], dev)                      # another batch size
    with pytest.raises(RuntimeError, match="must be"):
        ops.Units.apply(_cfg(1, 0), *flat)
    # the launch path is intact afterwards (a failed launch drops the cached ticket buffer)
    _, _, good = _flat(inp, dev, 0)
    res = ops.Units.apply(_cfg(1, 0), *good)
    assert torch.isfinite(res[0]).all()


@pytest.mark.parametrize("shape", [(2, 33, 70), (12, 192, 640)])
@pytest.mark.parametrize("n", [1, 3])
def test_group_sum_written_by_the_launch(dev, shape, n):
    """want_sum: the finishing kernel writes the sum of the group's losses (unit order) and the backward pass takes the
    sum's upstream gradient as one more device scalar (no reduce / expand / copy launches around a
    group).  The sum equals the sequential fp32 sum of the per-unit losses bit for bit; gradients through the sum
    equal gradients through `losses.sum()` bit for bit; both paths at once add their upstream gradients; repeated
    launches on one stream leave the (unit + launch) ticket counters clean."""
    from mono_vifi_amd import ops
    B, H, W = shape
    inps = [_inputs(6100 + 17 * u + H, B, H, W, 0, u == 2) for u in range(n)]

    def run(mode):
        flat, leaves = [], []
        for inp in inps:
            d, Tt, f = _flat(inp, dev, 0)
            flat += f
            leaves.append((d, Tt))
        res = ops.Units.apply(_cfg(n, 0, want_sum=(mode != "losses")), *flat)
        if mode == "losses":
            (2.0 * res[0].sum()).backward()
        elif mode == "sum":
            (2.0 * res[-1]).backward()
        else:
            (0.5 * res[0].sum() + 1.5 * res[-1]).backward()
        return res, [(N(d.grad), N(Tt.grad)) for d, Tt in leaves]

    r_l, g_l = run("losses")
    for _ in range(2):                                   # twice: the tickets are left zero
        r_s, g_s = run("sum")
        seq = np.float32(0.0)
        for v in N(r_s[0]):
            seq = np.float32(seq + v)
        assert np.float32(float(r_s[-1])) == seq and r_s[-1].dim() == 0
        assert np.array_equal(N(r_s[0]), N(r_l[0]))
        for (a, b), (c, d) in zip(g_s, g_l):
            assert np.array_equal(a, c) and np.array_equal(b, d)
    _, g_b = run("both")
    for (a, b), (c, d) in zip(g_b, g_l):                 # 0.5 + 1.5 == 2.0: the two upstream gradients add exactly
        assert np.array_equal(a, c) and np.array_equal(b, d)


@pytest.mark.parametrize("shape", [(2, 33, 70), (12, 192, 640)])
def test_six_units_mixed_masks_and_chained_sum(dev, shape):
    """Round 5: the single-frame and the affine units of a step as ONE launch of six (three without and three with
    mask_rec, train.py:747-760 + 837-882), identity maps wanted for the first three only, and the launch's total
    chained into the next launch's finishing kernel (`sum_in`): every unit == the same unit launched alone, bit for
    bit; total == ((0 + l0) + l1 ...) in unit order with the running total in front."""
    from mono_vifi_amd import ops
    B, H, W = shape
    inps = [_inputs(5100 + 17 * u + H, B, H, W, 0, with_mask=(u >= 3)) for u in range(6)]
    flat, leaves = [], []
    for inp in inps:
        d, Tt, f = _flat(inp, dev, 0)
        flat += f
        leaves.append((d, Tt))
    res = ops.Units.apply(_cfg(6, 0, want_ident=[True] * 3 + [False] * 3, want_sum=True), *flat)
    losses, per, total = res[0], res[2:-1], res[-1]
    for u in range(6):
        assert (per[4 * u + 3].numel() > 0) == (u < 3), "identity maps only where they were asked for"
    # second launch: two more units, the first launch's total as its running total
    inps2 = [_inputs(5300 + 19 * u + H, B, H, W, 0, with_mask=False) for u in range(2)]
    flat2, leaves2 = [], []
    for inp in inps2:
        d, Tt, f = _flat(inp, dev, 0)
        flat2 += f
        leaves2.append((d, Tt))
    res2 = ops.Units.apply(_cfg(2, 0, want_sum=True, sum_in=True), *flat2, total)
    total2 = res2[-1]
    (total2 * 1.5).backward()

    seq = np.float32(0.0)
    for u in range(6):
        seq = np.float32(seq + np.float32(float(losses[u].detach())))
    assert np.float32(float(total.detach())) == seq
    for u in range(2):
        seq = np.float32(seq + np.float32(float(res2[0][u].detach())))
    assert np.float32(float(total2.detach())) == seq

    for u, inp in enumerate(inps + inps2):
        d, Tt, f = _flat(inp, dev, 0)
        r1 = ops.Units.apply(_cfg(1, 0), *f)
        (r1[0].sum() * 1.5).backward()
        many = (res if u < 6 else res2)
        uu = u if u < 6 else u - 6
        lv = leaves[u] if u < 6 else leaves2[uu]
        assert float(many[0][uu].detach()) == float(r1[0][0].detach()), f"unit {u}: loss"
        assert np.array_equal(N(many[2 + 4 * uu + 1]), N(r1[2 + 1])), f"unit {u}: argmin"
        assert np.array_equal(N(lv[0].grad), N(d.grad)), f"unit {u}: grad_disp"
        assert np.array_equal(N(lv[1].grad), N(Tt.grad)), f"unit {u}: grad_T"


@pytest.mark.parametrize("shape,G", [((2, 33, 70), 3), ((12, 192, 640), 6)])
@pytest.mark.parametrize("with_depth_grad", [False, True])
def test_deferred_unit_gradients_through_the_disparity_head(dev, shape, G, with_depth_grad):
    """Round 5: units that read a disparity head's output in place leave their RAW gradients to the head's adjoint
    kernel (ops.HeadSink, mvf_disp_head_bwd_units: (raw - shift_b) * g applied on load).  Against the round-4 route
    -- k_fb_scale writes scaled gradients, autograd stacks the groups, mvf_disp_head_bwd reads the stack -- the logit
    gradient and grad_T are identical bit for bit, with and without a second consumer of the head (depth)."""
    from types import SimpleNamespace
    from mono_vifi_amd import ops
    from mono_vifi_amd.losses import HotPathLosses
    B, H, W = shape
    n = min(G, 3)
    inps = [_inputs(6100 + 23 * u + H, B, H, W, 0, with_mask=(u == 1)) for u in range(G)]
    logit0 = torch.empty((B * G, 1, H, W), device=dev)
    for g in range(G):
        dnp = np.clip(inps[g]["disp"], 1e-4, 1 - 1e-4)
        logit0.view(B, G, 1, H, W)[:, g] = T(np.log(dnp / (1 - dnp)).astype(np.float32), dev)
    wdep = torch.rand((B * G, 1, H, W), device=dev)

    class L(HotPathLosses):
        pass

    def run(defer):
        l = L()
        l.opt = SimpleNamespace(min_depth=0.1, max_depth=100.0, no_ssim=False, avg_reprojection=False,
                                disable_automasking=False, disparity_smoothness=1e-3, inkernel_noise=False,
                                batch_units=True, defer_unit_grads=defer)
        logit = logit0.clone().requires_grad_(True)
        disp, depth, part, sink = ops.disp_head(logit, 0.1, 100.0, want_depth=True, want_sink=True)
        assert sink is not None
        views = torch.unbind(disp.view(B, G, 1, H, W), 1)
        parts = torch.unbind(part.view(B, G, 32), 1)
        units, Ts = [], []
        for u in range(n):
            inp = inps[u]
            Tt = T(inp["T"], dev, True)
            Ts.append(Tt)
            l.tie_break_noise = None
            units.append(dict(disp_tgt={("disp", 0): views[u], ("disp_mean_partials", 0): parts[u].contiguous(),
                                        ("disp_head_sink", 0): sink},
                              img_tgt=T(inp["tgt"], dev), poses=Tt, imgs_src=[T(inp["src"][0], dev), T(inp["src"][1], dev)],
                              K=T(inp["K"], dev), inv_K=T(inp["inv_K"], dev),
                              mask_rec=T(inp["mask_rec"], dev) if inp["mask_rec"] is not None else None))
        torch.manual_seed(7)          # the tie-break draw (a tensor here) is the same in both runs
        total, _, _ = l.compute_units(units, want_sum=True)
        loss = total * 0.75
        if with_depth_grad:
            loss = loss + (depth * wdep).mean()
        loss.backward()
        return float(total.detach()), N(logit.grad), [N(t.grad) for t in Ts], len(sink.entries)

    a = run(False)
    b = run(True)
    assert b[3] == 0, "the head's backward consumed the deposited entries"
    assert a[0] == b[0]
    assert np.array_equal(a[1], b[1]), "logit gradient: deferred route != scaled-tensor route"
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    # the groups no unit read got no disparity gradient (only the depth term, if any)
    if not with_depth_grad and G > n:
        assert float(np.abs(b[1].reshape(B, G, -1)[:, n:]).max()) == 0.0


def test_more_deferred_units_than_the_head_kernel_takes_fall_back(dev):
    """a disparity head's adjoint kernel takes the raw gradients of at most MAX_UNITS units; the backward
    used to raise when more had been deposited.  Ten units in two launches read ONE head's output in place: eight are
    deferred, the other two return their scaled gradient through autograd, and the logit gradient equals the
    all-through-autograd route within rounding (the two routes add the same terms in another order)."""
    from types import SimpleNamespace
    from mono_vifi_amd import ops
    from mono_vifi_amd import _native as nat
    from mono_vifi_amd.losses import HotPathLosses
    B, H, W, G = 1, 40, 72, 10
    assert G > nat.MAX_UNITS
    inps = [_inputs(7300 + 31 * u, B, H, W, 0, with_mask=False) for u in range(G)]
    logit0 = torch.empty((B * G, 1, H, W), device=dev)
    for g in range(G):
        dnp = np.clip(inps[g]["disp"], 1e-4, 1 - 1e-4)
        logit0.view(B, G, 1, H, W)[:, g] = T(np.log(dnp / (1 - dnp)).astype(np.float32), dev)

    class L(HotPathLosses):
        pass

    def run(defer):
        l = L()
        l.opt = SimpleNamespace(min_depth=0.1, max_depth=100.0, no_ssim=False, avg_reprojection=False,
                                disable_automasking=False, disparity_smoothness=1e-3, inkernel_noise=False,
                                batch_units=True, defer_unit_grads=defer)
        logit = logit0.clone().requires_grad_(True)
        disp, _, part, sink = ops.disp_head(logit, 0.1, 100.0, want_depth=False, want_sink=True)
        views = torch.unbind(disp.view(B, G, 1, H, W), 1)
        parts = torch.unbind(part.view(B, G, 32), 1)
        total = None
        for lo, hi in ((0, 5), (5, 10)):
            units = []
            for u in range(lo, hi):
                inp = inps[u]
                units.append(dict(disp_tgt={("disp", 0): views[u], ("disp_mean_partials", 0): parts[u].contiguous(),
                                            ("disp_head_sink", 0): sink},
                                  img_tgt=T(inp["tgt"], dev), poses=T(inp["T"], dev, True),
                                  imgs_src=[T(inp["src"][0], dev), T(inp["src"][1], dev)], K=T(inp["K"], dev),
                                  inv_K=T(inp["inv_K"], dev), mask_rec=None))
            torch.manual_seed(11 + lo)
            total, _, _ = l.compute_units(units, want_sum=True, sum_in=total)
        total.backward()
        return float(total.detach()), N(logit.grad), sink.claimed

    a, b = run(False), run(True)
    assert a[2] == 0 and b[2] == nat.MAX_UNITS
    assert a[0] == b[0]
    assert float(np.abs(a[1] - b[1]).max()) <= 1e-6 * float(np.abs(a[1]).max())
    assert float(np.abs(b[1].reshape(B, G, -1)).min(axis=(0, 2)).max()) >= 0.0 and all(
        float(np.abs(b[1].reshape(B, G, -1)[:, g]).max()) > 0.0 for g in range(G))       # every group got its gradient


@pytest.mark.parametrize("shape", [(2, 33, 70), (3, 37, 53), (1, 64, 200), (4, 192, 640), (2, 320, 1024)])
def test_preparing_launch_means_equal_the_heads_partials(dev, shape):
    """The mean disparity a launch computes itself (k_units_prepare: 32 chunk sums per image, sixteen loads in
    flight per lane, folded by the last block to arrive) equals, bit for bit, the one it forms from the partials the
    disparity head supplies (k_disp_head_fwd: the same partition and summation order) -- stats[:, 0:2] = (mean, mean +
    1e-7) of both routes, ragged chunk tails included; losses and raw gradients follow."""
    from mono_vifi_amd import ops
    B, H, W = shape
    inp = _inputs(8800 + H, B, H, W, 0, with_mask=False)
    dnp = np.clip(inp["disp"], 1e-4, 1 - 1e-4)
    logit = T(np.log(dnp / (1 - dnp)).astype(np.float32), dev)
    disp, _, part = ops.disp_head(logit, 0.1, 100.0, want_depth=False, want_sink=False)
    disp = disp.detach()
    res = []
    for parts in (None, [part.contiguous()]):
        d = disp.clone().requires_grad_(True)
        _, Tt, flat = _flat(inp, dev, 0, disp=d)
        torch.manual_seed(3)
        out = ops.Units.apply(_cfg(1, 0, mean_parts=parts), *flat)
        out[0].sum().backward()
        res.append((N(out[0]), N(out[1]), N(d.grad), N(Tt.grad)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]), "loss: own mean != head partials"
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
