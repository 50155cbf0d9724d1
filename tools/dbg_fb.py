import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from oracle import oracle as O
from mono_vifi_amd import ops, synthetic
dev = torch.device('cuda:0')
T = lambda a, g=False: torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(g)
B, H, W = 2, 24, 40
inp = synthetic.unit_inputs(5, B, H, W, pose_scale=0.03, with_mask=False)
T_np = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1)) for k in range(2)], 0)
ref = O.unit(inp["disp"], inp["tgt"], inp["src"], T_np, inp["K"], inp["inv_K"], inp["noise"], None, 0, want_grads=True)
disp, Tt = T(inp["disp"], True), T(T_np, True)
cfg = (2, 0, 1e-3, 0.1, 100.0, 1e-7, True, False)
loss, am, argmin, _, _ = ops.Unit.apply(disp, T(inp["tgt"]), Tt, T(inp["K"]), T(inp["inv_K"]), None, T(inp["noise"]), cfg, T(inp["src"][0]), T(inp["src"][1]))
loss.backward()
g = disp.grad.cpu().numpy(); r = ref["grad_disp"]
err = np.abs(g - r) / np.abs(r).max()
print("loss", float(loss), ref["loss"], "argmin eq", np.array_equal(argmin.cpu().numpy().astype(np.int32), ref["idx"]))
np.set_printoptions(linewidth=250, precision=1)
print((err[0, 0] > 1e-3).astype(int))
print("gT", Tt.grad.cpu().numpy()[0, 0], ref["grad_T"][0, 0])
