"""torch.inverse of a [12,3,3] device tensor in the middle of a busy stream: does it synchronise?"""
import time
import torch
dev = torch.device("cuda")
A = torch.eye(3, device=dev).repeat(12, 1, 1) + 0.1 * torch.randn(12, 3, 3, device=dev)
big = torch.randn(8192, 8192, device=dev)
def busy():
    for _ in range(30):
        torch.mm(big, big)
def adj_inv(M):
    a, b, c = M[:, 0], M[:, 1], M[:, 2]
    r0, r1, r2 = torch.linalg.cross(b, c), torch.linalg.cross(c, a), torch.linalg.cross(a, b)
    det = (a * r0).sum(1, keepdim=True)
    return torch.stack([r0, r1, r2], 2) / det.unsqueeze(2)
for name, fn in (("torch.inverse", torch.inverse), ("linalg.inv_ex", lambda M: torch.linalg.inv_ex(M).inverse),
                 ("adjugate", adj_inv), ("cpu round trip", lambda M: torch.inverse(M.cpu()).to(dev))):
    fn(A); torch.cuda.synchronize()
    busy()
    t0 = time.perf_counter()
    X = fn(A)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    err = float((X @ A - torch.eye(3, device=dev)).abs().max())
    print(f"{name:16s} host time behind a busy stream {1e3 * (t1 - t0):8.2f} ms (drain afterwards {1e3 * (t2 - t1):7.2f} ms)  |X A - I| {err:.1e}")
ref = torch.inverse(A)
print("adjugate vs inverse max rel", float(((adj_inv(A) - ref).abs() / ref.abs().clamp_min(1e-3)).max()))
print("inv_ex vs inverse equal", bool(torch.equal(torch.linalg.inv_ex(A).inverse, ref)))
