"""Small constant device tensors, created once per (values, device, dtype).

``torch.tensor([...], device="cuda")`` inside a step is a pageable host-to-device copy per
call: a hidden synchronisation point in eager mode and not permitted while a HIP graph is being
captured (``Trainer --hip_graph``).  The first (eager warm-up) use creates the tensor; later
uses -- including the capture -- find it here.  Callers must not modify the result in place."""
import torch

_CACHE = {}


def const_tensor(values, device, dtype=torch.float32):
    key = (tuple(float(v) for v in values), str(device), dtype)
    t = _CACHE.get(key)
    if t is None:
        t = torch.tensor([float(v) for v in values], dtype=dtype, device=device)
        _CACHE[key] = t
    return t
