"""What torch's own fill_ / copy_ reach on 16 GB of HBM on this box (the practical bandwidth ceiling the normalisation kernels are priced against)."""
import torch, time
x = torch.empty(2_000_000_000, dtype=torch.float64, device="cuda")   # 16 GB
y = torch.empty_like(x)
for name, fn in (("fill_", lambda: x.fill_(2.0)), ("copy_", lambda: y.copy_(x))):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.time() - t) / 5
    print(name, "%.2f ms" % (dt * 1e3), "%.2f TB/s" % ((16e9 * (2 if name == "copy_" else 1)) / dt / 1e12))
