import torch, time
x = torch.empty(2 * 1024**3, dtype=torch.float32, device="cuda")  # 8 GiB
y = torch.empty_like(x)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
dt = t(lambda: x.fill_(1.0)); print("fill  8 GiB: %.2f ms  %.2f TB/s write" % (dt*1e3, x.numel()*4/dt/1e12))
dt = t(lambda: y.copy_(x));   print("copy  8 GiB: %.2f ms  %.2f TB/s read + same write" % (dt*1e3, x.numel()*4/dt/1e12))
dt = t(lambda: x.sum());      print("sum   8 GiB: %.2f ms  %.2f TB/s read" % (dt*1e3, x.numel()*4/dt/1e12))
