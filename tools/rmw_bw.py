import torch, time
x = torch.zeros(160*1024*1024, dtype=torch.float32, device="cuda")   # 640 MB
y = torch.empty_like(x)
for name, fn, bytes_ in (("rmw add_", lambda: x.add_(1.0), 2*x.numel()*4), ("copy", lambda: y.copy_(x), 2*x.numel()*4), ("read sum", lambda: x.sum(), x.numel()*4), ("fill", lambda: y.fill_(1.0), x.numel()*4)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/10
    print("%-10s %.3f ms  %.2f TB/s" % (name, ms, bytes_/ms/1e9))
