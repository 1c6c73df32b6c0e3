import torch, time
x = torch.empty(400*1024*1024//8, dtype=torch.float64, device="cuda")
y = torch.randn_like(x)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
mb = x.numel()*8/1e6
for name, fn, traffic in [("fill", lambda: x.fill_(1.0), mb), ("zero_", lambda: x.zero_(), mb), ("copy", lambda: x.copy_(y), 2*mb), ("sum(read)", lambda: y.sum(), mb), ("mul_ inplace (r+w)", lambda: x.mul_(1.0001), 2*mb)]:
    us = t(fn)
    print("%-20s %8.1f us  %7.2f TB/s" % (name, us, traffic/us/1e6*1e0))
