import time, torch
x = torch.zeros(1024, device="cuda")
torch.cuda.synchronize()
ts = []
for _ in range(200):
    t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("idle synchronize: median %.1f us" % (sorted(ts)[100] * 1e6))
ev = torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); ev.record(); ts.append(time.perf_counter() - t0)
print("event record: median %.1f us" % (sorted(ts)[100] * 1e6))
ts = []
for _ in range(200):
    x.add_(1); t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("tiny kernel + synchronize: median %.1f us" % (sorted(ts)[100] * 1e6))
ts = []
for _ in range(200):
    ev2 = torch.cuda.Event(); x.add_(1); ev2.record(); t0 = time.perf_counter()
    while not ev2.query(): pass
    ts.append(time.perf_counter() - t0)
print("tiny kernel + event poll: median %.1f us" % (sorted(ts)[100] * 1e6))
