import sys, time, torch, json
sys.path.insert(0, '/root/repo')
from freegaussian_amd.harness import main_loss
dev = torch.device('cuda', 0)
res = {}
for (W, H) in ((480, 270), (960, 540), (1920, 1080)):
    pred = torch.rand(H, W, 3, device=dev, requires_grad=True)
    gt = torch.rand(H, W, 3, device=dev)
    for _ in range(10):
        pred.grad = None
        main_loss(pred, gt).backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        pred.grad = None
        main_loss(pred, gt).backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    # device time via events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    res[f"{W}x{H}"] = round(dt, 4)
print(json.dumps(res))
from torch.profiler import profile, ProfilerActivity
pred = torch.rand(1080, 1920, 3, device=dev, requires_grad=True); gt = torch.rand(1080, 1920, 3, device=dev)
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        pred.grad = None
        main_loss(pred, gt).backward()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:12]
print({e.key[:60]: round(e.device_time_total / 5, 1) for e in rows})
print("total device us/step", round(sum(e.device_time_total for e in prof.key_averages()) / 5, 1), "kernels/step", sum(e.count for e in prof.key_averages()) / 5)
