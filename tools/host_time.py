import time, torch, sys
sys.path.insert(0, '.')
from r3m_amd import R3M, augment
from r3m_amd.parallel import SingleDevice
from r3m_amd.trainer import Trainer
torch.set_num_threads(4)
dev = "cuda:0"
m = R3M("cuda", 1e-4, 1024, size=34, l2weight=1e-5, l1weight=1e-5, langweight=0.0, tcnweight=1.0, precision="bf16")
net = SingleDevice(m).to(dev)
B = 512
raw = torch.randint(0, 256, (B, 5, 3, 256, 256), device=dev, dtype=torch.int32).to(torch.uint8)
g = torch.Generator().manual_seed(1)
tr = Trainer(1)
def frames(): return augment.random_resized_crop(raw, per_clip=True, generator=g, fused=True)
for i in range(5): tr.update(net, (frames(), None), i)
torch.cuda.synchronize()
t0 = time.perf_counter(); ts = []
for i in range(10):
    a = time.perf_counter(); f = frames(); b = time.perf_counter(); tr.update(net, (f, None), i); c = time.perf_counter()
    ts.append((b - a, c - b))
torch.cuda.synchronize(); t1 = time.perf_counter()
print("wall per step %.2f ms" % ((t1 - t0) / 10 * 1e3))
print("host: frames() %.2f ms, update() %.2f ms (per step, mean); per-step update ms:" % (sum(x[0] for x in ts) / 10 * 1e3, sum(x[1] for x in ts) / 10 * 1e3), [round(x[1] * 1e3, 1) for x in ts])
