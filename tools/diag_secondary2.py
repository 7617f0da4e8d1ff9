"""Diagnostic: bench.py's workload sequence, then configs[4] measured several times in the same process under variations."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = {"rank": 0, "world": 1, "dev": dev, "use_dist": False}
mode = sys.argv[1]
if mode == "threads1":
    torch.set_num_threads(1)
w1 = dict(size=50, clips=256, precision="fp32", langweight=0.0, doaug="none", unfused_crop=False, encoder_only_frames=0)
r = bench.measure(w1, 10, 3, 2.0, ctx); print(mode, "c1", r["ms_per_step"], flush=True)
sw3 = dict(bench.SECONDARY[1][1], unfused_crop=False, encoder_only_frames=0)
r = bench.measure(sw3, 10, 5, 2.0, ctx); print("c3", r["ms_per_step"], flush=True)
sw = dict(bench.SECONDARY[2][1], unfused_crop=False, encoder_only_frames=0)
if mode == "nocrop":
    sw["doaug"] = "none"
for i in range(5):
    if mode == "nogc":
        gc.disable()
    r = bench.measure(sw, 10, 5, 2.0, ctx, kernel_timing=(mode != "notiming"))
    print(mode, "configs[4]", r["ms_per_step"], flush=True)
