#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -n 3 -k "stem" > gpurun_out/stem_ops.log 2>&1; tail -3 gpurun_out/stem_ops.log
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from r3m_amd import _lib
L = _lib.lib()
Fr = 1280
x = torch.rand((Fr, 224, 224, 3), device='cuda')
w = torch.randn((64, 7, 7, 3), device='cuda') * 0.05
y = torch.empty((Fr, 112, 112, 64), device='cuda')
stats = torch.empty((Fr * 49, 2, 64), device='cuda')
dy = torch.randn_like(y)
dw = torch.empty_like(w)
wsb = L.r3m_stem_conv_wgrad_workspace_bytes()
ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
fl = 2.0 * Fr * 12544 * 64 * 147
t = timeit(lambda: L.r3m_stem_conv_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), stats.data_ptr(), Fr, st))
print(f"stem fwd   {t:.3f} ms  {fl/t/1e9:.1f} TF")
t = timeit(lambda: L.r3m_stem_conv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, Fr, 0, st))
print(f"stem wgrad {t:.3f} ms  {fl/t/1e9:.1f} TF")
PY
