#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 300 -p no:cacheprovider -k "stem_on_bf16" > gpurun_out/stem16.log 2>&1; tail -25 gpurun_out/stem16.log
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from r3m_amd import _lib
L = _lib.lib()
Fr = 1280
x = torch.rand((Fr, 3, 224, 224), device='cuda') * 255
w = torch.randn((64, 7, 7, 3), device='cuda') * 0.05
xn16 = torch.empty(L.r3m_stem_xn16_bytes(Fr) // 2, dtype=torch.bfloat16, device='cuda')
y = torch.empty((Fr, 112, 112, 64), dtype=torch.bfloat16, device='cuda')
stats = torch.empty((Fr * 49, 2, 64), device='cuda')
dy = torch.randn((Fr, 112, 112, 64), device='cuda').bfloat16()
dw = torch.empty_like(w)
wsb = L.r3m_stem_conv_wgrad_bf16_workspace_bytes()
ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=5):
    for _ in range(2): assert fn() == 0, L.r3m_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("prep16  %.3f ms" % timeit(lambda: L.r3m_stem_prep_bf16(x.data_ptr(), xn16.data_ptr(), Fr, st)))
print("fwd16   %.3f ms" % timeit(lambda: L.r3m_stem_conv_fwd_bf16(xn16.data_ptr(), w.data_ptr(), y.data_ptr(), stats.data_ptr(), Fr, st)))
print("wgrad16 %.3f ms" % timeit(lambda: L.r3m_stem_conv_wgrad_bf16(xn16.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, Fr, 0, st)))
PY
