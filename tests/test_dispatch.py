"""CPU: which kernel family the convolution dispatch picks for every forward / input-gradient launch of the ResNet encoders at the
bench sizes — through r3m_debug_conv_route (csrc/conv.hip gg_route: a pure function of the launch parameters, nothing is launched).
DESIGN.md §4.1 claims that the whole ResNet-50 fp32 step runs the persistent kernel (conv_pw.hip, three forms) or the 3x3 window
kernel; a shape that silently fell back to the per-tile gather kernels would only show up as lost throughput on the GPU.
The layer table is torchvision's (the graph /root/reference/r3m/models/models_r3m.py:41-55 instantiates); the epilogue flags are
the ones csrc/engine.hip requests for each input gradient."""
import ctypes as C

import pytest

WIN, PW_POINT, PW_GATHER, PW_STRIDED, BF16, BF16_HALO, BF16_ROW = 1, 11, 12, 13, 30, 31, 32
STATS, ACCUM, MASKED_ADD, BNRED = 1, 2, 4, 64


@pytest.fixture(scope="module")
def route():
    from r3m_amd import _lib
    L = _lib.lib()

    def f(N, H, Ci, Co, k, s, p, dgrad=0, flags=0, bits=0, dt=0):
        buf = (C.c_int * 8)()
        n = L.r3m_debug_conv_route(N, H, H, Ci, Co, k, s, p, dgrad, flags, bits, dt, buf, 8)
        assert n >= 1, L.r3m_last_error()
        return list(buf[:n])
    return f


def _layers(size):
    """(name, H_in, Ci, Co, k, stride, pad, dgrad_flags, mask_bits) for every convolution behind the stem, dgrad flags as the engine sets
    them: inner BatchNorms get their backward partials from the producing dgrad with the mask recomputed (64); the first conv of a
    block joins the residual gradient and feeds the previous block's last BatchNorm, masked by that block's output bits (4 | 64,
    bits) unless the block has a downsample branch (plain store, then the downsample dgrad accumulates: 2)."""
    bottleneck = size == 50
    blocks = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3]}[size]
    out, H, cin = [], 56, 64
    for li, nb in enumerate(blocks):
        c = 64 << li
        for b in range(nb):
            s = 2 if (b == 0 and li > 0) else 1
            cout = 4 * c if bottleneck else c
            ds = b == 0 and (s != 1 or cin != cout)
            first = (0, 0) if ds else (MASKED_ADD | BNRED, 1)
            name = f"layer{li + 1}.{b}"
            if bottleneck:
                out.append((name + ".conv1", H, cin, c, 1, 1, 0) + first)
                out.append((name + ".conv2", H, c, c, 3, s, 1, BNRED, 0))
                out.append((name + ".conv3", H // s, c, cout, 1, 1, 0, BNRED, 0))
            else:
                out.append((name + ".conv1", H, cin, c, 3, s, 1) + first)
                out.append((name + ".conv2", H // s, c, c, 3, 1, 1, BNRED, 0))
            if ds:
                out.append((name + ".downsample", H, cin, cout, 1, s, 0, ACCUM, 0))
            H //= s
            cin = cout
    return out


@pytest.mark.parametrize("size,frames", [(50, 1280), (34, 2560), (18, 2560)])
def test_fp32_convolutions_run_the_persistent_or_the_window_kernel(route, size, frames):
    fast = {WIN, PW_POINT, PW_GATHER, PW_STRIDED}
    seen = {}
    for (name, H, Ci, Co, k, s, p, dflags, bits) in _layers(size):
        fwd = route(frames, H, Ci, Co, k, s, p, 0, STATS)
        assert len(fwd) == 1 and fwd[0] in fast, f"resnet{size} {name} forward -> {fwd}"
        if name == "layer1.0.conv1" or name == "layer1.0.downsample":
            continue                                   # their input is the stem's output: no input gradient is computed
        dg = route(frames, H, Ci, Co, k, s, p, 1, dflags, bits)
        assert all(r in fast for r in dg), f"resnet{size} {name} dgrad (flags {dflags}, bits {bits}) -> {dg}"
        # a stride-2 dgrad is one launch per output parity class that has taps: four for 3x3, one for 1x1
        assert len(dg) == (1 if s == 1 else (4 if k == 3 else 1)), (name, dg)
        if s == 2:
            assert set(dg) == {PW_STRIDED}, (name, dg)
        for r in fwd + dg:
            seen[r] = seen.get(r, 0) + 1
    assert PW_GATHER in seen and (size != 50 or PW_POINT in seen) and WIN in seen
    print(f"resnet{size}: launches by kernel family {dict(sorted(seen.items()))}")


def test_bf16_launches_take_the_bf16_path(route):
    assert route(1280, 56, 64, 256, 1, 1, 0, 0, STATS, 0, 1) == [BF16]


@pytest.mark.parametrize("size,frames", [(50, 1280), (34, 2560), (18, 2560), (18, 8)])
def test_bf16_3x3_stride1_launches_run_the_kernel_row_kernel(route, size, frames):
    """Round 6 (csrc/conv_row16.hip): every 3x3 / stride-1 forward and input-gradient launch of the bf16 plans — whatever the frame
    count, so that plans of different sizes accumulate in the same order — runs the persistent kernel-row kernel; everything else of
    the bf16 path stays on the gather kernel (30). A shape that fell back to the per-tile halo kernel (31) would only show as lost
    throughput and as a plan-size-dependent rounding."""
    n = 0
    for (name, H, Ci, Co, k, s, p, dflags, bits) in _layers(size):
        fwd = route(frames, H, Ci, Co, k, s, p, 0, STATS, 0, 1)
        want = [BF16_ROW] if (k == 3 and s == 1) else [BF16]
        assert fwd == want, f"resnet{size} {name} forward -> {fwd}"
        if name == "layer1.0.conv1" or name == "layer1.0.downsample":
            continue
        dg = route(frames, H, Ci, Co, k, s, p, 1, dflags, bits, 1)
        if k == 3 and s == 1:
            assert dg == [BF16_ROW], f"resnet{size} {name} dgrad (flags {dflags}) -> {dg}"
            n += 1
        else:
            assert all(r == BF16 for r in dg), (name, dg)
    assert n >= (12 if size == 50 else 7)
    assert route(1280, 56, 128, 128, 3, 2, 1, 1, BNRED, 0, 1) == [BF16] * 4


def test_shapes_outside_the_fast_forms_fall_back(route):
    # channel counts that are not multiples of 64 (the fuzz tests' geometries), mask bits on gathered 128-wide rows
    assert route(2, 13, 32, 96, 3, 2, 1, 0, STATS)[0] not in (WIN, PW_POINT, PW_GATHER, PW_STRIDED)
    assert route(64, 56, 128, 128, 3, 1, 1, 1, MASKED_ADD | BNRED, 1)[0] not in (PW_POINT, PW_GATHER, PW_STRIDED)
