"""ctypes binding of libr3m_hip.so (C ABI: include/r3m_hip.h).

The HIP library is the product: there is NO fallback. `lib()` raises if the shared object is missing, and every
wrapper raises RuntimeError(r3m_last_error()) on a non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("R3M_HIP_LIB") or os.path.join(_HERE, "lib", "libr3m_hip.so")   # env: A/B builds of the same ABI

_lib = None

c_f = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())
c_i = C.c_int
c_ll = C.c_longlong
c_sz = C.c_size_t
c_fl = C.c_float
c_d = C.c_double

# name -> (restype, argtypes). Mirrors include/r3m_hip.h one to one (tests/test_abi.py checks the header against it).
SIGNATURES = {
    "r3m_abi_version": (c_i, []),
    "r3m_last_error": (C.c_char_p, []),
    "r3m_debug_occupancy": (c_i, [C.POINTER(c_i)]),
    "r3m_debug_occupy": (c_i, [c_i, c_i, C.c_double, C.c_void_p]),
    "r3m_debug_set_dynamic_tiles": (c_i, [c_i]),
    "r3m_debug_set_pw16": (c_i, [c_i]),
    "r3m_debug_set_conv3x3_bf16": (c_i, [c_i]),
    "r3m_debug_set_fused_inference": (c_i, [c_i]),
    "r3m_debug_conv_route": (c_i, [c_i] * 12 + [C.POINTER(c_i), c_i]),
    "r3m_profile_enable": (None, [c_i]),
    "r3m_profile_classes": (C.c_uint, [C.c_uint]),
    "r3m_profile_collect": (c_i, [C.POINTER(c_d), C.POINTER(c_ll), C.POINTER(c_d)]),
    "r3m_profile_collect_bytes": (c_i, [C.POINTER(c_d)]),
    "r3m_profile_dump_to": (c_i, [C.c_char_p]),
    "r3m_resnet_create": (C.c_void_p, [c_i, c_i]),
    "r3m_resnet_destroy": (None, [C.c_void_p]),
    "r3m_resnet_out_dim": (c_i, [C.c_void_p]),
    "r3m_resnet_num_params": (c_ll, [C.c_void_p]),
    "r3m_resnet_num_buffers": (c_ll, [C.c_void_p]),
    "r3m_resnet_arena_bytes": (c_ll, [C.c_void_p]),
    "r3m_resnet_num_tensors": (c_i, [C.c_void_p]),
    "r3m_resnet_tensor_info": (c_i, [C.c_void_p, c_i, C.c_char_p, c_i, C.POINTER(c_i), C.POINTER(c_ll), C.POINTER(c_i),
                                     C.POINTER(c_i)]),
    "r3m_resnet_stage_range": (c_i, [C.c_void_p, c_i, C.POINTER(c_ll), C.POINTER(c_ll)]),
    "r3m_resnet_forward": (c_i, [C.c_void_p, c_f, c_f, c_f, c_f, c_f, c_i, c_f]),
    "r3m_resnet_forward_crop": (c_i, [C.c_void_p, c_f, c_i, c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_f]),
    "r3m_resnet_set_fused_bn_reduce": (c_i, [C.c_void_p, c_i]),
    "r3m_resnet_set_bn_pair": (c_i, [C.c_void_p, c_i]),
    "r3m_resnet_backward": (c_i, [C.c_void_p, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    "r3m_conv2d_stats_rows": (c_i, [c_i] * 7),
    "r3m_conv2d_fwd": (c_i, [c_f, c_f, c_f, c_f] + [c_i] * 8 + [c_f]),
    "r3m_conv2d_dgrad_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "r3m_conv2d_dgrad": (c_i, [c_f, c_f, c_f, c_f, c_sz] + [c_i] * 8 + [c_f]),
    "r3m_conv2d_wgrad_workspace_bytes": (c_sz, [c_i] * 8),
    "r3m_conv2d_wgrad": (c_i, [c_f, c_f, c_f, c_f, c_sz] + [c_i] * 9 + [c_f]),
    "r3m_stem_prep": (c_i, [c_f, c_f, c_i, c_f]),
    "r3m_stem_conv_fwd": (c_i, [c_f, c_f, c_f, c_f, c_i, c_f]),
    "r3m_stem_conv_wgrad_workspace_bytes": (c_sz, []),
    "r3m_stem_conv_wgrad": (c_i, [c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_f]),
    "r3m_bn_workspace_bytes": (c_sz, [c_ll, c_i]),
    "r3m_bn_train_coeffs": (c_i, [c_f, c_i, c_ll, c_f, c_f, c_f, c_f, c_fl, c_fl, c_f, c_f, c_sz, c_i, c_f]),
    "r3m_bn_eval_coeffs": (c_i, [c_f, c_f, c_f, c_f, c_fl, c_f, c_i, c_f]),
    "r3m_bn_act_fwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_ll, c_i, c_i, c_f, c_f]),
    "r3m_bn_bwd": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_ll, c_i, c_i, c_i, c_f]),
    "r3m_maxpool_fwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "r3m_maxpool_bwd": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "r3m_avgpool_fwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f]),
    "r3m_avgpool_bwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f]),
    "r3m_resnet_create_dt": (C.c_void_p, [c_i, c_i, c_i]),
    "r3m_resnet_dtype": (c_i, [C.c_void_p]),
    "r3m_convert_bf16": (c_i, [c_f, c_f, c_ll, c_f]),
    "r3m_conv2d_fwd_dt": (c_i, [c_f, c_f, c_f, c_f] + [c_i] * 9 + [c_f]),
    "r3m_conv2d_dgrad_dt": (c_i, [c_f, c_f, c_f, c_f, c_sz] + [c_i] * 9 + [c_f]),
    "r3m_conv2d_dgrad_bnred_rows": (c_i, [c_i] * 4),
    "r3m_conv2d_dgrad_bnred_dt": (c_i, [c_f, c_f, c_f, c_f, c_sz] + [c_i] * 8 + [c_f] * 8 + [c_i, c_f]),
    "r3m_conv2d_wgrad_workspace_bytes_dt": (c_sz, [c_i] * 9),
    "r3m_conv2d_wgrad_dt": (c_i, [c_f, c_f, c_f, c_f, c_sz] + [c_i] * 10 + [c_f]),
    "r3m_stem_conv_fwd_dt": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_f]),
    "r3m_stem_conv_wgrad_dt": (c_i, [c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_i, c_f]),
    "r3m_stem_xn16_bytes": (c_sz, [c_i]),
    "r3m_stem_prep_bf16": (c_i, [c_f, c_f, c_i, c_f]),
    "r3m_stem_prep_crop": (c_i, [c_f, c_i, c_f, c_i, c_i, c_i, c_f, c_i, c_i, c_f]),
    "r3m_stem_conv_fwd_bf16": (c_i, [c_f, c_f, c_f, c_f, c_i, c_f]),
    "r3m_stem_conv_wgrad_bf16_workspace_bytes": (c_sz, []),
    "r3m_stem_conv_wgrad_bf16": (c_i, [c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_f]),
    "r3m_bn_act_fwd_dt": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_ll, c_i, c_i, c_f, c_i, c_f]),
    "r3m_bn_bwd_dt": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_ll, c_i, c_i, c_i, c_i, c_f]),
    "r3m_bn_relu_maxpool_fwd_dt": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "r3m_bn_maxpool_bwd_dt": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "r3m_maxpool_fwd_dt": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "r3m_maxpool_bwd_dt": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "r3m_avgpool_fwd_dt": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "r3m_avgpool_bwd_dt": (c_i, [c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "r3m_linear_fwd": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "r3m_crop_resize": (c_i, [c_f, c_i, c_f, c_f, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "r3m_resize_crop": (c_i, [c_f, c_i, c_f, c_ll, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "r3m_langrew_num_params": (c_ll, [c_i, c_i, c_i]),
    "r3m_langrew_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "r3m_langrew_forward": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_i, c_i, c_f]),
    "r3m_langrew_backward": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_i, c_i, c_i, c_f]),
    "r3m_langrew_forward_dt": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_i, c_i, c_i, c_f]),
    "r3m_langrew_backward_dt": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_f]),
    "r3m_langrew_call_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "r3m_langrew_call_forward": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_i, c_i, c_f]),
    "r3m_langrew_call_backward": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_i, c_i, c_i, c_f]),
    "r3m_loss_workspace_bytes": (c_sz, [c_i]),
    "r3m_loss_tcn_lp": (c_i, [c_f, c_f, c_f, c_f, c_f, c_sz, c_i, c_i, c_i, c_fl, c_fl, c_fl, c_f]),
    "r3m_loss_lang_infonce": (c_i, [c_f, c_f, c_f, c_f, c_sz, c_i, c_fl, c_f]),
    "r3m_loss_finalize": (c_i, [c_f, c_sz, c_i, c_i, c_f, c_fl, c_fl, c_fl, c_fl, c_f]),
    "r3m_adam_step": (c_i, [c_f, c_f, c_f, c_f, c_ll, c_d, c_d, c_d, c_d, c_ll, c_fl, c_f]),
    "r3m_sgd_step": (c_i, [c_f, c_f, c_f, c_ll, c_d, c_d, c_d, c_d, c_i, c_ll, c_fl, c_f]),
}


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle. Raises HipLibraryMissing if the .so has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"r3m_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or r3m_amd/csrc/build.sh (hipcc --offload-arch=gfx950). There is no CPU / eager fallback.")
    # torch (when already imported) has loaded its own libamdhip64.so.7; same SONAME -> one HIP runtime per process.
    h = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    ab_build = bool(os.environ.get("R3M_HIP_LIB"))         # an A/B library of an older tree (tools/build_ab.sh) may predate diagnostics
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(h, name)  # AttributeError here means header/binding/library drifted apart
        except AttributeError:
            if ab_build and name.startswith("r3m_debug_"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = h
    return h


def last_error():
    msg = lib().r3m_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"r3m_hip {what} failed (code {rc}): {last_error()}")


_pinned = {}   # (shape, dtype) -> [slots, next]; a slot is [pinned buffer, event of the last copy that read it]
_PIN_RING = 4


def upload_small(t, device, dtype=None):
    """Host tensor -> `device` without stalling the host: a copy from PAGEABLE memory is stream-ordered AND blocks the calling
    thread on ROCm, i.e. the host waits for everything queued before it (the previous step's backward when it sits at the top of a
    step) and cannot queue ahead. Small per-step tensors (crop boxes, permutations) go through a ring of pinned staging buffers
    per (shape, dtype) instead. Every slot remembers the event recorded behind the copy that last read it; a slot whose copy has
    not run yet is NOT overwritten — a caller that is more than four uploads ahead of the GPU (an encoder-only loop that never
    synchronises, a deep prefetcher) gets a fresh pinned buffer, the ring grows to what that caller needs and stays bounded by
    how far the host can run ahead."""
    import torch
    if t.is_cuda or torch.device(device).type != "cuda":
        return t.to(device=device, dtype=dtype or t.dtype)
    src = t.to(dtype or t.dtype).contiguous()
    key = (tuple(src.shape), src.dtype)
    ring = _pinned.setdefault(key, [[], 0])
    slots = ring[0]
    if len(slots) < _PIN_RING:
        slot = [torch.empty(src.shape, dtype=src.dtype, pin_memory=True), None]
        slots.append(slot)
    else:
        i = ring[1] % len(slots)                        # the oldest slot
        slot = slots[i]
        if slot[1] is not None and not slot[1].query():  # its copy has not run yet: leave its host memory alone
            slot = [torch.empty(src.shape, dtype=src.dtype, pin_memory=True), None]
            slots.insert(i, slot)                       # the cursor moves on to the still-busy oldest slot
        ring[1] = i + 1
    slot[0].copy_(src)
    with torch.cuda.device(device):
        out = slot[0].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()                                     # on the stream the copy was queued on (the current one)
    slot[1] = ev
    return out


def stream_ptr(device=None):
    """HIP stream handle torch is currently using on `device` (a torch.device / index; None = the current device)."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def on(t):
    """Context manager: make the device that owns tensor `t` current for the enclosed C-ABI calls. The library launches on
    the stream it is handed and never calls hipSetDevice itself (include/r3m_hip.h: "device selected by the caller"), so
    a module living on cuda:1 while cuda:0 is current must switch here — kernels, hipFuncSetAttribute and the profiling
    events all bind to the current device."""
    import torch
    return torch.cuda.device(t.device)


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()
