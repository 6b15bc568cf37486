"""ctypes binding of ``libfrcnn_hip.so`` (declared in ``include/frcnn_hip.h``).

PyTorch-ROCm is used for exactly three things here: device allocation, the current HIP stream and
DLPack import.  All arithmetic happens in the hand-written HIP kernels behind the C ABI.  There is
no CPU fallback: if the shared library or a GPU is missing every entry point raises.
"""

from __future__ import annotations

import ctypes
import os
from typing import Any, Dict, Optional, Sequence, Tuple

import torch  # must be imported before the library so both share one HIP runtime (libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRCNN_HIP_LIB overrides the path (diagnostic builds only); the default is the in-tree library.
LIB_PATH = os.environ.get("FRCNN_HIP_LIB") or os.path.join(_HERE, "libfrcnn_hip.so")

FRCNN_OK = 0
FRCNN_EINVAL = -1
FRCNN_EUNSUPPORTED = -2
FRCNN_EHIP = -3

_c_float_p = ctypes.POINTER(ctypes.c_float)

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/frcnn_hip.h one to one.
SIGNATURES: Dict[str, Tuple[Any, Sequence[Any]]] = {
    "frcnn_version": (_i, []),
    "frcnn_last_error": (ctypes.c_char_p, []),
    "frcnn_debug_set_schedule": (_i, [ctypes.c_char_p, _i, _i]),
    "frcnn_generate_anchors": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "frcnn_decode_boxes": (_i, [_vp, _i, _vp, _c_float_p, _i64, _i64, _vp, _vp]),
    "frcnn_encode_boxes": (_i, [_vp, _i, _vp, _c_float_p, _i64, _i64, _vp, _vp]),
    "frcnn_iou_map": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "frcnn_iou_reduce_workspace_bytes": (_sz, [_i, _i, _i]),
    "frcnn_iou_map_reduce": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_top_k": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "frcnn_combined_nms_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "frcnn_combined_nms": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_roi_bbox_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "frcnn_roi_bbox": (_i, [_vp, _vp, _vp, _c_float_p, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_roi_bbox_multi": (_i, [_vp, _vp, _vp, _c_float_p, _i, _i, _i, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "frcnn_crop_and_resize": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "frcnn_roi_pooling_workspace_bytes": (_sz, [_i, _i]),
    "frcnn_roi_pooling": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "frcnn_roi_pooling_prepare": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "frcnn_roi_pooling_prepared": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "frcnn_roi_pooling_prepared_multi": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "frcnn_crop_and_resize_multi": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "frcnn_decoder_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "frcnn_decoder": (_i, [_vp, _vp, _vp, _c_float_p, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_decoder_multi": (_i, [_vp, _vp, _vp, _c_float_p, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "frcnn_gather_copy": (_i, [_vp, _vp, _sz, _vp]),
    "frcnn_gather_copy_multi": (_i, [_vp, _vp, _vp, _i, _vp]),
    "frcnn_host_gather_create": (_i, [_i, ctypes.POINTER(_vp)]),
    "frcnn_host_gather_submit": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "frcnn_host_gather_flush": (_i, [_vp]),
    "frcnn_host_gather_stats": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_longlong)]),
    "frcnn_host_gather_destroy": (_i, [_vp]),
    "frcnn_select_mask": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "frcnn_select_mask_seeded": (_i, [_vp, _vp, _i, ctypes.c_uint64, _i, _i, _vp, _vp]),
    "frcnn_rpn_targets_workspace_bytes": (_sz, [_i, _i, _i]),
    "frcnn_rpn_targets": (_i, [_vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _i, _i, _i, _i, _i, _c_float_p, _vp, _vp,
                               _vp, _sz, _vp]),
    "frcnn_rpn_targets_static": (_i, [_vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _i, _i, _i, _i, _i, _c_float_p, _vp, _vp,
                                      _vp]),
    "frcnn_roi_delta_workspace_bytes": (_sz, [_i, _i, _i]),
    "frcnn_roi_delta": (_i, [_vp, _vp, _vp, _vp, _vp, ctypes.c_uint64, _i, _i, _i, _i, _i, _i, _c_float_p, _vp, _vp,
                             _vp, _sz, _vp]),
}

_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """Load ``libfrcnn_hip.so``; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C tf-faster-rcnn_amd/csrc`). There is no CPU fallback."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header and library disagree
            fn.restype = restype
            fn.argtypes = list(argtypes)
        _lib = handle
    return _lib


def set_schedule(name: str, value: Optional[int]) -> None:
    """Diagnostic: force a schedule knob (``FRCNN_*`` name); ``None`` restores the default."""
    check(lib().frcnn_debug_set_schedule(name.encode(), 0 if value is None else 1, 0 if value is None else int(value)))


class UnsupportedSizeError(RuntimeError):
    """A size beyond what the kernels are built for (FRCNN_EUNSUPPORTED)."""


def check(rc: int) -> None:
    """Turn a C-ABI return code into the exception the reference's TF op would have raised."""
    if rc == FRCNN_OK:
        return
    msg = lib().frcnn_last_error().decode("utf-8", "replace")
    if rc == FRCNN_EINVAL:
        raise ValueError(msg)  # TF: InvalidArgumentError
    if rc == FRCNN_EUNSUPPORTED:
        raise UnsupportedSizeError(msg)
    raise RuntimeError(msg)


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("tf-faster-rcnn_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU fallback.")
    return torch.device("cuda", torch.cuda.current_device())


def device_tensor(x: Any, dtype: torch.dtype, device: Optional[torch.device] = None) -> torch.Tensor:
    """Return ``x`` as a contiguous device tensor of ``dtype`` (DLPack objects are imported zero-copy)."""
    if not isinstance(x, torch.Tensor):
        if hasattr(x, "__dlpack__"):
            x = torch.from_dlpack(x)
        else:
            x = torch.as_tensor(x)
    if device is None:
        device = x.device if x.is_cuda else require_gpu()
    if x.device != device:
        x = x.to(device)
    if x.dtype != dtype:
        x = x.to(dtype)
    return x.contiguous()


class _NoGuard:
    def __enter__(self) -> None:
        return None

    def __exit__(self, *exc: Any) -> bool:
        return False


_NO_GUARD = _NoGuard()


def on_device(device: torch.device):
    """Context in which ``device`` is the current HIP device: the library's launches and ``hipFuncSetAttribute`` calls
    act on the current device, not on the device of the pointers they are given.  Free when it already is current."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr_table(tensors: Sequence[torch.Tensor]):
    """Host array of device pointers (the `*_multi` entry points' pointer tables)."""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def gather_copy_multi(pairs: Sequence[Tuple[int, int, int]], stream: int) -> None:
    """One launch copying every ``(src_ptr, dst_ptr, nbytes)`` of ``pairs`` (at most 8; device -> device / pinned host)."""
    n = len(pairs)
    srcs = (ctypes.c_void_p * n)(*[p[0] for p in pairs])
    dsts = (ctypes.c_void_p * n)(*[p[1] for p in pairs])
    sizes = (ctypes.c_size_t * n)(*[p[2] for p in pairs])
    check(lib().frcnn_gather_copy_multi(srcs, dsts, sizes, n, stream))


def float4(values: Sequence[float]):
    """Host array of 4 floats (the variances), rounded to fp32 like a TF constant."""
    vals = [float(v) for v in values]
    if len(vals) != 4:
        raise ValueError(f"expected 4 variances, got {len(vals)}")
    return (ctypes.c_float * 4)(*vals)


_workspaces: Dict[Tuple[int, int], torch.Tensor] = {}


def workspace(nbytes: int, device: torch.device, stream: Optional[int] = None) -> torch.Tensor:
    """Scratch buffer cached per (device, stream); stream-ordered reuse is race free.  ``stream``: the raw handle of the
    stream the buffer will be used on (default: the current stream)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           stream_ptr(device) if stream is None else stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


# ---------------------------------------------------------------------------------------------------------------------
# Raw HIP events.  torch.cuda.Event cannot set AMD's release-scope flags: a default event performs a system-scope release
# when it is recorded (an L2 write-back and invalidate behind every kernel it follows); events that only order one stream
# of this GPU after another need a device-scope release.
# ---------------------------------------------------------------------------------------------------------------------
hipEventDefault = 0x0
hipEventDisableTiming = 0x2
hipEventDisableSystemFence = 0x20000000
hipEventReleaseToDevice = 0x40000000

_hip: Optional[ctypes.CDLL] = None


def hip() -> ctypes.CDLL:
    """The HIP runtime torch already loaded (same soname, so the same instance)."""
    global _hip
    if _hip is None:
        handle = None
        for name in ("libamdhip64.so.7", "libamdhip64.so"):
            try:
                handle = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if handle is None:
            raise RuntimeError("libamdhip64.so not found; a ROCm runtime is required (there is no CPU fallback)")
        handle.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(_vp), ctypes.c_uint]
        handle.hipEventDestroy.argtypes = [_vp]
        handle.hipEventRecord.argtypes = [_vp, _vp]
        handle.hipStreamWaitEvent.argtypes = [_vp, _vp, ctypes.c_uint]
        handle.hipEventSynchronize.argtypes = [_vp]
        handle.hipEventQuery.argtypes = [_vp]
        handle.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), _vp, _vp]
        handle.hipGetErrorString.restype = ctypes.c_char_p
        handle.hipGetErrorString.argtypes = [ctypes.c_int]
        _hip = handle
    return _hip


def _hip_check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what}: {hip().hipGetErrorString(rc).decode()} ({rc})")


class HipEvent:
    """hipEvent_t with explicit flags; the device current at construction owns it."""

    __slots__ = ("handle", "_hip")

    def __init__(self, flags: int = hipEventDisableTiming | hipEventReleaseToDevice) -> None:
        self._hip = hip()
        h = _vp()
        _hip_check(self._hip.hipEventCreateWithFlags(ctypes.byref(h), flags), "hipEventCreateWithFlags")
        self.handle = h

    def record(self, stream: int) -> None:
        _hip_check(self._hip.hipEventRecord(self.handle, stream), "hipEventRecord")

    def wait(self, stream: int) -> None:
        """Make ``stream`` wait for the most recent record of this event (no host block)."""
        _hip_check(self._hip.hipStreamWaitEvent(stream, self.handle, 0), "hipStreamWaitEvent")

    def synchronize(self) -> None:
        _hip_check(self._hip.hipEventSynchronize(self.handle), "hipEventSynchronize")

    def elapsed_time(self, end: "HipEvent") -> float:
        """Milliseconds between this event and ``end`` (both created without hipEventDisableTiming, both complete)."""
        ms = ctypes.c_float()
        _hip_check(self._hip.hipEventElapsedTime(ctypes.byref(ms), self.handle, end.handle), "hipEventElapsedTime")
        return float(ms.value)

    def __del__(self) -> None:
        try:
            if self.handle:
                self._hip.hipEventDestroy(self.handle)
        except Exception:
            pass

