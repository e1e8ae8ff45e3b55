"""HIP streams restricted to a subset of the chip's CUs (hipExtStreamCreateWithCUMask).

Why: the two modality branches of ModelCross are independent between fusion points.  A 256x256-tile GEMM
occupies every CU with one workgroup, so two GEMMs on two ordinary streams mostly serialise — and inside one
GEMM all CUs are in the same phase, so the HBM-bound epilogue never overlaps MFMA work.  Giving each branch
its own half of the CUs makes the two kernel sequences run side by side and out of phase: one branch's output
bursts, LayerNorms and attention overlap the other's MFMA loops.

Measured with xvit_cu_trace (tools/cu_mask_probe.py) on MI355X / ROCm 7.2: a CONTIGUOUS bit range [lo, hi) of
the 256-bit mask selects (hi - lo) / 8 CUs on each of the 8 XCDs (bits 0..127 -> 16 CUs per XCD); sparse
patterns (every other bit, b % 8 < 4, ...) are not honoured and leave the stream unrestricted.  So a split
gives every part the same slice of CUs on all XCDs; the blockIdx -> XCD round-robin is unchanged.
"""
import ctypes as C

import torch

_hip = None


def _libhip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        _hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    return _hip


def masked_stream(device, bits):
    """A torch stream on `device` that may only use the CUs whose mask bit is in `bits` (iterable of ints)."""
    bits = sorted(set(int(b) for b in bits))
    if not bits:
        raise ValueError("empty CU mask")
    nwords = bits[-1] // 32 + 1
    words = (C.c_uint32 * nwords)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    handle = C.c_void_p()
    with torch.cuda.device(device):
        rc = _libhip().hipExtStreamCreateWithCUMask(C.byref(handle), nwords, words)
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed (hipError {rc})")
    return torch.cuda.ExternalStream(handle.value, device=device)


def split_masks(n_cu, n_parts):
    """Contiguous bit ranges: part i may use mask bits [i * n_cu / n_parts, (i + 1) * n_cu / n_parts)."""
    return [range(i * n_cu // n_parts, (i + 1) * n_cu // n_parts) for i in range(n_parts)]
