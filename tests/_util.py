"""Helpers for the GPU parity tests."""
import torch

# Tolerances: kernel and oracle get the SAME bf16-rounded
# operands; the kernel accumulates in fp32.
TOL_F32 = 1e-3    # fp32 outputs: the north-star's 1e-3 relative gate (observed ~1e-5)
TOL_BF16 = 3e-3   # bf16 outputs: 1e-3 + one bf16 rounding of the result (~1.6e-3 RMS, <=3.9e-3 max)


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rt(t):
    """bf16 round trip on an fp32 tensor."""
    return t.to(torch.bfloat16).to(torch.float32)


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def assert_close(out, ref, what=""):
    tol = TOL_F32 if out.dtype == torch.float32 else TOL_BF16
    e = rel(out.float(), ref)
    assert torch.isfinite(out.float()).all(), f"{what}: non-finite output"
    assert e <= tol, f"{what}: rel-L2 {e:.3e} > {tol:g} ({out.dtype})"
    return e


def note(name, value):
    """Record a measured distance (XVIT_MEASURE_LOG=path appends "name value"): how the stated gates were calibrated."""
    import os
    path = os.environ.get("XVIT_MEASURE_LOG")
    if path:
        with open(path, "a") as f:
            f.write(f"{name} {value:.4e}\n")
    return value
