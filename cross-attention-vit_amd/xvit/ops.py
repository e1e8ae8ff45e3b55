"""Thin tensor-level wrappers over the C ABI: pointer/stride extraction, current-stream
plumbing, loud failure.  No autograd here (see functional.py) and no fallback: a tensor that
is not on the GPU is an error."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import GemmArgs

BF16, F32 = 0, 1
NT, NN, TN = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_DGELU = 0, 1, 2


# Host cost matters at the reference's batch (8 volume pairs: ~500 launches per step against 4.4 ms of GPU work): the current device
# and its current stream are read through torch's C entry points (what torch.cuda.current_device / current_stream wrap, without the
# lazy-init checks and the Stream object per call: 9 us -> 0.3 us).
_cur_dev = torch._C._cuda_getDevice
_raw_stream = torch._C._cuda_getCurrentRawStream


def _stream() -> int:
    return _raw_stream(_cur_dev())


# Optional live per-kernel timing (bench.py): when PROFILE is a list, every wrapper brackets its
# launch with HIP events recorded on the launch stream and appends
# (kernel family, algorithmic work, "flop" | "byte", start_event, end_event).
PROFILE = None
# Deterministic gradients (SURVEY.md 8(b) "Determinism"): XVIT_DETERMINISTIC=1 or set_deterministic(True).  The kernels that
# normally meet in fp32 atomics (LayerNorm dgamma / dbeta and the bias column sums next to it, xvit_colsum, the class-head
# wgrad) then store per-block partial sums and add them in a fixed order, and the GEMM epilogue's fused column sums are
# replaced by a separate xvit_colsum: every .grad is bit-identical from run to run (attention backward and split-K already are).
DETERMINISTIC = os.environ.get("XVIT_DETERMINISTIC", "0") == "1"


def set_deterministic(on: bool = True):
    global DETERMINISTIC
    DETERMINISTIC = bool(on)


GEMM_TILE = 0            # mirror of xvit_set_option("gemm_tile") for the family names below


def set_option(name: str, value: int):
    """xvit_set_option (include/xvit.h): process-wide tuning knobs, e.g. ("gemm_tile", 1) forces the 128x128 kernel."""
    global GEMM_TILE
    _lib.check(_lib.load().xvit_set_option(name.encode(), int(value)), "xvit_set_option")
    if name == "gemm_tile":
        GEMM_TILE = int(value)


_DROP_EPOCH = None       # keeps the registered counter tensor alive


def set_dropout_epoch(counter):
    """xvit_set_dropout_epoch (include/xvit.h): register a one-element int64 device tensor whose value is mixed into every dropout seed at
    kernel run time (captured steps: xvit.graph.GraphedStep increments it at the head of its graph), or None to switch it off."""
    global _DROP_EPOCH
    if counter is not None and not (counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1):
        raise RuntimeError("set_dropout_epoch: need a one-element int64 GPU tensor (or None)")
    _lib.check(_lib.load().xvit_set_dropout_epoch(counter.data_ptr() if counter is not None else None), "xvit_set_dropout_epoch")
    _DROP_EPOCH = counter


PROFILE_SHAPES = False   # append the GEMM shape/epilogue to the family name (bench.py --detail)


def _run(name, work, kind, rc_fn, what):
    if PROFILE is None:
        _lib.check(rc_fn(), what)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    _lib.check(rc_fn(), what)
    e.record()
    PROFILE.append((name, float(work), kind, s, e))


def _ptr(t) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("xvit: tensor is not on the GPU; the HIP path has no CPU fallback")
    if t.device.index != _cur_dev():
        # launches go to the CURRENT device's current stream (_stream()): a tensor of another GPU would be written by a
        # kernel on the wrong device, unordered with torch's own work on the tensor's device
        raise RuntimeError(f"xvit: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device(...) (one process per GPU) or wrap the call in torch.cuda.device(...)")
    return t.data_ptr()


def _dt(t) -> int:
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"xvit: unsupported dtype {t.dtype}")


def _rows2d(t):
    """Leading dimension of a 2-D row-major view."""
    assert t.dim() == 2 and t.stride(1) == 1, f"need a row-major 2-D tensor, got shape {tuple(t.shape)} strides {t.stride()}"
    return t.stride(0)


def gemm(layout, A, B, C_out, *, bias=None, residual=None, aux=None, act=ACT_NONE, accumulate=False,
         split_k=1, res_row_mod=0, res_row_off=0, out_seg=(0, 0, 0), M=None, N=None, K=None, colsum=None, dropout=None, aux_mode=0):
    """C = op(A) op(B) with the fused epilogue of include/xvit.h.  2-D tensors, or 3-D
    [batch, rows, cols] for a strided batch (all of A, B, C and optional bias 2-D / residual /
    aux 3-D then carry the batch in dim 0)."""
    a = GemmArgs()
    batched = A.dim() == 3
    A2, B2, C2 = (A[0], B[0], C_out[0]) if batched else (A, B, C_out)
    if layout == NT:
        m, k = A2.shape; n = B2.shape[0]; assert B2.shape[1] == k
    elif layout == NN:
        m, k = A2.shape; n = B2.shape[1]; assert B2.shape[0] == k
    else:
        k, m = A2.shape; n = B2.shape[1]; assert B2.shape[0] == k
    a.layout, a.M, a.N, a.K = layout, M or m, N or n, K or k
    a.batch = A.shape[0] if batched else 1
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16
    a.c_dtype, a.act, a.accumulate, a.split_k = _dt(C_out), act, int(bool(accumulate)), split_k
    a.res_row_mod, a.res_row_off = res_row_mod, res_row_off
    a.aux_mode = aux_mode             # 1: aux carries gelu'(z) instead of z (include/xvit.h)
    a.out_seg_rows, a.out_seg_skip, a.out_row_off = out_seg
    a.A, a.B, a.C = _ptr(A), _ptr(B), _ptr(C_out)
    a.lda, a.ldb, a.ldc = _rows2d(A2), _rows2d(B2), _rows2d(C2)
    if batched:
        a.stride_a, a.stride_b, a.stride_c = A.stride(0), B.stride(0), C_out.stride(0)
    if bias is not None:
        assert bias.dtype == torch.float32
        a.bias = _ptr(bias)
        if batched:
            a.stride_bias = bias.stride(0)
    if residual is not None:
        assert residual.dtype == torch.float32
        r2 = residual[0] if batched else residual
        a.residual, a.ldr = _ptr(residual), _rows2d(r2)
        if batched:
            a.stride_r = residual.stride(0)
    if aux is not None:
        assert aux.dtype == torch.bfloat16
        x2 = aux[0] if batched else aux
        a.aux, a.ldaux = _ptr(aux), _rows2d(x2)
        if batched:
            a.stride_aux = aux.stride(0)
    if colsum is not None:
        assert colsum.dtype == torch.float32
        a.colsum = _ptr(colsum)
    if dropout is not None and dropout[0] > 0.0:   # (p, seed)
        a.dropout_p, a.dropout_seed = float(dropout[0]), int(dropout[1])
    ws = None
    if split_k > 1:
        need = _lib.load().xvit_gemm_workspace_bytes(C.byref(a))
        if need:
            ws = torch.empty(need // 4, dtype=torch.float32, device=C_out.device)
            a.workspace, a.workspace_bytes = _ptr(ws), need
    # one family per kernel symbol: gemm_big_kernel<..> (M, N >= 256) vs gemm_kernel<..>; "+splitk" brackets also
    # contain the splitk_epilogue_kernel launch that follows
    big = GEMM_TILE != 1 and a.M >= 256 and a.N >= 256 and ((a.M + 255) // 256) * ((a.N + 255) // 256) * a.batch * max(split_k, 1) > 128   # = use_big_tile() in gemm.hip
    kind = "big" if big else "small"
    tag = f"gemm_{kind}_" + ("nt", "nn", "tn")[layout] + ("+splitk" if split_k > 1 else "")
    if PROFILE is not None and PROFILE_SHAPES:
        epi = ("+b" if bias is not None else "") + ("+gelu" if act == ACT_GELU else "+dgelu" if act == ACT_DGELU else "") + ("+res" if residual is not None else "")
        tag += f"[{a.M}x{a.N}x{a.K}{epi}{'' if a.c_dtype == BF16 else ',f32'}{',s%d' % split_k if split_k > 1 else ''}]"
    _run(tag, 2.0 * a.M * a.N * a.K * a.batch, "flop",
         lambda: _lib.load().xvit_gemm(C.byref(a), _stream()), "xvit_gemm")
    return C_out


def _alt_ld(x, x_alt, seq_len):
    """Row stride between the x_alt rows of consecutive sequences: x_alt is either a tensor laid out like x (its rows
    k * seq_len are used) or a packed [sequences, d] copy of those rows."""
    if x_alt is None:
        return 0
    rows, d = x.shape
    if x_alt.shape == x.shape and x_alt.stride() == x.stride():
        return seq_len * _rows2d(x)
    assert seq_len > 0 and x_alt.shape == (rows // seq_len, d) and x_alt.stride(1) == 1 and x_alt.dtype == torch.float32
    return x_alt.stride(0)


def layernorm_fwd(x, gamma, beta, eps, *, x_alt=None, seq_len=0, out=None):
    """x fp32 [rows, d] (row stride free) -> (y bf16 [rows, d], mean, rstd)."""
    rows, d = x.shape
    y = out if out is not None else torch.empty(rows, d, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    ld_alt = _alt_ld(x, x_alt, seq_len)
    _run("layernorm_fwd", rows * d * 6.0, "byte",
         lambda: _lib.load().xvit_layernorm_fwd(_ptr(x), _ptr(x_alt), _rows2d(x), seq_len, ld_alt, _ptr(gamma), _ptr(beta), eps,
                                                _ptr(y), _rows2d(y), None, 0, _ptr(mean), _ptr(rstd), rows, d, _stream()), "xvit_layernorm_fwd")
    return y, mean, rstd


def layernorm_fwd_f32(x, gamma, beta, eps, want_bf16=True):
    """x fp32 [rows, d] (row stride free) -> (y fp32, y bf16 | None, mean, rstd): the single-token CLS path keeps fp32 operands."""
    rows, d = x.shape
    yf = torch.empty(rows, d, dtype=torch.float32, device=x.device)
    yb = torch.empty(rows, d, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().xvit_layernorm_fwd(_ptr(x), None, _rows2d(x), 0, 0, _ptr(gamma), _ptr(beta), eps, _ptr(yb), d, _ptr(yf), d,
                                              _ptr(mean), _ptr(rstd), rows, d, _stream()), "xvit_layernorm_fwd")
    return yf, yb, mean, rstd


def linear_f32(x, W, bias=None, *, act=ACT_NONE, residual=None, want_z=False, want_bf16=False, dropout=None):
    """fp32 Linear of the single-token CLS path (xvit_linear_f32): x fp32 [M, K] (row stride free), W fp32 [N, K] (the master
    weight itself) -> (y fp32 [M, N], y bf16 | None, z bf16 | None); z = the GELU pre-activation (act = ACT_GELU)."""
    M, K = x.shape
    N = W.shape[0]
    assert x.dtype == torch.float32 and W.dtype == torch.float32 and W.shape[1] == K
    if K % 16 or x.data_ptr() % 16 or W.data_ptr() % 16 or x.stride(0) % 4 or W.stride(0) % 4:
        # the kernel walks K in 16-byte steps of aligned rows: a width that is not a multiple of 16 (the reference accepts any
        # hidden_dim / mlp_dim) or an oddly offset view is zero-padded into aligned scratch copies first (exact: the pad adds 0)
        K16 = (K + 15) // 16 * 16
        xp = torch.zeros(M, K16, dtype=torch.float32, device=x.device)
        wp = torch.zeros(N, K16, dtype=torch.float32, device=x.device)
        xp[:, :K].copy_(x)
        wp[:, :K].copy_(W)
        x, W, K = xp, wp, K16
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    yb = torch.empty(M, N, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    zb = torch.empty(M, N, dtype=torch.bfloat16, device=x.device) if want_z else None
    lib = _lib.load()
    need = lib.xvit_linear_f32_workspace_bytes(M, N, K)
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device) if need else None
    dp, seed = (float(dropout[0]), int(dropout[1])) if dropout is not None and dropout[0] > 0.0 else (0.0, 0)
    _run("linear_f32", 2.0 * M * N * K, "flop",
         lambda: lib.xvit_linear_f32(_ptr(x), _rows2d(x), _ptr(W), _rows2d(W), _ptr(bias), _ptr(y), N, M, N, K, act, _ptr(zb), N,
                                     _ptr(residual), _rows2d(residual) if residual is not None else 0, _ptr(yb), N, dp, seed, _ptr(ws), need, _stream()),
         "xvit_linear_f32")
    return y, yb, zb


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, *, x_alt=None, seq_len=0, dres=None, want_bf16=False, dxsum=None, dressum=None):
    """-> (dx fp32, dx_bf16 | None); dgamma/dbeta (fp32, [d]) are accumulated into; so are the optional
    column sums dxsum (of dx) and dressum (of dres)."""
    rows, d = x.shape
    dx = torch.empty(rows, d, dtype=torch.float32, device=x.device)
    dxb = torch.empty(rows, d, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    nbytes = rows * d * (2.0 + 4.0 + 4.0 + (4.0 if dres is not None else 0.0) + (2.0 if want_bf16 else 0.0))
    ws, ws_bytes = None, 0
    if DETERMINISTIC:
        ws_bytes = _lib.load().xvit_layernorm_bwd_workspace_bytes(rows, d)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    ld_alt = _alt_ld(x, x_alt, seq_len)
    _run("layernorm_bwd", nbytes, "byte", lambda: _lib.load().xvit_layernorm_bwd(
        _ptr(dy), _rows2d(dy), _ptr(x), _ptr(x_alt), _rows2d(x), seq_len, ld_alt, _ptr(mean), _ptr(rstd), _ptr(gamma),
        _ptr(dres), _rows2d(dres) if dres is not None else 0, _ptr(dx), d, _ptr(dxb), d,
        _ptr(dgamma), _ptr(dbeta), _ptr(dxsum), _ptr(dressum), rows, d, _ptr(ws), ws_bytes, _stream()), "xvit_layernorm_bwd")
    return dx, dxb


def attn_fwd(qkv, B, N, H, scale, dropout=(0.0, 0)):
    """qkv bf16 [B*N, 3d] (q | k | v along columns) -> (o bf16 [B*N, d], lse fp32 [B, H, N]).  dropout = (p, seed): dropout
    on the attention probabilities (model.py:169); the mask is that of ops.dropout on a [B, H, N, N] tensor with that seed."""
    d = qkv.shape[1] // 3
    dh = d // H
    o = torch.empty(B * N, d, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    ld = _rows2d(qkv)
    p = qkv.data_ptr()
    lib = _lib.load()
    need = lib.xvit_attn_fwd_workspace_bytes(B, H, N) if dropout[0] == 0.0 else 0    # > 0: the CLS-peel form (include/xvit.h)
    ws = torch.empty(need // 4, dtype=torch.float32, device=qkv.device) if need else None
    _run("attn_fwd", 4.0 * B * H * N * N * dh, "flop",
         lambda: lib.xvit_attn_fwd(p, p + 2 * d, p + 4 * d, N * ld, ld, _ptr(o), N * d, d, _ptr(lse), B, H, N, dh, scale,
                                   float(dropout[0]), int(dropout[1]), _ptr(ws), need, _stream()),
         "xvit_attn_fwd")
    return o, lse


def attn_fwd_fp8(qkv, B, N, H, scale):
    """MX-fp8 forward attention (xvit_attn_fwd_fp8): qkv bf16 [B*N, 3d] -> (o bf16 [B*N, d], lse fp32 [B, H, N]); opt-in."""
    d = qkv.shape[1] // 3
    dh = d // H
    o = torch.empty(B * N, d, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    ld = _rows2d(qkv)
    need = _lib.load().xvit_attn_fp8_workspace_bytes(B, H, N, dh)
    ws = torch.empty(need, dtype=torch.uint8, device=qkv.device)
    p = _ptr(qkv)
    _run("attn_fwd_fp8", 4.0 * B * H * N * N * dh, "flop",
         lambda: _lib.load().xvit_attn_fwd_fp8(p, p + 2 * d, p + 4 * d, N * ld, ld, _ptr(o), N * d, d, _ptr(lse), B, H, N, dh, scale, _ptr(ws), need, _stream()),
         "xvit_attn_fwd_fp8")
    return o, lse


def attn_bwd(qkv, o, d_o, lse, B, N, H, scale, dropout=(0.0, 0)):
    """-> dqkv bf16 [B*N, 3d]."""
    d = qkv.shape[1] // 3
    dh = d // H
    dqkv = torch.empty_like(qkv)
    lib = _lib.load()
    need = lib.xvit_attn_bwd_workspace_bytes(B, H, N)
    ws = torch.empty(need // 4, dtype=torch.float32, device=qkv.device)   # rowsum(do*o) | -lse*log2e | CLS-peel partials
    ld = _rows2d(qkv)
    assert dqkv.stride(0) == ld and _rows2d(o) == d and _rows2d(d_o) == d
    p, g = qkv.data_ptr(), dqkv.data_ptr()
    _run("attn_bwd", 10.0 * B * H * N * N * dh, "flop",
         lambda: lib.xvit_attn_bwd(p, p + 2 * d, p + 4 * d, N * ld, ld, _ptr(o), _ptr(d_o), N * d, d, _ptr(lse), _ptr(ws), need,
                                   g, g + 2 * d, g + 4 * d, B, H, N, dh, scale, float(dropout[0]), int(dropout[1]), _stream()), "xvit_attn_bwd")
    return dqkv


def cls_xattn_fwd(q, kv, B, N, H, scale, dropout=(0.0, 0), want_f32=False):
    """q bf16 or fp32 [B, d]; kv bf16 [B*N, 2d] (k | v) -> (o bf16 [B, d], p fp32 [B, H, N]) (+ o fp32 with want_f32)."""
    d = q.shape[1]
    o = torch.empty(B, d, dtype=torch.bfloat16, device=q.device)
    of = torch.empty(B, d, dtype=torch.float32, device=q.device) if want_f32 else None
    p = torch.empty(B, H, N, dtype=torch.float32, device=q.device)
    ld = _rows2d(kv)
    kp = kv.data_ptr()
    qb, qf = (q, None) if q.dtype == torch.bfloat16 else (None, q)
    _run("cls_xattn_fwd", B * N * 2.0 * d * 2, "byte",
         lambda: _lib.load().xvit_cls_xattn_fwd(_ptr(qb), _rows2d(qb) if qb is not None else 0, _ptr(qf), _rows2d(qf) if qf is not None else 0,
                                                kp, kp + 2 * d, N * ld, ld, _ptr(o), d, _ptr(of), d, _ptr(p), B, H, N, d // H, scale,
                                                float(dropout[0]), int(dropout[1]), _stream()), "xvit_cls_xattn_fwd")
    return (o, p, of) if want_f32 else (o, p)


def cls_xattn_bwd(q, kv, p, d_o, B, N, H, scale, dropout=(0.0, 0), low_rank=False):
    """-> (dq fp32 [B, d], dkv bf16 [B*N, 2d])   or, low_rank: (dq, coef fp32 [B, N, 2H]) with dk[n] = coef[.., h] q_h and
    dv[n] = coef[.., H + h] dO_h (include/xvit.h): the input of xattn_kv_dgrad."""
    d = q.shape[1]
    dq = torch.empty(B, d, dtype=torch.float32, device=q.device)
    ld = _rows2d(kv)
    kp = kv.data_ptr()
    if low_rank:
        out = torch.empty(B, N, 2 * H, dtype=torch.float32, device=q.device)
        gk = gv = None
        work = B * N * 2.0 * d * 2
    else:
        out = torch.empty_like(kv)
        assert out.stride(0) == ld
        gk, gv = out.data_ptr(), out.data_ptr() + 2 * d
        work = B * N * 2.0 * d * 2 * 2
    _run("cls_xattn_bwd", work, "byte",
         lambda: _lib.load().xvit_cls_xattn_bwd(_ptr(q), _rows2d(q), kp, kp + 2 * d, N * ld, ld, _ptr(p), _ptr(d_o), _rows2d(d_o), _ptr(dq), d,
                                                gk, gv, _ptr(out) if low_rank else None, B, H, N, d // H, scale, float(dropout[0]), int(dropout[1]), _stream()),
         "xvit_cls_xattn_bwd")
    return dq, out


def head_rows(x, W, out, H, out_bf16=None):
    """out[b, h, :] = x[b, 64h:64h+64] @ W[64h:64h+64, :] (xvit_head_rows; fp32, per head).  x fp32 [B, d]; W fp32 [d, d] (a master
    weight); out fp32 and out_bf16 (optional): any tensors indexable as [b, h, c] with a unit last stride — e.g. a [H, B, d] slab
    transposed to [B, H, d], or a [B, 16, d] GEMM operand buffer, whose head rows H .. 15 the kernel zeroes."""
    B, d = x.shape
    assert x.dtype == torch.float32 and W.dtype == torch.float32 and out.dtype == torch.float32 and out.shape[-1] == d and out.stride(-1) == 1
    ob = out_bf16
    assert ob is None or (ob.dtype == torch.bfloat16 and ob.stride(-1) == 1)
    _run("head_linear", 2.0 * B * d * 64, "flop",
         lambda: _lib.load().xvit_head_rows(_ptr(x), _rows2d(x), _ptr(W), _rows2d(W), _ptr(out), out.stride(0), out.stride(1), _ptr(ob),
                                            ob.stride(0) if ob is not None else 0, ob.stride(1) if ob is not None else 0, ob.shape[1] if ob is not None else 0,
                                            B, H, d, _stream()), "xvit_head_rows")
    return out


def head_cols(t, W, H, row_scale=None, bias=None, want_bf16=False, bias_scale=None):
    """out[b, 64h+e] = row_scale[b, h] * (t[b, h, :] . W[64h+e, :]) + bias_scale[b, h] * bias[64h+e] (xvit_head_cols).  t fp32 [B, >=H, d] -> (fp32 [B, d], bf16 | None)."""
    B, _, d = t.shape
    assert t.dtype == torch.float32 and t.stride(2) == 1 and W.dtype == torch.float32
    assert bias_scale is None or (bias_scale.dtype == torch.float32 and bias_scale.stride(-1) == 1 and bias_scale.shape == (B, H))
    out = torch.empty(B, d, dtype=torch.float32, device=t.device)
    ob = torch.empty(B, d, dtype=torch.bfloat16, device=t.device) if want_bf16 else None
    _run("head_linear", 2.0 * B * d * 64, "flop",
         lambda: _lib.load().xvit_head_cols(_ptr(t), t.stride(0), t.stride(1), _ptr(W), _rows2d(W), _ptr(row_scale), row_scale.stride(0) if row_scale is not None else 0,
                                            _ptr(bias), _ptr(bias_scale), bias_scale.stride(0) if bias_scale is not None else 0, _ptr(out), d, _ptr(ob), d, B, H, d,
                                            _stream()), "xvit_head_cols")
    return out, ob


def head_wgrad(x, t, H, row_scale=None, out=None):
    """dW[64h+e, :] = sum_b x[b, 64h+e] row_scale[b, h] t[b, h, :] (xvit_head_wgrad) -> fp32 [d, d]."""
    B, d = x.shape
    assert x.dtype == torch.float32 and t.dtype == torch.float32 and t.stride(2) == 1 and t.shape[0] == B and t.shape[2] == d
    dW = out if out is not None else torch.empty(d, d, dtype=torch.float32, device=x.device)
    assert dW.shape == (d, d) and dW.is_contiguous() and dW.dtype == torch.float32
    _run("head_linear", 2.0 * B * d * 64, "flop",
         lambda: _lib.load().xvit_head_wgrad(_ptr(x), _rows2d(x), _ptr(t), t.stride(0), t.stride(1), _ptr(row_scale), row_scale.stride(0) if row_scale is not None else 0,
                                             _ptr(dW), d, B, H, d, _stream()), "xvit_head_wgrad")
    return dW


def cls_softmax_fwd(s, H, scale, dropout=None):
    """s fp32 [B, N, 16] (scores of the H heads in the first columns) -> (e bf16 [B, N, 16] = exp(scale (s - max_n s)), zero past H; rz fp32 [B, H] = 1 / sum_n e).
    dropout = (p, seed) on the probabilities (model_cross.py:97): -> (e, stat fp32 [3, B, H] = (rz, rz / (1 - p), rz / (1 - p) * sum_n e_kept), e_kept bf16 [B, N, 16])."""
    B, N, ld = s.shape
    assert s.dtype == torch.float32 and s.is_contiguous() and ld == 16
    e = torch.empty(B, N, 16, dtype=torch.bfloat16, device=s.device)
    drop = dropout is not None and dropout[0] > 0.0
    rz = torch.empty((3, B, H) if drop else (B, H), dtype=torch.float32, device=s.device)
    em = torch.empty_like(e) if drop else None
    _run("cls_softmax", B * N * 16 * 6.0, "byte",
         lambda: _lib.load().xvit_cls_softmax_fwd(_ptr(s), ld, _ptr(e), 16, _ptr(rz), B, H, N, scale, _ptr(em), float(dropout[0]) if drop else 0.0,
                                                  int(dropout[1]) if drop else 0, _stream()), "xvit_cls_softmax_fwd")
    return (e, rz, em) if drop else (e, rz)


def head_bias_grad(x, w, H):
    """out[j] = sum_b x[b, j] w[b, j // 64] (xvit_head_bias_grad): bv's gradient when the weights in front of it are w, not one."""
    B, d = x.shape
    assert x.dtype == torch.float32 and w.dtype == torch.float32 and w.shape == (B, H) and x.stride(1) == 1 and w.stride(1) == 1
    out = torch.empty(d, dtype=torch.float32, device=x.device)
    _run("head_linear", 2.0 * B * d, "flop", lambda: _lib.load().xvit_head_bias_grad(_ptr(x), x.stride(0), _ptr(w), w.stride(0), _ptr(out), B, H, d, _stream()), "xvit_head_bias_grad")
    return out


def cls_softmax_bwd(e, rz, dp, H, scale, dropout=None):
    """-> (coef fp32 [B, N, 2H] = (ds | p'), ds bf16 [B, N, 16]); p = e rz, ds = scale p (dp~ - sum_n p dp~); with dropout = (p, seed) dp~ = m dp / (1 - p) and
    p' = m p / (1 - p) (the forward's mask, regenerated), else dp~ = dp, p' = p.  rz: fp32 [B, H] (row 0 of the forward's stat block under dropout)."""
    B, N, _ = e.shape
    drop = dropout is not None and dropout[0] > 0.0
    assert rz.shape == (B, H) and rz.is_contiguous()
    assert e.dtype == torch.bfloat16 and e.is_contiguous() and dp.dtype == torch.float32 and dp.is_contiguous() and dp.shape == (B, N, 16)
    coef = torch.empty(B, N, 2 * H, dtype=torch.float32, device=e.device)
    dsb = torch.empty(B, N, 16, dtype=torch.bfloat16, device=e.device)
    _run("cls_softmax", B * N * (16 * 8.0 + 2 * H * 4.0), "byte",
         lambda: _lib.load().xvit_cls_softmax_bwd(_ptr(e), 16, _ptr(rz), _ptr(dp), 16, _ptr(coef), _ptr(dsb), 16, B, H, N, scale, float(dropout[0]) if drop else 0.0,
                                                  int(dropout[1]) if drop else 0, _stream()), "xvit_cls_softmax_bwd")
    return coef, dsb


def xattn_kv_dgrad(coef, R, B, N, H, d):
    """dhn[b, n, :] (bf16) = sum_j coef[b, n, j] R[j, b, :] (xvit_xattn_kv_dgrad); coef fp32 [B, N, 2H], R fp32 [2H, B, d]."""
    assert coef.is_contiguous() and R.is_contiguous() and R.shape == (2 * H, B, d)
    dhn = torch.empty(B * N, d, dtype=torch.bfloat16, device=coef.device)
    _run("xattn_kv_dgrad", B * N * d * 2.0 + B * N * 2 * H * 4.0, "byte",
         lambda: _lib.load().xvit_xattn_kv_dgrad(_ptr(coef), _ptr(R), _ptr(dhn), d, B, H, N, d, _stream()), "xvit_xattn_kv_dgrad")
    return dhn


def patchify(img, patch, pad_cls_row=False, concat=False):
    """img [B, M, 1, D, H, W] fp32|bf16 contiguous -> bf16 patch rows.
      default            [M, B*P, pd]           per modality, no padding
      pad_cls_row        [M, B*(P+1), pd]       per modality, a zero row in front of every sample (ModelCross)
      concat             [B*(M*P+1), pd]        all modalities of a sample in one sequence behind one zero row (ModelVIT)"""
    assert img.dim() == 6 and img.shape[2] == 1 and img.is_contiguous()
    B, M, _, D, H, W = img.shape
    dp, hp, wp = patch
    P, pd = (D // dp) * (H // hp) * (W // wp), dp * hp * wp
    if concat:
        rows = M * P + 1
        out = torch.empty(B * rows, pd, dtype=torch.bfloat16, device=img.device)
        place = (rows, P, 1, B, rows)
    else:
        pad = int(bool(pad_cls_row))
        out = torch.empty(M, B * (P + pad), pd, dtype=torch.bfloat16, device=img.device)
        place = (P + pad, B * (P + pad), pad, M * B if pad else 0, P + pad)
    _run("patchify", img.numel() * (img.element_size() + 2.0), "byte",
         lambda: _lib.load().xvit_patchify(_ptr(img), _dt(img), _ptr(out), B, M, D, H, W, dp, hp, wp, *place, _stream()), "xvit_patchify")
    return out


def _patch_geom(img, patch, cls_rows):
    B, M, _, D, H, W = img.shape
    g = _lib.PatchGeom()
    g.B, g.M, g.D, g.H, g.W = B, M, D, H, W
    g.dp, g.hp, g.wp = patch
    g.cls_rows = int(cls_rows)
    return g


def patch_embed_supported(img, patch, d, cls_rows=1):
    """True when the fused gather kernels (xvit_patch_embed_*) take this input: a contiguous bf16 [B, M, 1, D, H, W] volume
    tensor and a geometry xvit_patch_embed_supported accepts (include/xvit.h).  XVIT_PATCH_EMBED=unfused forces the
    patchify + GEMM path (A/B measurements)."""
    if os.environ.get("XVIT_PATCH_EMBED", "fused") == "unfused":
        return False
    if img.dim() != 6 or img.shape[2] != 1 or img.dtype != torch.bfloat16 or not img.is_contiguous() or not img.is_cuda:
        return False
    return _lib.load().xvit_patch_embed_supported(C.byref(_patch_geom(img, patch, cls_rows)), int(d)) == 1


def patch_embed_fwd(img, patch, w_b, bias, pos, cls_rows=1):
    """x [M*B*(cls_rows + P), d] fp32 = patches(img) W^T + bias + pos[token], the patch matrix never stored
    (model_cross.py:193-197).  w_b: bf16 [d, pd]; pos: fp32 [cls_rows + P, d] or None."""
    g = _patch_geom(img, patch, cls_rows)
    B, M, _, D, H, W = img.shape
    P, pd = (D // patch[0]) * (H // patch[1]) * (W // patch[2]), patch[0] * patch[1] * patch[2]
    d = w_b.shape[0]
    assert w_b.dtype == torch.bfloat16 and w_b.shape[1] == pd and (pos is None or pos.shape == (cls_rows + P, d))
    rows = M * B * (cls_rows + P)
    x = torch.empty(rows, d, dtype=torch.float32, device=img.device)
    _run("gemm_big_nt", 2.0 * M * B * P * d * pd, "flop",
         lambda: _lib.load().xvit_patch_embed_fwd(_ptr(img), C.byref(g), _ptr(w_b), w_b.stride(0), _ptr(bias) if bias is not None else None,
                                                  _ptr(pos) if pos is not None else None, pos.stride(0) if pos is not None else 0, _ptr(x), x.stride(0), d, _stream()),
         "xvit_patch_embed_fwd")
    return x


def patch_embed_wgrad(img, patch, dx_b, cls_rows=1, out=None):
    """dW [d, pd] fp32 = sum over patch rows of dx[row]^T patch(row); dx_b: bf16 [M*B*(cls_rows + P), d]."""
    g = _patch_geom(img, patch, cls_rows)
    B, M, _, D, H, W = img.shape
    P, pd = (D // patch[0]) * (H // patch[1]) * (W // patch[2]), patch[0] * patch[1] * patch[2]
    d = dx_b.shape[1]
    assert dx_b.dtype == torch.bfloat16 and dx_b.shape[0] == M * B * (cls_rows + P) and dx_b.is_contiguous()
    need = _lib.load().xvit_patch_embed_wgrad_workspace_bytes(C.byref(g), d)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=img.device)
    dW = out if out is not None else torch.empty(d, pd, dtype=torch.float32, device=img.device)
    assert dW.shape == (d, pd) and dW.is_contiguous() and dW.dtype == torch.float32
    _run("gemm_big_tn+splitk", 2.0 * M * B * P * d * pd, "flop",
         lambda: _lib.load().xvit_patch_embed_wgrad(_ptr(img), C.byref(g), _ptr(dx_b), dx_b.stride(0), _ptr(dW), dW.stride(0), d, _ptr(ws), need, _stream()),
         "xvit_patch_embed_wgrad")
    return dW


def cls_row_fwd(cls, pos, x, MB, N, d):
    _lib.check(_lib.load().xvit_cls_row_fwd(_ptr(cls), _ptr(pos), _ptr(x), MB, N, d, _stream()), "xvit_cls_row_fwd")


def embed_bwd(dx, dpos, dcls, MB, N, d):
    _run("embed_bwd", float(MB) * N * d * 4, "byte", lambda: _lib.load().xvit_embed_bwd(_ptr(dx), _ptr(dpos), _ptr(dcls), MB, N, d, _stream()), "xvit_embed_bwd")


def cast_bf16(src, out=None):
    """fp32 -> bf16 (contiguous)."""
    assert src.dtype == torch.float32 and src.is_contiguous()
    out = out if out is not None else torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    assert out.is_contiguous() and out.numel() == src.numel()
    _run("cast_f32_bf16", src.numel() * 6.0, "byte", lambda: _lib.load().xvit_cast_f32_bf16(_ptr(src), _ptr(out), src.numel(), _stream()), "xvit_cast_f32_bf16")
    return out


def add_cast(a, b):
    """-> (a + b fp32, its bf16 copy) in one pass (xvit_add_cast_f32_bf16); contiguous fp32 tensors of one shape, numel % 8 == 0."""
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    out = torch.empty_like(a)
    outb = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device)
    _run("add_cast", a.numel() * 14.0, "byte", lambda: _lib.load().xvit_add_cast_f32_bf16(_ptr(a), _ptr(b), _ptr(out), _ptr(outb), a.numel(), _stream()), "xvit_add_cast_f32_bf16")
    return out, outb


def rows_combine(dst, a=None, b=None, dst2=None):
    """dst[r, :] (and dst2[r, :]) = a[r, :] + b[r, :] (xvit_rows_combine): 2-D views [rows, d] with a unit last stride and any row stride
    (e.g. t[:, 0] of a [B, N, d] tensor), fp32 or bf16 each; a missing operand is zero; dst may be a or b."""
    rows, d = dst.shape

    def arg(t):
        if t is None:
            return None, 0, 0
        assert t.shape == (rows, d) and t.stride(1) == 1, f"rows_combine: need [rows, d] views with a unit last stride, got {tuple(t.shape)} {t.stride()}"
        return _ptr(t), _dt(t), t.stride(0)
    pd, dd, ld = arg(dst)
    p2, d2, l2 = arg(dst2)
    pa, da, la = arg(a)
    pb, db, lb = arg(b)
    _run("rows_combine", rows * d * 8.0, "byte", lambda: _lib.load().xvit_rows_combine(pd, dd, ld, p2, d2, l2, pa, da, la, pb, db, lb, rows, d, _stream()), "xvit_rows_combine")
    return dst


def colsum(x, out=None, accumulate=False):
    rows, n = x.shape
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=x.device)
        accumulate = False
    ws, ws_bytes = None, 0
    if DETERMINISTIC:
        ws_bytes = _lib.load().xvit_colsum_workspace_bytes(rows, n)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    _run("colsum", float(rows) * n * x.element_size(), "byte",
         lambda: _lib.load().xvit_colsum(_ptr(x), _dt(x), _rows2d(x), _ptr(out), rows, n, int(accumulate), _ptr(ws), ws_bytes, _stream()), "xvit_colsum")
    return out


def dropout(x, p, seed, out=None):
    assert x.is_contiguous()
    out = out if out is not None else torch.empty_like(x)
    _lib.check(_lib.load().xvit_dropout(_ptr(x), _ptr(out), _dt(x), x.numel(), p, seed, _stream()), "xvit_dropout")
    return out


def small_linear_fwd(x, W, b):
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().xvit_small_linear_fwd(_ptr(x), _rows2d(x), _ptr(W), _ptr(b), _ptr(y), M, N, K, _stream()), "xvit_small_linear_fwd")
    return y


def small_linear_bwd(dy, x, W, dW, db, z=None):
    """dx = (dy W) * gelu'(z) if z is given (x = gelu(z))."""
    M, K = x.shape
    N = W.shape[0]
    dx = torch.empty(M, K, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().xvit_small_linear_bwd(_ptr(dy), _ptr(x), _rows2d(x), _ptr(W), _ptr(z), _rows2d(z) if z is not None else 0,
                                                 _ptr(dx), K, _ptr(dW), _ptr(db), M, N, K, int(DETERMINISTIC), _stream()), "xvit_small_linear_bwd")
    return dx


def mean_ce(logits_m, labels, smoothing):
    M, B, Cn = logits_m.shape
    logits = torch.empty(B, Cn, dtype=torch.float32, device=logits_m.device)
    loss = torch.empty((), dtype=torch.float32, device=logits_m.device)
    dl = torch.empty_like(logits_m)
    _lib.check(_lib.load().xvit_mean_ce(_ptr(logits_m), _ptr(labels), smoothing, _ptr(logits), _ptr(loss), _ptr(dl), M, B, Cn, _stream()), "xvit_mean_ce")
    return logits, loss, dl


def resize_pad_crop_i16(vol, img_size, pad_value=-1.0):
    """int16 [B, M, Ds, Hs, Ws] (NIfTI voxels, on the GPU) -> bf16 [B, M, 1, D, H, W]: the reference's
    ResizeWithPadOrCropd(img_size, constant_values=-1) + float cast (dataset_ucsf.py:84-88,152-158) in one pass."""
    assert vol.dtype == torch.int16 and vol.dim() == 5 and vol.is_contiguous()
    B, M, Ds, Hs, Ws = vol.shape
    D, H, W = img_size
    out = torch.empty(B, M, 1, D, H, W, dtype=torch.bfloat16, device=vol.device)
    _run("resize_pad_crop_i16", vol.numel() * 2.0 + out.numel() * 2.0, "byte",
         lambda: _lib.load().xvit_resize_pad_crop_i16(_ptr(vol), _ptr(out), B * M, Ds, Hs, Ws, D, H, W, float(pad_value), _stream()),
         "xvit_resize_pad_crop_i16")
    return out
