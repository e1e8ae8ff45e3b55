"""CPU: the C-ABI library loads and exports every symbol include/xvit.h declares (no compute)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from xvit import _lib
    header = open(os.path.join(ROOT, "include", "xvit.h")).read()
    declared = sorted(set(re.findall(r"\b(xvit_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in xvit.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, (sorted(_lib.EXPORTS), declared)   # the ctypes table covers the header
    assert lib.xvit_version() == int(re.search(r"#define XVIT_VERSION (\d+)", header).group(1))


def test_argument_errors_do_not_launch():
    """Validation happens on the host before any launch: callable without a GPU."""
    import ctypes as C
    from xvit import _lib
    lib = _lib.load()
    a = _lib.GemmArgs()
    assert lib.xvit_gemm(C.byref(a), None) < 0
    assert b"M,N,K" in lib.xvit_last_error_string() or b"layout" in lib.xvit_last_error_string() or len(lib.xvit_last_error_string()) > 0
    assert lib.xvit_attn_fwd(None, None, None, 0, 0, None, 0, 0, None, 1, 1, 1, 64, 1.0, 0.0, 0, None, 0, None) < 0
    assert lib.xvit_layernorm_fwd(None, None, 0, 0, 0, None, None, 1e-5, None, 0, None, 0, None, None, 1, 8, None) < 0
    assert lib.xvit_linear_f32(None, 0, None, 0, None, None, 0, 4, 8, 24, 0, None, 0, None, 0, None, 0, 0.0, 0, None, 0, None) < 0
    assert lib.xvit_linear_f32_workspace_bytes(126, 768, 3072) > 0 and lib.xvit_linear_f32_workspace_bytes(0, 8, 16) == 0
    assert lib.xvit_set_option(b"no_such_option", 1) < 0 and lib.xvit_set_option(b"gemm_tile", 0) == 0
    assert lib.xvit_rows_combine(64, 1, 100, None, 0, 0, 64, 1, 768, None, 0, 0, 8, 768, None) < 0 and b"row stride" in lib.xvit_last_error_string()
    assert lib.xvit_rows_combine(64, 7, 768, None, 0, 0, None, 0, 0, None, 0, 0, 8, 768, None) < 0          # unknown dtype
    assert lib.xvit_add_cast_f32_bf16(64, 64, 64, 64, 12, None) < 0                                       # n not a multiple of 8


def test_attention_workspaces_and_peel_predicate():
    """Host-only: which shapes take the CLS-peel form (N = 64 m + 1, no probability dropout, large grids unless forced), the sizes of the
    caller-owned workspaces, and that too small a workspace is refused before any launch."""
    from xvit import _lib
    lib = _lib.load()
    fw, bw = lib.xvit_attn_fwd_workspace_bytes, lib.xvit_attn_bwd_workspace_bytes
    assert fw(126, 12, 513) == 126 * 12 * 16 * 80 * 4                 # one 80-float slot per wave: (N - 1) / 32 = 16 slots per (b, head)
    assert fw(126, 12, 512) == 0 and fw(126, 12, 3376) == 0 and fw(8, 12, 513) == 0    # not 64 m + 1 / a grid of 384 workgroups (< 768)
    assert fw(8, 12, 4097) > 0 and fw(16, 12, 513) > 0                 # configs[4]: 3072 workgroups; 768 workgroups
    assert bw(126, 12, 513) == (2 * 126 * 12 * 513 + 3 * 126 * 12 * 16 * 64) * 4
    assert bw(8, 12, 513) == 2 * 8 * 12 * 513 * 4
    try:
        assert lib.xvit_set_option(b"attn_peel", 2) == 0
        assert fw(8, 12, 513) > 0 and fw(2, 3, 65) > 0 and fw(2, 3, 1) == 0
        assert lib.xvit_set_option(b"attn_peel", 0) == 0
        assert fw(126, 12, 513) == 0
        assert lib.xvit_set_option(b"attn_peel", 3) < 0
    finally:
        lib.xvit_set_option(b"attn_peel", 1)
    # a workspace that is too small is an argument error (nothing is launched: callable without a GPU; the pointers are never dereferenced)
    d = 12 * 64
    rc = lib.xvit_attn_fwd(64, 64, 64, 513 * 3 * d, 3 * d, 64, 513 * d, d, 64, 126, 12, 513, 64, 0.125, 0.0, 0, 64, 1024, None)
    assert rc < 0 and b"workspace" in lib.xvit_last_error_string()
    rc = lib.xvit_attn_bwd(64, 64, 64, 513 * 3 * d, 3 * d, 64, 64, 513 * d, d, 64, 64, 1024, 64, 64, 64, 126, 12, 513, 64, 0.125, 0.0, 0, None)
    assert rc < 0 and b"workspace" in lib.xvit_last_error_string()


def test_head_linear_argument_errors_do_not_launch():
    """The low-rank fusion's per-head kernels (csrc/head_linear.hip) refuse bad shapes on the host: d != 64 H, more than 16 heads in the
    CLS softmax, misaligned strides, null pointers.  Pointers are never dereferenced (no launch): callable without a GPU."""
    from xvit import _lib
    lib = _lib.load()
    P = 256   # any non-null, 16-byte aligned "address"
    assert lib.xvit_head_rows(None, 768, P, 768, P, 768, 64, None, 0, 0, 0, 8, 12, 768, None) < 0 and b"null" in lib.xvit_last_error_string()
    assert lib.xvit_head_rows(P, 768, P, 768, P, 768, 64, None, 0, 0, 0, 8, 12, 700, None) < 0 and b"64 H" in lib.xvit_last_error_string()
    assert lib.xvit_head_rows(P, 766, P, 768, P, 768, 64, None, 0, 0, 0, 8, 12, 768, None) < 0            # ldx not a multiple of 4 floats / < d
    assert lib.xvit_head_cols(P, 768, 64, P, 766, None, 0, None, None, 0, P, 768, None, 0, 8, 12, 768, None) < 0   # ldw not a multiple of 4 floats
    assert lib.xvit_head_cols(P + 4, 768, 64, P, 768, None, 0, None, None, 0, P, 768, None, 0, 8, 12, 768, None) < 0   # t not 16-byte aligned
    assert lib.xvit_head_bias_grad(P, 512, P, 12, P, 8, 12, 768, None) < 0                                # ldx < d
    assert lib.xvit_head_bias_grad(P, 768, P, 8, P, 8, 12, 768, None) < 0                                 # ldw < H
    assert lib.xvit_head_wgrad(P, 512, P, 768, 64, None, 0, P, 768, 8, 12, 768, None) < 0                 # ldx < d
    assert lib.xvit_cls_softmax_fwd(P, 32, P, 16, P, 8, 32, 513, 0.125, None, 0.0, 0, None) < 0 and b"H <= 16" in lib.xvit_last_error_string()
    assert lib.xvit_cls_softmax_fwd(P, 12, P, 8, P, 8, 12, 513, 0.125, None, 0.0, 0, None) < 0            # lde < H
    assert lib.xvit_cls_softmax_fwd(P, 16, P, 16, P, 8, 12, 513, 0.125, None, 0.25, 1, None) < 0 and b"e_masked" in lib.xvit_last_error_string()
    assert lib.xvit_cls_softmax_bwd(P, 16, P, P, 12, P, P, 32, 8, 12, 513, 0.125, 0.0, 0, None) < 0       # ldb > 16
    assert lib.xvit_cls_softmax_bwd(P, 16, None, P, 12, P, P, 16, 8, 12, 513, 0.125, 0.0, 0, None) < 0    # null rz
    assert lib.xvit_cls_softmax_bwd(P, 16, P, P, 16, P, P, 16, 8, 12, 513, 0.125, 1.0, 0, None) < 0       # dropout_p out of range
    assert lib.xvit_set_dropout_epoch(P + 4) < 0 and lib.xvit_set_dropout_epoch(None) == 0                 # 8-byte aligned counter, NULL = off


def test_modules_have_reference_state_dict_keys():
    import ref_cpu as R
    import xvit
    for name in ("tiny", "small"):
        cfg = R.make_config(name)
        model = xvit.ModelCross(cfg)
        assert set(model.state_dict()) == set(R.make_state_dict(cfg)), name
        res = model.load_state_dict(R.make_state_dict(cfg), strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    cfgv = R.make_config("small", num_layers=2)
    assert set(xvit.ModelVIT(cfgv).state_dict()) == set(R.make_vit_state_dict(cfgv))
    from types import SimpleNamespace
    enc = xvit.Encoder(SimpleNamespace(hidden_size=256, transformer=dict(num_heads=4, mlp_dim=512, dropout_rate=0.0, attention_dropout_rate=0.0, num_layers=2)))
    assert set(enc.state_dict()) == set(R.make_encoder_state_dict(256, 512, 2))


def test_patch_embed_geometry_predicate_and_workspaces():
    """Host-only entry points of the fused patch embedding and the fp8 attention: which geometries the gather kernels take
    (include/xvit.h) and what scratch they ask for — no GPU needed."""
    import ctypes as C
    from xvit import _lib
    lib = _lib.load()

    def geom(B, M, vol, patch, cls=1):
        g = _lib.PatchGeom()
        g.B, g.M, (g.D, g.H, g.W), (g.dp, g.hp, g.wp), g.cls_rows = B, M, vol, patch, cls
        return g

    ok = lambda g, d: lib.xvit_patch_embed_supported(C.byref(g), d)   # noqa: E731
    assert ok(geom(126, 2, (128, 128, 128), (16, 16, 16)), 768) == 1          # configs[1]
    assert ok(geom(8, 2, (128, 128, 128), (8, 8, 8)), 768) == 1               # configs[4]
    assert ok(geom(2, 4, (240, 240, 240), (16, 16, 16)), 768) == 1            # configs[2]: 15 patches per axis (the weight gradient places k-rows per K-step)
    assert ok(geom(20, 4, (240, 240, 240), (16, 16, 16)), 768) == 0           # ... this volume tensor is beyond 2 GiB of bf16
    assert ok(geom(1, 1, (128, 128, 128), (16, 16, 16)), 768) == 0            # 513 rows: small-tile path
    assert ok(geom(126, 2, (128, 128, 128), (16, 16, 16)), 192) == 0          # d not a multiple of 256
    assert ok(geom(126, 2, (128, 128, 128), (16, 16, 4)), 768) == 0           # 8-byte runs
    assert ok(geom(126, 2, (128, 128, 120), (16, 16, 16)), 768) == 0          # W not divisible
    assert ok(geom(300, 2, (128, 128, 128), (16, 16, 16)), 768) == 0          # volume tensor beyond 2 GiB
    assert ok(geom(126, 2, (128, 128, 128), (16, 16, 16), cls=2), 768) == 0
    g = geom(126, 2, (128, 128, 128), (16, 16, 16))
    ws = lib.xvit_patch_embed_wgrad_workspace_bytes(C.byref(g), 768)
    assert ws > 0 and ws % (768 * 4096 * 4) == 0                              # split-K slabs of the [768, 4096] gradient
    assert lib.xvit_patch_embed_wgrad_workspace_bytes(C.byref(geom(1, 1, (128, 128, 128), (16, 16, 16))), 768) == 0
    # unsupported geometry: refused before any launch
    bad = geom(1, 1, (128, 128, 128), (16, 16, 16))
    assert lib.xvit_patch_embed_fwd(1, C.byref(bad), 1, 4096, None, None, 0, 1, 768, 768, None) < 0
    assert b"not supported" in lib.xvit_last_error_string()
    # fp8 attention scratch: q8 + k8 + v8t (B H Np 64 each) + scales, Np = N rounded up to 64
    need = lib.xvit_attn_fp8_workspace_bytes(8, 12, 4097, 64)
    assert 3 * 8 * 12 * 4160 * 64 <= need < 3.2 * 8 * 12 * 4160 * 64
    assert lib.xvit_attn_fp8_workspace_bytes(8, 12, 4097, 32) == 0
    assert lib.xvit_attn_fwd_fp8(None, None, None, 0, 0, None, 0, 0, None, 1, 1, 64, 64, 0.125, None, 0, None) < 0
