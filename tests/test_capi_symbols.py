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
    assert lib.xvit_attn_fwd(None, None, None, 0, 0, None, 0, 0, None, 1, 1, 1, 64, 1.0, 0.0, 0, None) < 0
    assert lib.xvit_layernorm_fwd(None, None, 0, 0, 0, None, None, 1e-5, None, 0, None, 0, None, None, 1, 8, None) < 0
    assert lib.xvit_linear_f32(None, 0, None, 0, None, None, 0, 4, 8, 24, 0, None, 0, None, 0, None, 0, 0.0, 0, None, 0, None) < 0
    assert lib.xvit_linear_f32_workspace_bytes(126, 768, 3072) > 0 and lib.xvit_linear_f32_workspace_bytes(0, 8, 16) == 0
    assert lib.xvit_set_option(b"no_such_option", 1) < 0 and lib.xvit_set_option(b"gemm_tile", 0) == 0


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
