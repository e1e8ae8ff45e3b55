"""GPU: on-device epoch statistics (xvit_binary_metrics_step / xvit.metrics) against the oracle restatement of the
reference's log_stats (model_cross.py:243-255 -> utils.py:18-62), which tests/test_oracle.py pins against scikit-learn."""
import pytest
import torch

import ref_cpu as R
from _util import dev

pytestmark = pytest.mark.gpu


def _steps(seed, sizes):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(b, 2, generator=g) * 2.0, torch.randint(0, 2, (b,), generator=g)) for b in sizes]


def test_epoch_metrics_match_oracle_over_ragged_steps():
    from xvit.metrics import KEYS, BinaryEpochMetrics
    steps = _steps(0, (126, 126, 37, 1, 8, 300))                  # the last batch of an epoch is ragged; B = 1 is legal
    acc = BinaryEpochMetrics(dev())
    for logits, labels in steps:
        acc.update(logits.to(dev()), labels.to(dev()))
    got = acc.compute("train", sync_dist=False)
    ref = R.epoch_metrics(steps)
    for k in KEYS:
        assert abs(got[f"train_{k}"] - ref[k]) < 1e-9, (k, got[f"train_{k}"], ref[k])
    counts = [R.binary_step_metrics(a, b)["counts"] for a, b in steps]
    conf = got["train_confusion"]
    assert (conf["tn"], conf["fp"], conf["fn"], conf["tp"]) == tuple(sum(c[i] for c in counts) for i in range(4))
    assert conf["samples"] == sum(len(b) for _, b in steps) and conf["steps"] == len(steps)
    acc.reset()
    assert acc.compute(sync_dist=False)["confusion"]["samples"] == 0


@pytest.mark.parametrize("case", ["ties", "no_negatives", "no_positives", "saturated"])
def test_edge_cases_match_oracle(case):
    from xvit.metrics import KEYS, BinaryEpochMetrics
    if case == "ties":
        logits = torch.tensor([[0.5, 0.5], [1.0, -1.0], [1.0, -1.0], [-2.0, 2.0], [-2.0, 2.0], [0.0, 0.0]])
        labels = torch.tensor([1, 0, 1, 1, 0, 0])
    elif case == "no_negatives":
        logits, labels = torch.randn(9, 2), torch.ones(9, dtype=torch.int64)
    elif case == "no_positives":
        logits, labels = torch.randn(9, 2), torch.zeros(9, dtype=torch.int64)
    else:   # probabilities that round to exactly 1.0 / 0.0 in fp32 tie with each other, as in the reference's softmax
        logits = torch.tensor([[-40.0, 40.0], [-30.0, 50.0], [40.0, -40.0], [35.0, -45.0], [0.0, 0.1]])
        labels = torch.tensor([1, 0, 0, 1, 1])
    acc = BinaryEpochMetrics(dev())
    acc.update(logits.to(dev()), labels.to(dev()))
    got, ref = acc.compute(sync_dist=False), R.binary_step_metrics(logits, labels)
    for k in KEYS:
        assert abs(got[k] - ref[k]) < 1e-9, (k, got[k], ref[k])


def test_argument_errors_and_input_dtypes():
    from xvit import _lib
    from xvit.metrics import BinaryEpochMetrics
    acc = BinaryEpochMetrics(dev())
    with pytest.raises(ValueError):
        acc.update(torch.randn(4, 3, device=dev()), torch.zeros(4, dtype=torch.int64, device=dev()))
    with pytest.raises(RuntimeError):
        acc.update(torch.randn(4, 2), torch.zeros(4, dtype=torch.int64))
    st = torch.zeros(16, dtype=torch.float64, device=dev())
    lg = torch.randn(4, 2, device=dev())
    lb = torch.zeros(4, dtype=torch.int64, device=dev())
    assert _lib.load().xvit_binary_metrics_step(lg.data_ptr(), 2, lb.data_ptr(), 4, 3, st.data_ptr(), 0) < 0      # C != 2
    assert _lib.load().xvit_binary_metrics_step(lg.data_ptr(), 2, lb.data_ptr(), 0, 2, st.data_ptr(), 0) < 0      # empty batch
    # bf16 logits / int32 labels (what a mixed-precision loop hands over) are converted on the way in
    steps = _steps(4, (33,))
    acc.update(steps[0][0].to(dev(), torch.bfloat16), steps[0][1].to(dev(), torch.int32))
    ref = R.binary_step_metrics(steps[0][0].bfloat16().float(), steps[0][1])
    assert abs(acc.compute(sync_dist=False)["acc"] - ref["acc"]) < 1e-9


def test_model_training_step_accumulates_like_the_reference_logs():
    """ModelCross.training_step -> log_stats (model_cross.py:258-265): after an 'epoch' of three steps epoch_stats
    returns what the reference would have logged, computed from the very logits the steps produced."""
    import xvit
    cfg = R.make_config("tiny")
    model = xvit.ModelCross(cfg).to(dev())
    model.eval()
    seen = []
    hook = model.register_forward_hook(lambda m, i, o: seen.append((o[0].detach().float().cpu(), i[1].cpu())))
    for seed, b in ((1, 5), (2, 5), (3, 2)):
        img, labels = R.make_inputs(cfg, b, seed=seed)
        with torch.no_grad():
            model.validation_step((img.to(dev()), labels.to(dev())), 0)
    hook.remove()
    got = model.epoch_stats("val")
    ref = R.epoch_metrics(seen)
    for k, v in ref.items():
        assert abs(got[f"val_{k}"] - v) < 1e-9, k
    assert got["val_confusion"]["samples"] == 12
    assert model.epoch_stats("val")["val_confusion"]["samples"] == 0          # reset by the read
