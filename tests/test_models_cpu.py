import itertools

import numpy as np
import pytest
import torch

from mitune.data import (create_dummy_data, features, get_data_loaders, split_into_intervals)
from mitune.models.transformer import (CustomEncoderLayer, DepthwiseSeparableConv, LinearAttention,
                                       PositionalEncoding, SimpleTransformerModel, TransformerModel, sinusoid_table)
from mitune.ops.attention import attention_reference
from mitune.train.losses import elementwise_loss, elementwise_loss_grad, get_criterion, mape_loss


def test_feature_lists_match_the_workload_schema():
    assert len(features.features_1) == 9 and len(features.features) == 81
    assert len(set(features.features)) == 81
    assert features.heart_rate_features_2[0] == "HeartRate_15_Mean"          # no unit suffix for heart rate
    assert features.sleep_features_2[1] == "Sleep_15min_Std"
    assert features.steps_features_2[-1] == "Steps_1440min_Std"
    assert features.temporal_features == ["MinuteOfDay_Sin", "MinuteOfDay_Cos", "DayOfWeek_Sin", "DayOfWeek_Cos",
                                          "Is_Weekend"]


def test_split_into_intervals_contract():
    data = np.arange(960 * 3, dtype=np.float32).reshape(960, 3)
    w = split_into_intervals(data, 96, 96)
    assert w.shape == (10, 96, 3)
    assert np.array_equal(w[1], data[96:192])
    w2 = split_into_intervals(data, 10, 3)
    assert w2.shape == ((960 - 10) // 3 + 1, 10, 3)
    assert np.array_equal(w2[5], data[15:25])
    assert split_into_intervals(data[:5], 10, 1).shape == (0, 10, 3)


def test_positional_encoding_values_and_growth():
    pe = PositionalEncoding(16, max_seq_length=8, dropout=0.0)
    x = torch.zeros(2, 12, 16)                    # longer than the table: grows instead of failing
    y = pe(x)
    t = sinusoid_table(12, 16)
    assert torch.allclose(y[0], t) and torch.allclose(y[1], t)
    assert t[3, 0].item() == pytest.approx(np.sin(3.0)) and t[3, 1].item() == pytest.approx(np.cos(3.0))


def test_depthwise_separable_conv_keeps_width():
    m = DepthwiseSeparableConv(24, 96)
    assert m(torch.randn(3, 20, 24)).shape == (3, 20, 24)


def test_linear_attention_is_rowwise_convex_combination():
    torch.manual_seed(0)
    q, k = torch.randn(2, 2, 7, 8), torch.randn(2, 2, 7, 8)
    v = torch.ones(2, 2, 7, 8)
    o = attention_reference(q, k, v, "relu")
    assert torch.all(o <= 1.0 + 1e-5) and torch.all(o >= 0)
    o2 = attention_reference(q, k, v, "softmax")
    assert torch.allclose(o2, torch.ones_like(o2), atol=1e-5)


def test_softmax_attention_matches_torch_sdpa():
    torch.manual_seed(1)
    q, k, v = (torch.randn(2, 4, 50, 16) for _ in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    assert torch.allclose(attention_reference(q, k, v), ref, atol=1e-5)


@pytest.mark.parametrize("attention_type,dsc,shared", list(itertools.product(
    ["scaled_dot_product", "linear_attention", "multi_head_attention"], [False, True], [False, True])))
def test_every_transformer_variant_trains(attention_type, dsc, shared):
    torch.manual_seed(0)
    m = TransformerModel(input_size=10, num_heads=4, num_encoder_layers=2, d_model=32, dim_feedforward=64, dropout=0.1,
                         max_seq_length=50, attention_type=attention_type, attn_kernel_size=5,
                         depthwise_separable_conv=dsc, key_dim_scaling=0.5, shared_weights=shared,
                         stochastic_depth_rate=0.1)
    x, y = torch.randn(6, 96, 10), torch.randn(6)
    out = m(x)
    assert out.shape == (6, 1)
    loss = get_criterion("huber")(out.squeeze(-1), y)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    if shared:
        assert len(m.layers) == 1


def test_simple_transformer_shapes_and_eval_determinism():
    m = SimpleTransformerModel(10, 32, 4, 2, dropout=0.3).eval()
    x = torch.randn(5, 50, 10)
    assert m(x).shape == (5,)
    assert torch.equal(m(x), m(x))


@pytest.mark.parametrize("name", ["mse", "mae", "huber", "mape"])
def test_losses_and_their_gradients(name):
    torch.manual_seed(2)
    o = torch.randn(64, requires_grad=True)
    t = torch.randn(64) * 2
    ref = get_criterion(name)(o, t)
    assert torch.allclose(elementwise_loss(name, o, t).mean(), ref, atol=1e-6)
    ref.backward()
    assert torch.allclose(elementwise_loss_grad(name, o.detach(), t) / 64, o.grad, atol=1e-6)


def test_mape_uses_absolute_denominator():
    o, t = torch.tensor([-1.0]), torch.tensor([-2.0])
    assert mape_loss(o, t).item() == pytest.approx(50.0)


def test_dummy_data_and_glucose_pipeline():
    tr, va = create_dummy_data(num_samples=200, seq_len=12, input_size=5)
    xb, yb = next(iter(tr))
    assert xb.shape == (32, 12, 5) and yb.shape == (32,)
    assert len(tr.dataset) == 160 and len(va.dataset) == 40
    tr, va = get_data_loaders("MMCS0002")
    xb, yb = next(iter(tr))
    assert xb.shape[1:] == (96, 81) and yb.dim() == 1
    n = len(tr.dataset) + len(va.dataset)
    assert len(va.dataset) == int(np.ceil(0.3 * n))


def test_packed_sequence_family_trains_on_cpu():
    """The packed executor takes sequence data [N, S, F] and the sample Transformer family (Torch engine on the CPU)."""
    import math
    from mitune import tune
    from mitune.data import create_dummy_data
    from mitune.packed import PackedTrainable
    tl, vl = create_dummy_data(num_samples=96, seq_len=8, input_size=5)
    tr = PackedTrainable(model="simple_transformer", train=tuple(tl.dataset.tensors), val=tuple(vl.dataset.tensors),
                         max_epochs=2, dtype="f32", capacity=2, model_kwargs={"batch_size": 16, "optimizer": "adamw"})
    assert tr.in_features == 5
    an = tune.run(tr, config={"d_model": tune.choice([16, 32]), "num_heads": 2, "num_layers": 1, "dropout": 0.1,
                              "learning_rate": 1e-3}, num_samples=3, local_dir=None, verbose=0, seed=0,
                  metric="validation_loss", mode="min", device=torch.device("cpu"))
    assert len(an.trials) == 3 and all(t.training_iteration == 2 for t in an.trials)
    assert math.isfinite(an.best_result["validation_loss"])
    with pytest.raises(ValueError):
        PackedTrainable(model="mlp", train=tuple(tl.dataset.tensors), val=tuple(vl.dataset.tensors))


def test_rng_scope_switches_the_counter():
    from mitune.ops import rng
    mine = rng.new_counter("cpu", 5)
    base = rng.counter("cpu")
    before = int(base.ctr)
    with rng.scope(mine):
        assert rng.counter("cpu") is mine
        a = rng.next_seed("cpu")
        b = rng.next_seed("cpu")
        with rng.scope(None):
            assert rng.counter("cpu") is mine
    assert rng.counter("cpu") is base and int(base.ctr) == before
    assert int(a) == rng._i64(5 + rng._STRIDE) and int(b) == rng._i64(5 + 2 * rng._STRIDE)
    assert a.data_ptr() + 8 == b.data_ptr()                       # neighbours of one table: no launch per draw
    old = [int(rng.next_seed("cpu")) for _ in range(3)]
    with rng.scope(mine):
        for _ in range(rng.TABLE):                                # runs over the table end: refill continues the sequence
            last = rng.next_seed("cpu")
        assert int(a) == rng._i64(5 + rng._STRIDE) and int(last) == rng._i64(5 + (rng.TABLE + 2) * rng._STRIDE)  # the saved view still holds its seed
        mine.seed(5)
        assert int(rng.next_seed("cpu")) == int(a)                # pinned: reproducible
    assert len(set(old)) == 3


def test_model_from_config_lowers_indivisible_heads():
    from mitune.models.transformer import SimpleTransformerModel, TransformerModel
    m = TransformerModel.from_config({"d_model": 48, "num_heads": 32, "num_encoder_layers": 1}, 7)
    assert m.layers[0].attention.h == 16
    s = SimpleTransformerModel.from_config({"d_model": 32, "num_heads": 4, "num_layers": 2}, 7)
    assert len(s.transformer) == 2
