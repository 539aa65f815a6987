"""Transformer-encoder regressors (sequence [B, S, F] -> scalar per sequence).

Capability parity with the reference's model zoo, re-implemented so that every sampled
hyper-parameter is live and every combination runs:

* ``PositionalEncoding``      - sinusoidal table added to the sequence, then dropout
  (/root/reference/ray-tune-hpo-regression.py:25-54; that version indexes its 2-D buffer with three
  indices and breaks when ``max_seq_length`` < S; here the table grows on demand).
* ``DepthwiseSeparableConv``  - depthwise conv over time + pointwise expansion as the FFN
  alternative (:59-82; that version returns ``dim_feedforward`` channels into a ``d_model``
  residual; here a second pointwise conv projects back).
* ``LinearAttention``         - ``relu(Q K^T) V`` without softmax (:87-117), here with a head split,
  row normalisation, ``key_dim_scaling`` and an optional local depthwise mixing of width
  ``kernel_size``; returns ``(out, None)`` like ``nn.MultiheadAttention`` so callers need no branch.
* ``SelfAttention``           - multi-head softmax attention (the reference's "scaled_dot_product"
  and "multi_head_attention" options are both ``nn.MultiheadAttention``, :138-143).
* ``CustomEncoderLayer``      - post-norm block ``x = LN(x + drop(attn(x)))``, ``x = LN(x + drop(ffn(x)))``
  (:122-178) plus stochastic depth.
* ``TransformerModel``        - projection -> PE -> L layers -> last token -> ReLU MLP head
  ``d -> 128 -> 64 -> 32 -> 16 -> 1`` (:183-240), accepting the ``key_dim_scaling`` /
  ``shared_weights`` / ``stochastic_depth_rate`` arguments its training function passes (:279-283).
* ``SimpleTransformerModel``  - the sample script's model: projection -> stock post-norm encoder
  (dff = 4 d) -> last token -> ``Linear(d, 1)`` (/root/reference/ray-tune-hpo-regression-sample.py:60-83).

Attention, dropout+residual+LayerNorm, PE add + dropout, the depthwise convolution and every dense projection
(``ops/linear.py``: exact-f32 or bf16 MFMA tile kernels, no BLAS-library GEMM) go through ``mitune.ops``, which runs
hand-written HIP kernels on a GPU and plain PyTorch on the CPU.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops.attention import self_attention, self_attention_qkv
from ..ops.elementwise import depthwise_conv1d, pe_add_dropout, relu_dropout
from ..ops.layernorm import residual_layer_norm
from ..ops.linear import linear

ATTENTION_TYPES = ("scaled_dot_product", "multi_head_attention", "linear_attention")


def sinusoid_table(length: int, d_model: int) -> torch.Tensor:
    pos = torch.arange(length, dtype=torch.float32).unsqueeze(1)
    freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    table = torch.zeros(length, d_model)
    table[:, 0::2] = torch.sin(pos * freq)
    table[:, 1::2] = torch.cos(pos * freq[: d_model // 2])
    return table


class PositionalEncoding(nn.Module):
    def __init__(self, d_model: int, max_seq_length: int = 512, dropout: float = 0.1):
        super().__init__()
        self.d_model = d_model
        self.drop = nn.Dropout(dropout)
        self.register_buffer("pe", sinusoid_table(max_seq_length, d_model), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        s = x.size(1)
        if s > self.pe.size(0):      # a table sized by a hyper-parameter must never bound the data
            self.pe = sinusoid_table(s, self.d_model).to(self.pe.device)
        return pe_add_dropout(x, self.pe, self.drop.p, self.training)


class DepthwiseSeparableConv(nn.Module):
    """FFN alternative: per-channel temporal conv, pointwise expand, ReLU, pointwise project back."""

    def __init__(self, d_model: int, dim_feedforward: int, kernel_size: int = 3):
        super().__init__()
        if kernel_size % 2 == 0:
            raise ValueError("kernel_size must be odd")
        self.depthwise = nn.Conv1d(d_model, d_model, kernel_size, padding=kernel_size // 2, groups=d_model)
        self.pointwise = nn.Conv1d(d_model, dim_feedforward, 1)
        self.project = nn.Conv1d(dim_feedforward, d_model, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:          # [B, S, d], channel-last throughout
        y = depthwise_conv1d(x, self.depthwise.weight, self.depthwise.bias)
        y = F.relu(linear(y, self.pointwise.weight.squeeze(-1), self.pointwise.bias))     # k=1 conv == linear
        return linear(y, self.project.weight.squeeze(-1), self.project.bias)


class SelfAttention(nn.Module):
    """Multi-head softmax self-attention; ``key_dim_scaling`` shrinks the query/key width per head."""

    kind = "softmax"

    def __init__(self, d_model: int, num_heads: int, dropout: float = 0.0, key_dim_scaling: float = 1.0):
        super().__init__()
        if d_model % num_heads:
            raise ValueError(f"d_model {d_model} is not divisible by num_heads {num_heads}")
        self.h = num_heads
        self.dv = d_model // num_heads
        self.dk = max(1, int(round(self.dv * key_dim_scaling)))
        # one projection for queries, keys and values: [all q | all k | all v], each block head-major
        self.qkv_proj = nn.Linear(d_model, 2 * self.h * self.dk + d_model)
        self.out_proj = nn.Linear(d_model, d_model)
        self.p = dropout

    mixes_values = False

    def _mix(self, v: torch.Tensor) -> torch.Tensor:
        return v

    def forward(self, query: torch.Tensor, key: Optional[torch.Tensor] = None, value: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, None]:
        x = query                                  # self-attention: key/value arguments are accepted, not used
        qkv = linear(x, self.qkv_proj.weight, self.qkv_proj.bias)        # [B, S, 2*h*dk + d_model]
        p = self.p if self.training else 0.0
        if self.mixes_values:
            qk, v = qkv.split([2 * self.h * self.dk, self.h * self.dv], dim=-1)
            o = self_attention(qk, self._mix(v), self.h, kind=self.kind, dropout_p=p)
        else:
            o = self_attention_qkv(qkv, self.h, self.dk, kind=self.kind, dropout_p=p)
        return linear(o, self.out_proj.weight, self.out_proj.bias), None


class LinearAttention(SelfAttention):
    """Softmax-free attention ``normalise(relu(Q K^T / sqrt(dk))) V`` with optional local value mixing."""

    kind = "relu"

    def __init__(self, d_model: int, num_heads: int, kernel_size: int = 3, dropout: float = 0.0,
                 key_dim_scaling: float = 1.0):
        super().__init__(d_model, num_heads, dropout, key_dim_scaling)
        self.kernel_size = kernel_size
        self.local = (nn.Conv1d(d_model, d_model, kernel_size, padding=kernel_size // 2, groups=d_model)
                      if kernel_size and kernel_size > 1 else None)
        self.mixes_values = self.local is not None

    def _mix(self, v: torch.Tensor) -> torch.Tensor:
        if self.local is None:
            return v
        return v + depthwise_conv1d(v, self.local.weight, self.local.bias)


def build_attention(attention_type: str, d_model: int, num_heads: int, dropout: float, attn_kernel_size: int,
                    key_dim_scaling: float) -> nn.Module:
    if attention_type in ("scaled_dot_product", "multi_head_attention"):
        return SelfAttention(d_model, num_heads, dropout, key_dim_scaling)
    if attention_type == "linear_attention":
        return LinearAttention(d_model, num_heads, attn_kernel_size, dropout, key_dim_scaling)
    raise ValueError(f"attention_type must be one of {ATTENTION_TYPES}, got {attention_type!r}")


class CustomEncoderLayer(nn.Module):
    def __init__(self, d_model: int, num_heads: int, dim_feedforward: int, dropout: float,
                 attention_type: str = "scaled_dot_product", attn_kernel_size: int = 3,
                 depthwise_separable_conv: bool = False, key_dim_scaling: float = 1.0, drop_path: float = 0.0):
        super().__init__()
        self.attention = build_attention(attention_type, d_model, num_heads, dropout, attn_kernel_size,
                                         key_dim_scaling)
        if depthwise_separable_conv:
            self.feedforward = DepthwiseSeparableConv(d_model, dim_feedforward, kernel_size=3)
        else:
            self.feedforward = nn.Sequential(nn.Linear(d_model, dim_feedforward), nn.ReLU(),
                                             nn.Linear(dim_feedforward, d_model))
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout
        self.drop_path = drop_path

    def _branch_scale(self, x: torch.Tensor, rate: float) -> Optional[torch.Tensor]:
        """Stochastic depth: per-sample keep mask scaled by 1/keep (None when inactive)."""
        if not self.training or rate <= 0.0:
            return None
        keep = 1.0 - rate
        from ..ops.elementwise import keep_mask
        return keep_mask((x.size(0), 1, 1), rate, x.device).to(x.dtype) / keep

    def forward(self, src: torch.Tensor, drop_path: Optional[float] = None) -> torch.Tensor:
        rate = self.drop_path if drop_path is None else drop_path
        a, _ = self.attention(src, src, src)
        m = self._branch_scale(src, rate)
        if m is not None:
            a = a * m
        src = residual_layer_norm(a, src, self.norm1.weight, self.norm1.bias, self.norm1.eps,
                                  dropout_p=self.p if self.training else 0.0)
        ff = self.feedforward
        if isinstance(ff, nn.Sequential):          # same parameters, dense layers through the fused-backward path
            f = linear(F.relu(linear(src, ff[0].weight, ff[0].bias)), ff[2].weight, ff[2].bias)
        else:
            f = ff(src)
        m = self._branch_scale(src, rate)
        if m is not None:
            f = f * m
        return residual_layer_norm(f, src, self.norm2.weight, self.norm2.bias, self.norm2.eps,
                                   dropout_p=self.p if self.training else 0.0)


class TransformerModel(nn.Module):
    def __init__(self, input_size: int, num_heads: int, num_encoder_layers: int, d_model: int, dim_feedforward: int,
                 dropout: float, max_seq_length: int = 512, attention_type: str = "scaled_dot_product",
                 attn_kernel_size: int = 3, depthwise_separable_conv: bool = False, key_dim_scaling: float = 1.0,
                 shared_weights: bool = False, stochastic_depth_rate: float = 0.0):
        super().__init__()
        self.input_projection = nn.Linear(input_size, d_model)
        self.positional_encoding = PositionalEncoding(d_model, max_seq_length, dropout)
        self.num_encoder_layers = num_encoder_layers
        self.shared_weights = shared_weights
        n_unique = 1 if shared_weights else num_encoder_layers
        self.layers = nn.ModuleList(
            CustomEncoderLayer(d_model, num_heads, dim_feedforward, dropout, attention_type, attn_kernel_size,
                               depthwise_separable_conv, key_dim_scaling) for _ in range(n_unique))
        # drop-path rate grows linearly with depth, 0 for the first layer
        self.drop_rates = [stochastic_depth_rate * i / max(1, num_encoder_layers - 1)
                           for i in range(num_encoder_layers)]
        self.fc1 = nn.Linear(d_model, 128)
        self.fc2 = nn.Linear(128, 64)
        self.fc3 = nn.Linear(64, 32)
        self.fc4 = nn.Linear(32, 16)
        self.fc5 = nn.Linear(16, 1)

    @classmethod
    def from_config(cls, config, input_size: int) -> "TransformerModel":
        """Model of one trial of the full search space (/root/reference/ray-tune-hpo-regression.py:266-284); a head
        count that does not divide ``d_model`` is lowered to the largest divisor among the sampled values."""
        d_model = int(config["d_model"])
        heads = int(config.get("num_heads", 4))
        if d_model % heads:
            heads = max(h for h in (1, 2, 4, 8, 16, 32) if d_model % h == 0 and h <= heads)
        return cls(input_size=input_size, num_heads=heads,
                   num_encoder_layers=int(config.get("num_encoder_layers", config.get("num_layers", 2))),
                   d_model=d_model, dim_feedforward=int(config.get("dim_feedforward", 4 * d_model)),
                   dropout=float(config.get("dropout", 0.1)), max_seq_length=int(config.get("max_seq_length", 512)),
                   attention_type=config.get("attention_type", "scaled_dot_product"),
                   attn_kernel_size=int(config.get("attn_kernel_size", 3)),
                   depthwise_separable_conv=bool(config.get("depthwise_separable_conv", False)),
                   key_dim_scaling=float(config.get("key_dim_scaling", 1.0)),
                   shared_weights=bool(config.get("shared_weights", False)),
                   stochastic_depth_rate=float(config.get("stochastic_depth_rate", 0.0)))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.positional_encoding(linear(x, self.input_projection.weight, self.input_projection.bias))
        for i in range(self.num_encoder_layers):
            layer = self.layers[0] if self.shared_weights else self.layers[i]
            x = layer(x, drop_path=self.drop_rates[i])
        fcs = (self.fc1, self.fc2, self.fc3, self.fc4, self.fc5)
        h = x[:, -1, :]                                  # last token of every sequence
        # Layer by layer: with the one-launch dense backward of ``ops/linear.py`` this is 90-200 us forward + backward at
        # batch 32-256; a single-kernel head with all five weight matrices staged through LDS was built and measured at
        # 160-590 us (its workgroups stream the whole first matrix for eight rows each) and removed.
        for i, fc in enumerate(fcs):
            h = linear(h, fc.weight, fc.bias)
            if i < len(fcs) - 1:
                h = F.relu(h)
        return h


class _PostNormLayer(nn.Module):
    """Stock post-norm encoder layer (ReLU FFN, dropout in attention, after attention, inside and after FFN)."""

    def __init__(self, d_model: int, num_heads: int, dim_feedforward: int, dropout: float):
        super().__init__()
        self.self_attn = SelfAttention(d_model, num_heads, dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        p = self.p if self.training else 0.0
        a, _ = self.self_attn(x)
        x = residual_layer_norm(a, x, self.norm1.weight, self.norm1.bias, self.norm1.eps, dropout_p=p)
        f = linear(relu_dropout(linear(x, self.linear1.weight, self.linear1.bias), p, self.training),
                   self.linear2.weight, self.linear2.bias)
        return residual_layer_norm(f, x, self.norm2.weight, self.norm2.bias, self.norm2.eps, dropout_p=p)


class SimpleTransformerModel(nn.Module):
    def __init__(self, input_size: int, d_model: int, num_heads: int, num_layers: int, dropout: float = 0.1):
        super().__init__()
        self.input_projection = nn.Linear(input_size, d_model)
        self.transformer = nn.ModuleList(_PostNormLayer(d_model, num_heads, 4 * d_model, dropout)
                                         for _ in range(num_layers))
        self.fc = nn.Linear(d_model, 1)

    @classmethod
    def from_config(cls, config, input_size: int) -> "SimpleTransformerModel":
        """Model of one trial of the sample search space (/root/reference/ray-tune-hpo-regression-sample.py:94-99)."""
        return cls(input_size, int(config["d_model"]), int(config["num_heads"]), int(config["num_layers"]),
                   float(config.get("dropout", 0.1)))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = linear(x, self.input_projection.weight, self.input_projection.bias)
        for layer in self.transformer:
            x = layer(x)
        return linear(x[:, -1, :], self.fc.weight, self.fc.bias).squeeze(-1)
