"""FT-Transformer-style tabular regressor (feature tokenizer + Transformer encoder + [CLS] head).

Not part of the reference (its models are sequence encoders, /root/reference/ray-tune-hpo-regression.py:183-240);
it is the model family of the "tabular regression in bf16 with population-based training" target
configuration.  Every numeric feature becomes one token (``x_j * w_j + b_j``), a learned [CLS] token
is appended, ``n_blocks`` post-norm encoder blocks with a ReGLU feed-forward mix the tokens, and the
[CLS] representation is regressed to a scalar.  With F <= 127 features the token sequence fits the fused
short-sequence attention kernel (S <= 128); tokenizer, projections, LayerNorm and ReGLU run on this package's own
kernels, under bf16 autocast on a GPU.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn as nn

from .transformer import SelfAttention
from ..ops.attention import self_attention_qkv
from ..ops.elementwise import reglu_dropout
from ..ops.layernorm import residual_layer_norm_mixed
from ..ops.linear import linear
from ..ops.tokens import cls_head, feature_tokens


class FeatureTokenizer(nn.Module):
    def __init__(self, n_features: int, d_token: int):
        super().__init__()
        bound = 1.0 / math.sqrt(d_token)
        self.weight = nn.Parameter(torch.empty(n_features, d_token).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(n_features, d_token).uniform_(-bound, bound))
        self.cls = nn.Parameter(torch.empty(1, 1, d_token).uniform_(-bound, bound))

    def forward(self, x: torch.Tensor, want_bf16: bool = False):   # [B, F] -> ([B, F + 1, d] f32, its bf16 copy or None)
        return feature_tokens(x, self.weight, self.bias, self.cls, want_bf16)


class ReGLU(nn.Module):
    def __init__(self, d_token: int, d_hidden: int, dropout: float):
        super().__init__()
        self.up = nn.Linear(d_token, 2 * d_hidden)
        self.down = nn.Linear(d_hidden, d_token)
        self.p = dropout

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        u = linear(x, self.up.weight, self.up.bias)                     # [a | g]
        return linear(reglu_dropout(u, self.p, self.training), self.down.weight, self.down.bias)


class FTBlock(nn.Module):
    def __init__(self, d_token: int, n_heads: int, d_hidden: int, attn_dropout: float, ffn_dropout: float,
                 residual_dropout: float):
        super().__init__()
        self.attn = SelfAttention(d_token, n_heads, attn_dropout)
        self.ffn = ReGLU(d_token, d_hidden, ffn_dropout)
        self.norm1, self.norm2 = nn.LayerNorm(d_token), nn.LayerNorm(d_token)
        self.p = residual_dropout

    def forward(self, x: torch.Tensor, x16=None, last: bool = False):
        """``x``: the f32 residual stream; ``x16``: its bf16 copy when the caller has one (under bf16 autocast the GEMMs
        read that, and the LayerNorm kernels take the GEMMs' bf16 outputs as they are: no cast kernels in between).
        Returns the new stream and its bf16 copy (``None`` outside autocast).

        ``last``: only the summary token of the last block's output is ever read, so past the attention (which needs
        every token as key and value) the block carries that one row per sample: output projection, both LayerNorms and
        the feed-forward run on ``[B, d]`` instead of ``[B, F + 1, d]``.  Returns ``([B, d], None)``."""
        p = self.p if self.training else 0.0
        amp = torch.is_autocast_enabled() and x.is_cuda
        src = x16 if x16 is not None else x
        if last:
            at = self.attn
            qkv = linear(src, at.qkv_proj.weight, at.qkv_proj.bias)
            o = self_attention_qkv(qkv, at.h, at.dk, kind=at.kind, dropout_p=at.p if self.training else 0.0)
            a = linear(o[:, -1, :], at.out_proj.weight, at.out_proj.bias)
            y, y16 = residual_layer_norm_mixed(a, x[:, -1, :].float(), self.norm1.weight, self.norm1.bias, self.norm1.eps,
                                               dropout_p=p, want_bf16=amp)
            f = self.ffn(y16 if y16 is not None else y)
            return residual_layer_norm_mixed(f, y, self.norm2.weight, self.norm2.bias, self.norm2.eps, dropout_p=p,
                                             want_bf16=False)
        a, _ = self.attn(src)
        x, x16 = residual_layer_norm_mixed(a, x.float(), self.norm1.weight, self.norm1.bias, self.norm1.eps, dropout_p=p,
                                           want_bf16=amp)
        f = self.ffn(x16 if x16 is not None else x)
        return residual_layer_norm_mixed(f, x, self.norm2.weight, self.norm2.bias, self.norm2.eps, dropout_p=p,
                                         want_bf16=amp)


class FTTransformer(nn.Module):
    def __init__(self, n_features: int, d_token: int = 192, n_blocks: int = 3, n_heads: int = 8,
                 ffn_factor: float = 4.0 / 3.0, attn_dropout: float = 0.2, ffn_dropout: float = 0.1,
                 residual_dropout: float = 0.0):
        super().__init__()
        if d_token % n_heads:
            raise ValueError("d_token must be divisible by n_heads")
        self.tokenizer = FeatureTokenizer(n_features, d_token)
        d_hidden = max(8, int(round(d_token * ffn_factor / 8)) * 8)
        self.blocks = nn.ModuleList(FTBlock(d_token, n_heads, d_hidden, attn_dropout, ffn_dropout, residual_dropout)
                                    for _ in range(n_blocks))
        self.head_norm = nn.LayerNorm(d_token)
        self.head = nn.Linear(d_token, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:            # [B, F] -> [B]
        t, t16 = self.tokenizer(x, want_bf16=torch.is_autocast_enabled() and x.is_cuda)
        for i, blk in enumerate(self.blocks):
            t, t16 = blk(t, t16, last=i == len(self.blocks) - 1)
        cls = t if t.dim() == 2 else t[:, -1, :]          # the last block hands back the summary token only
        return cls_head(cls.float(), self.head_norm.weight, self.head_norm.bias, self.head.weight, self.head.bias,
                        self.head_norm.eps)

    @classmethod
    def from_config(cls, config: Dict, n_features: int) -> "FTTransformer":
        return cls(n_features, d_token=int(config.get("d_token", 192)), n_blocks=int(config.get("n_blocks", 3)),
                   n_heads=int(config.get("n_heads", 8)), ffn_factor=float(config.get("ffn_factor", 4.0 / 3.0)),
                   attn_dropout=float(config.get("attention_dropout", 0.2)),
                   ffn_dropout=float(config.get("ffn_dropout", 0.1)),
                   residual_dropout=float(config.get("residual_dropout", 0.0)))
