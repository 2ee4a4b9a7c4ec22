"""Baseline ViT -- mirror of reference spectre_vit/models/vit/vit.py:7-51.

The parameter containers are the very same stock modules (nn.TransformerEncoder of nn.TransformerEncoderLayer,
PatchEmbedding, nn.Sequential(nn.Linear)) so the state_dict is key-for-key the reference's; the forward runs on
libspv_hip.so (MFMA GEMMs, HIP attention core, fused residual+LayerNorm).

Reference quirk kept by default: the encoder layers are built with ``batch_first=False`` but fed
``(B, N, E)``, so attention runs ACROSS THE BATCH axis and the CLS row never sees image content.  ``batch_first=True``
(attention across tokens) is this build's own, labelled, extension.
"""
import torch
from torch import nn
from torch.nn import TransformerEncoder, TransformerEncoderLayer

from spectre_vit import hip_ops
from spectre_vit.modules.patch_embeddings import PatchEmbedding


def _encoder_layer_forward(layer: TransformerEncoderLayer, xs: torch.Tensor, training: bool) -> torch.Tensor:
    """post-norm stock layer (norm_first=False) on xs [seqs, len, E]: x = LN1(x + SA(x)); x = LN2(x + W2 gelu(W1 x))."""
    attn = layer.self_attn
    p_attn = attn.dropout if training else 0.0
    qkv = hip_ops.linear(xs, attn.in_proj_weight, attn.in_proj_bias)
    ctx = hip_ops.AttentionFn.apply(qkv, attn.num_heads, p_attn)
    a = hip_ops.linear(ctx, attn.out_proj.weight, attn.out_proj.bias)
    a = hip_ops.dropout(a, layer.dropout1.p, training)
    x1 = hip_ops.add_layernorm(a, xs, layer.norm1.weight, layer.norm1.bias, 1)
    h = hip_ops.GeluFn.apply(hip_ops.linear(x1, layer.linear1.weight, layer.linear1.bias))
    h = hip_ops.dropout(h, layer.dropout.p, training)
    f = hip_ops.linear(h, layer.linear2.weight, layer.linear2.bias)
    f = hip_ops.dropout(f, layer.dropout2.p, training)
    return hip_ops.add_layernorm(f, x1, layer.norm2.weight, layer.norm2.bias, 1)


class ViT(nn.Module):
    def __init__(self, img_size=32, patch_size=4, in_channels=3, num_classes=10, embed_dim=768, num_encoders=12, num_heads=12,
                 hidden_dim=3072, dropout=0.1, activation="gelu", method="attention", batch_first=False):
        super().__init__()
        if activation != "gelu":
            raise NotImplementedError("only the reference's activation='gelu' is built")
        num_patches = (img_size // patch_size) ** 2
        self.embeddings_block = PatchEmbedding(embed_dim, patch_size, num_patches, dropout, in_channels)
        encoder_layer = TransformerEncoderLayer(d_model=embed_dim, nhead=num_heads, dim_feedforward=hidden_dim, dropout=dropout,
                                                activation=activation)
        self.encoder_blocks = TransformerEncoder(encoder_layer, num_layers=num_encoders, enable_nested_tensor=False)
        self.mlp_head = nn.Sequential(nn.Linear(embed_dim, num_classes, 5))  # bias=5 is truthy, as in the reference (vit.py:40)
        self.batch_first = batch_first

    def forward(self, x, return_features=False):
        x = self.embeddings_block(x)  # (B, N, E)
        # batch_first=False (reference behaviour): dim 0 is the sequence axis -> sequences are the N token slots
        xs = x if self.batch_first else x.transpose(0, 1).contiguous()
        for layer in self.encoder_blocks.layers:
            xs = _encoder_layer_forward(layer, xs, self.training)
        x = xs if self.batch_first else xs.transpose(0, 1)
        cls_token = x[:, 0, :]
        head = self.mlp_head[0]
        dt = hip_ops.compute_dtype(cls_token)
        if dt == torch.bfloat16 and (head.in_features % 8 or head.out_features % 8):
            dt = torch.float32
        logits = hip_ops.linear(hip_ops.cast(cls_token.contiguous(), dt), head.weight, head.bias, True)
        if return_features:
            return logits, cls_token
        return logits
