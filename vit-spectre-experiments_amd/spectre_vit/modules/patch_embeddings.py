"""PatchEmbedding -- mirror of reference spectre_vit/modules/patch_embeddings.py:4-43 (Conv2d(k=P,s=P) patcher,
CLS token, learned position embeddings, dropout).  The conv is a per-patch GEMM on the MFMA kernel."""
import torch
import torch.nn as nn

from spectre_vit import hip_ops


class PatchEmbedding(nn.Module):
    def __init__(self, embed_dim, patch_size, num_patches, dropout, in_channels):
        super().__init__()
        self.embed_dim = embed_dim
        self.patcher = nn.Sequential(
            nn.Conv2d(in_channels=in_channels, out_channels=embed_dim, kernel_size=patch_size, stride=patch_size),
            nn.Flatten(2),
        )
        self.cls_token = nn.Parameter(torch.randn(1, 1, embed_dim))
        self.position_embeddings = nn.Parameter(torch.randn(1, num_patches + 1, embed_dim))
        self.dropout = nn.Dropout(p=dropout)
        # extension: uint8 NHWC batches straight from the loader are normalised inside the patch gather
        self.pixel_norm = hip_ops.PixelNorm()

    def forward(self, x):
        conv = self.patcher[0]
        w_full = conv.weight.reshape(self.embed_dim, -1)
        tok = hip_ops.patch_embed(x, w_full, conv.bias, self.cls_token, self.position_embeddings, conv.kernel_size[0],
                                  self.pixel_norm)
        return hip_ops.dropout(tok, self.dropout.p, self.training)
