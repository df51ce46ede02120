import torch
import torch.nn as nn
import torch.nn.functional as F


class SpatialNorm(nn.Module):  # unused on the CV-VAE path (norm_type="group")
    def __init__(self, *a, **k):
        raise NotImplementedError


class Attention(nn.Module):
    """Restatement of diffusers `Attention` + AttnProcessor2_0 for the only configuration the
    reference builds (vae_blocks3d_sd3.py:806-822): heads=1, dim_head=C, norm_num_groups=32,
    bias=True, residual_connection=True, rescale_output_factor=1, _from_deprecated_attn_block.
    See SURVEY.md Appendix B."""

    def __init__(self, query_dim, heads=8, dim_head=64, rescale_output_factor=1.0, eps=1e-5,
                 norm_num_groups=None, spatial_norm_dim=None, residual_connection=False, bias=False,
                 upcast_softmax=False, _from_deprecated_attn_block=False, **kw):
        super().__init__()
        assert spatial_norm_dim is None
        inner = heads * dim_head
        self.heads = heads
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps, affine=True) if norm_num_groups else None
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kw):
        residual = hidden_states
        b, c, h, w = hidden_states.shape
        x = hidden_states.view(b, c, h * w).transpose(1, 2)
        if self.group_norm is not None:
            x = self.group_norm(x.transpose(1, 2)).transpose(1, 2)
        q, k, v = self.to_q(x), self.to_k(x), self.to_v(x)
        hd = q.shape[-1] // self.heads
        q = q.view(b, -1, self.heads, hd).transpose(1, 2)
        k = k.view(b, -1, self.heads, hd).transpose(1, 2)
        v = v.view(b, -1, self.heads, hd).transpose(1, 2)
        x = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
        x = x.transpose(1, 2).reshape(b, -1, self.heads * hd).to(q.dtype)
        x = self.to_out[1](self.to_out[0](x))
        x = x.transpose(-1, -2).reshape(b, c, h, w)
        if self.residual_connection:
            x = x + residual
        return x / self.rescale_output_factor
