"""Torch restatement of the two scene encoders over the product's parameter containers (ramp_amd/scene_encoder.py holds the
parameters only; the product computes the latents in HIP).  Test infrastructure: a CPU cross-check that the containers'
parameter names / shapes mean what the reference's do (obstacle_encoder.py:94-152, obstacle_encoder3d.py:55-94), against the
latents captured from the reference (tests/golden/scene_latents.npz)."""
import torch
import torch.nn.functional as F


def _pe(pos, v):
    out = torch.zeros(*v.shape[:-1], pos.d_model, device=v.device, dtype=v.dtype)
    out[..., 0::2] = torch.sin(v[..., 0, None] * pos.div_term) + torch.sin(v[..., 1, None] * pos.div_term)
    out[..., 1::2] = torch.cos(v[..., 0, None] * pos.div_term) + torch.cos(v[..., 1, None] * pos.div_term)
    return out


def _set_attention(a, x):
    B, N, C = x.shape
    q, k, v = a.qkv(x).reshape(B, N, 3, a.num_heads, a.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
    w = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) * a.scale, dim=-1)
    return a.proj(torch.matmul(w, v).transpose(1, 2).reshape(B, N, C))


def _block2d(b, x):
    x = x + _set_attention(b.attn, b.norm1(x))
    return x + b.mlp(b.norm2(x))


def encode_2d(enc, x):
    """ObstacleEncoderSet: cloud (b, No, Np, 2) -> (b, 320)."""
    b, no, npnt, _ = x.shape
    centres = x.mean(dim=2)
    relp = x - centres.unsqueeze(2)
    maxd, _ = torch.max(torch.abs(relp).view(b, no, -1), dim=-1, keepdim=True)
    pe_obs, pe_rel = _pe(enc.pos_encoder, centres), _pe(enc.pos_encoder, relp / (maxd.unsqueeze(-1) + 1e-8))
    emb = enc.point_embedding(x.reshape(b * no * npnt, -1)).view(b, no, npnt, -1)
    comb = torch.cat([emb, pe_obs.unsqueeze(2).expand(-1, -1, npnt, -1), pe_rel], dim=-1)
    comb = enc.combined_encoder(comb).view(b, no * npnt, -1)
    outs = []
    for tr, pool in zip(enc.set_transformers, enc.poolings):
        h = comb
        for blk in tr:
            h = _block2d(blk, h)
        outs.append(pool(h.mean(dim=1)))
    return torch.cat(outs, dim=-1)


def encode_3d(enc, pts):
    """ObstacleEncoder (eval mode): cloud (b, No, Np, 3) -> (b, 256)."""
    b, no, npnt, d = pts.shape
    pp = enc.point_processor
    x = pts.reshape(-1, npnt, d).transpose(2, 1)
    x = F.selu(pp.bn1(pp.conv1(x)))
    x = F.selu(pp.bn2(pp.conv2(x)))
    x = torch.max(x, 2)[0].view(b, no, enc.embedding_dim)
    for blk in enc.set_transformer_blocks:
        xn = blk.norm1(x).transpose(0, 1)
        a, _ = blk.mha(xn, xn, xn)
        x = x + a.transpose(0, 1)
        x = x + blk.ffn(blk.norm2(x))
    feat = enc.output_proj(x)
    return enc.global_pooling(torch.max(feat, dim=1)[0])
