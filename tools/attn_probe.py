"""What torch's memory-efficient SDPA op returns on this build (shape / meaning of its log-sum-exp), for dalm_attn_bwd."""
import torch

dev = torch.device("cuda:0")
B, H, T, hd = 2, 4, 256, 128
g = torch.Generator().manual_seed(0)
q, k, v = [torch.randn(B, T, H, hd, generator=g).bfloat16().to(dev).transpose(1, 2) for _ in range(3)]
start = torch.tensor([0, 37], device=dev)
col = torch.arange(T, device=dev)
mask = (col[None, None, :] <= col[None, :, None]) & (col[None, None, :] >= start[:, None, None])     # [B, T, T]
mask = mask[:, None]                                                                                  # [B, 1, T, T]
bias = torch.zeros(B, 1, T, T, device=dev, dtype=torch.bfloat16).masked_fill(~mask, float("-inf"))
scale = hd ** -0.5
for name, b in (("bias [B,1,T,T]", bias), ("bias expanded [B,H,T,T]", bias.expand(B, H, T, T))):
    try:
        out, lse, seed, off = torch.ops.aten._scaled_dot_product_efficient_attention(q, k, v, b, True, 0.0, False, scale=scale)
        print(name, "-> out", tuple(out.shape), out.stride(), "lse", tuple(lse.shape), lse.dtype, lse.stride())
        s = (q.float() @ k.float().transpose(-1, -2)) * scale + bias.float()
        ref = torch.logsumexp(s, -1)
        ok = torch.isfinite(ref)
        print("   lse vs natural-log reference on rows with a live key: max abs diff", float((lse[..., :T][ok] - ref[ok]).abs().max()),
              "  rows without a live key:", lse[..., :T][~ok][:4].tolist())
        ref_out = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask, scale=scale)
        print("   out vs F.sdpa(bool mask): max abs diff", float((out.float() - ref_out.float())[ok].abs().max()), "F.sdpa out strides", ref_out.stride())
    except Exception as e:
        print(name, "FAILED:", repr(e)[:300])
# what F.sdpa runs with a bool mask (kernel names) and grads of fully masked rows
qq = q.detach().clone().requires_grad_(True)
o = torch.nn.functional.scaled_dot_product_attention(qq, k, v, attn_mask=mask, scale=scale)
print("F.sdpa rows without a live key: out", o[1, 0, :2, :3].tolist())
o.float().sum().backward()
print("grad finite:", bool(torch.isfinite(qq.grad).all()), " grad_fn:", type(o.grad_fn).__name__)
