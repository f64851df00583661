"""Which scaled_dot_product_attention backend PyTorch-ROCm runs fastest at the generator's shape (B = 18, 32 heads, T = 256,
head 128, bf16, an explicit additive mask as transformers passes for left-padded rows): forward and forward + backward, per
backend.  hipGraph replay timing.    python tools/sdpa_probe.py"""
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (3 * iters)


def main():
    dev = torch.device("cuda:0")
    B, H, T, D = 18, 32, 256, 128
    q, k, v = (torch.randn(B, H, T, D, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    lens = torch.randint(60, T + 1, (B,), device=dev)
    pad = torch.arange(T, device=dev).unsqueeze(0) < (T - lens).unsqueeze(1)               # left padding
    causal = torch.tril(torch.ones(T, T, device=dev, dtype=torch.bool))
    keep = causal.unsqueeze(0) & ~pad.unsqueeze(1)
    mask_bool = keep.unsqueeze(1)                                                             # [B, 1, T, T]
    mask_f = torch.zeros(B, 1, T, T, device=dev, dtype=torch.bfloat16).masked_fill(~mask_bool, float("-inf"))
    mask_f = mask_f.masked_fill(pad.view(B, 1, T, 1), 0.0)      # fully masked (padding) query rows: keep them finite
    up = torch.randn(B, H, T, D, device=dev, dtype=torch.bfloat16)
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
        for mname, kw in (("float mask", dict(attn_mask=mask_f)), ("bool mask", dict(attn_mask=mask_bool)), ("is_causal", dict(is_causal=True))):
            try:
                with sdpa_kernel(be):
                    def fwd():
                        return F.scaled_dot_product_attention(q, k, v, **kw)

                    def both():
                        for t in (q, k, v):
                            t.grad = None
                        F.scaled_dot_product_attention(q, k, v, **kw).backward(up)

                    tf = timed(fwd)
                    tb = timed(both)
                print(f"{name:10s} {mname:11s} fwd {tf:8.1f} us   fwd+bwd {tb:8.1f} us   (bwd alone ~{tb - tf:7.1f} us)", flush=True)
            except Exception as e:
                print(f"{name:10s} {mname:11s} unavailable: {str(e)[:90]}", flush=True)


if __name__ == "__main__":
    main()
