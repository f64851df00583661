import torch, time
from transformers.models.llama.modeling_llama import LlamaRMSNorm
dev = torch.device("cuda:0")
x = torch.randn(4608, 4096, device=dev, dtype=torch.bfloat16, requires_grad=True)
m = LlamaRMSNorm(4096, eps=1e-5).to(dev).to(torch.bfloat16)
def hf(): return m(x)
def nat(): return torch.nn.functional.rms_norm(x, (4096,), m.weight, 1e-5)
for name, f in (("hf", hf), ("native", nat)):
    for _ in range(3):
        y = f(); y.float().sum().backward()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        y = f(); y.backward(torch.ones_like(y))
    b.record(); torch.cuda.synchronize()
    print(name, "fwd+bwd us:", a.elapsed_time(b) / 20 * 1000)
y1 = hf(); y2 = nat(); print("max abs diff", (y1.float() - y2.float()).abs().max().item())
with torch.autocast("cuda", dtype=torch.bfloat16):
    y3 = nat(); print("autocast native dtype", y3.dtype)
