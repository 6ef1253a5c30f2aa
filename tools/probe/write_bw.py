import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
x=torch.empty(2*96_000_000, dtype=torch.float32, device="cuda")   # 768 MB
y=torch.empty_like(x)
print("zero_ 768MB      %.4f ms"%t(lambda: x.zero_()))
print("fill_ 768MB      %.4f ms"%t(lambda: x.fill_(1.5)))
print("copy 768MB->768MB %.4f ms"%t(lambda: y.copy_(x)))
print("read-only sum    %.4f ms"%t(lambda: x.sum()))
h=x[:96_000_000]
print("zero_ 384MB      %.4f ms"%t(lambda: h.zero_()))
