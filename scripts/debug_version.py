import torch
p = [torch.nn.Parameter(torch.randn(10, device="cuda")) for _ in range(3)]
for fused, foreach in ((True, None), (False, True), (False, False)):
    opt = torch.optim.AdamW(p, lr=1e-3, fused=fused, foreach=foreach)
    for q in p: q.grad = torch.randn_like(q)
    v0 = [q._version for q in p]
    opt.step()
    print("fused", fused, "foreach", foreach, "versions", v0, "->", [q._version for q in p])
