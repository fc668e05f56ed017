import torch, sys
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for k in a:
    va, vb = a[k], b[k]
    if not torch.is_tensor(va):
        continue
    d = (va.double() - vb.double()).abs().max().item()
    s = vb.double().abs().max().item()
    flag = "  <<<<" if d > 1e-4 * max(s, 1e-12) else ""
    print(f"{k:40s} maxdiff {d:9.2e} scale {s:9.2e}{flag}")
