import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from neuralnj_amd import train_ops as T
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for nb, M, N, K in ((8, 50, 50, 65536), (8, 1, 49, 65536), (3, 26, 26, 65536), (8, 50, 50, 65536 + 1000)):
    a = torch.randn(nb, M, K, generator=g).to(d)
    b = torch.randn(nb, N, K, generator=g).to(d)
    got = T.Bmm.apply(a, b, True, 0.37)
    want = 0.37 * torch.bmm(a.double(), b.double().transpose(1, 2))
    err = float((got.double() - want).abs().max() / want.abs().max())
    print(nb, M, N, K, "rel err", err)
