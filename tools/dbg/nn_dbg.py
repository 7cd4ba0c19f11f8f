import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import pn2_amd as pn2
dev = torch.device("cuda:0")
rs = np.random.RandomState(0)
for (b, n, m) in [(1, 8, 64), (1, 64, 128), (2, 100, 1024)]:
    a = torch.from_numpy(rs.rand(b, n, 3).astype(np.float32)).to(dev)
    r = torch.from_numpy(rs.rand(b, m, 3).astype(np.float32)).to(dev)
    d, i = pn2.three_nn(a, r)
    dd = ((a.double()[:, :, None] - r.double()[:, None]) ** 2).sum(-1)
    rd, ri = torch.topk(dd, 3, dim=2, largest=False)
    print((b, n, m), "idx equal:", bool((ri.int() == i).all()), "max |d - ref|:", float((d.double() - rd).abs().max()))
    if not (ri.int() == i).all():
        print(" got", i[0, :3].tolist(), "ref", ri[0, :3].tolist())
        print(" got d", d[0, :3].tolist(), "ref", rd[0, :3].tolist())
