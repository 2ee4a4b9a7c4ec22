import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-spectre-experiments_amd"))
from spectre_vit import hip_ops as ops
dev = torch.device("cuda:0")
M, N, K = 33280, 768, 512
g = torch.Generator(device="cpu").manual_seed(1)
A = (torch.randn((M, K), generator=g) * 0.5).to(dev).to(torch.bfloat16)
B = (torch.randn((N, K), generator=g) * 0.1).to(dev).to(torch.bfloat16)
C0 = torch.randn((M, N), generator=g).to(dev).to(torch.bfloat16)
C2 = C0.clone()
ops._gemm(A, B, None, C2, M, N, K, K, K, N, 1, 1, None)
C32 = torch.empty((M, N), dtype=torch.float32, device=dev)
ops._gemm(A, B, None, C32, M, N, K, K, K, N, 0, 1, None)
ref = (C32 + C0.float()).to(torch.bfloat16)
bad = (C2 != ref)
print("mismatches", int(bad.sum()), "of", M * N)
idx = bad.nonzero()
if len(idx):
    r, c = idx[:, 0], idx[:, 1]
    print("rows mod 32 histogram", torch.bincount(r % 32, minlength=32).tolist())
    print("row // 32 (first 20 distinct)", torch.unique(r // 32)[:20].tolist())
    print("cols // 32 histogram", torch.bincount(c // 32, minlength=N // 32).tolist())
    print("first", idx[:5].tolist(), "got", C2[r[0], c[0]].item(), "ref", ref[r[0], c[0]].item(), "plain", C32[r[0], c[0]].item(), "old", C0[r[0], c[0]].item())
    # does the wrong value equal acc + a different old value?
    d = (C2.float() - C32)[bad]
    print("got - acc (first 8)", d[:8].tolist(), "old there", C0[bad][:8].float().tolist())
