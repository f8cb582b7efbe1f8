"""Diagnostics: fused mapping loss / fused Adam vs the same maths as chains of torch ops (what the reference runs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import loss_oracle   # test-infrastructure restatement of the reference's torch-op chain (device-agnostic)
from gs_icp_slam_amd.loss import mapper_loss
from gs_icp_slam_amd.optim import FusedAdam
dev = "cuda"
H, W, P = 680, 1200, 300_000
torch.manual_seed(0)
gt = torch.rand(3, H, W, device=dev); gtd = torch.rand(1, H, W, device=dev) + 1
img = (gt + 0.05 * torch.randn_like(gt)).requires_grad_(True); dep = (gtd + 0.02 * torch.randn_like(gtd)).requires_grad_(True)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
def torch_loss():
    img.grad = None; dep.grad = None
    loss_oracle.mapper_loss(img, dep, gt, gtd)[0].backward()
def fused_loss():
    img.grad = None; dep.grad = None
    mapper_loss(img, dep, gt, gtd).backward()
print(f"loss fwd+bwd  torch-op chain {timeit(torch_loss):8.1f} us   fused {timeit(fused_loss):8.1f} us")
shapes = [(P, 3), (P, 1, 3), (P, 1), (P, 3), (P, 4)]
ps = [torch.randn(s, device=dev, requires_grad=True) for s in shapes]
for p in ps: p.grad = torch.randn_like(p)
ref = torch.optim.Adam([{"params": [p], "lr": 1e-3} for p in ps], lr=0.0, eps=1e-15)
fus = FusedAdam([{"params": [p], "lr": 1e-3} for p in ps], lr=0.0, eps=1e-15)
print(f"Adam step     torch.optim.Adam {timeit(ref.step):8.1f} us   fused {timeit(fus.step):8.1f} us")
