import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, '/root/repo')
from alphazero_amd.nets import GomokuNet
from alphazero_amd.fused import FusedTowerEvaluator
torch.manual_seed(0)
net = GomokuNet().eval().cuda()
ev = FusedTowerEvaluator(net)
x = torch.randint(-1, 2, (16384, 9, 15, 15), device='cuda').to(torch.bfloat16)
for _ in range(3): ev(x)
torch.cuda.synchronize(); t = time.perf_counter()
n = 20
for _ in range(n): ev(x)
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / n * 1e3
print('fused tower + heads ms', ms, 'TFLOP/s', net.flops_per_eval() * 16384 / ms / 1e9)
