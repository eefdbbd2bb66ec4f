import sys, torch
sys.path.insert(0, '/root/repo')
from alphazero_amd.nets import GomokuNet
from alphazero_amd.fused import FusedTowerEvaluator
torch.manual_seed(0)
net = GomokuNet().eval().cuda()
ev = FusedTowerEvaluator(net)
x = torch.randint(-1, 2, (16384, 9, 15, 15), device='cuda').to(torch.bfloat16)
for _ in range(5): ev(x)
torch.cuda.synchronize()
