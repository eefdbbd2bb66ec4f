import sys, time, torch
sys.path.insert(0, '.')
from alphazero_amd import nets
import torch.nn.functional as F
torch.manual_seed(0)
dev = 'cuda'
m = nets.GomokuNet().eval().to(dev)
# randomize BN stats so folding is non-trivial
for mod in m.modules():
    if isinstance(mod, torch.nn.BatchNorm2d):
        mod.running_mean.normal_(0, 0.2); mod.running_var.uniform_(0.5, 1.5); mod.weight.data.uniform_(0.5,1.5); mod.bias.data.normal_(0,0.2)
x32 = (torch.randint(-1, 2, (16384, 9, 15, 15), device=dev)).float()
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
with torch.inference_mode():
    ref_p, ref_v = m(x32)
    mb = nets.GomokuNet().eval().to(dev); mb.load_state_dict(m.state_dict()); mb = mb.bfloat16().to(memory_format=torch.channels_last)
    xb = x32.bfloat16().contiguous(memory_format=torch.channels_last)
    print('baseline bf16 cl ms', bench(lambda: mb(xb)))
    def fold(conv, bn):
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = conv.weight * s.view(-1,1,1,1)
        b = bn.bias - bn.running_mean * s
        return w, b
    sl = m._shared_layers
    W = []
    W.append(fold(sl[0], sl[1]))
    for blk in (sl[3], sl[4]):
        W.append(fold(blk._conv1, blk._bn1)); W.append(fold(blk._conv2, blk._bn2))
    Wp = fold(m._policy_head[0], m._policy_head[1]); Wv = fold(m._value_head[0], m._value_head[1])
    def cast(t): return tuple(a.bfloat16() for a in t)
    Wb = [ (w.bfloat16().contiguous(memory_format=torch.channels_last), b.bfloat16()) for w,b in W]
    Wpb = (Wp[0].bfloat16().contiguous(memory_format=torch.channels_last), Wp[1].bfloat16()); Wvb = (Wv[0].bfloat16().contiguous(memory_format=torch.channels_last), Wv[1].bfloat16())
    pl = m._policy_head[4]; v1 = m._value_head[4]; v2 = m._value_head[6]
    plw, plb = pl.weight.bfloat16(), pl.bias.bfloat16(); v1w, v1b = v1.weight.bfloat16(), v1.bias.bfloat16(); v2w, v2b = v2.weight.bfloat16(), v2.bias.bfloat16()
    def heads(h, conv):
        p = F.relu(conv(h, *Wpb))  # (B,1,15,15)
        p = F.linear(p.flatten(1), plw, plb)
        v = F.relu(conv(h, *Wvb))
        # value head flatten must be NCHW order: (B,2,15,15) channels_last -> permute
        v = v.contiguous().flatten(1)
        v = torch.tanh(F.linear(F.relu(F.linear(v, v1w, v1b)), v2w, v2b))
        return p, v
    def fwd_folded(x):
        h = F.relu_(F.conv2d(x, *Wb[0], padding=1))
        for i in (1,3):
            y = F.relu_(F.conv2d(h, *Wb[i], padding=1))
            y = F.conv2d(y, *Wb[i+1], padding=1)
            h = F.relu_(y.add_(h))
        return heads(h, lambda a,w,b: F.conv2d(a,w,b))
    p,v = fwd_folded(xb)
    print('folded err', (p.float()-ref_p).abs().max().item(), (v.float()-ref_v).abs().max().item(), 'ref scale', ref_p.abs().max().item())
    print('folded conv2d+relu ms', bench(lambda: fwd_folded(xb)))
    cr = torch.ops.aten.miopen_convolution_relu
    car = torch.ops.aten.miopen_convolution_add_relu
    def fwd_fused(x):
        h = cr(x, Wb[0][0], Wb[0][1], [1,1],[1,1],[1,1],1)
        for i in (1,3):
            y = cr(h, Wb[i][0], Wb[i][1], [1,1],[1,1],[1,1],1)
            h = car(y, Wb[i+1][0], h, 1.0, Wb[i+1][1], [1,1],[1,1],[1,1],1)
        return heads(h, lambda a,w,b: cr(a,w,b,[1,1],[0,0],[1,1],1) )
    try:
        p,v = fwd_fused(xb)
        print('fused err', (p.float()-ref_p).abs().max().item(), (v.float()-ref_v).abs().max().item())
        print('miopen fused ms', bench(lambda: fwd_fused(xb)))
    except Exception as e:
        print('fused failed', repr(e)[:300])
    # fp16 too
    print('done')
