"""How much of a self-play move is not the 400 simulation steps? (GPU time by events, one GPU)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from alphazero_amd import Engine
from alphazero_amd.arena import SelfPlayRunner
from alphazero_amd.nets import GomokuNet

torch.manual_seed(0)
dev = torch.device("cuda", 0)
sims, slots = 400, 16384
model = GomokuNet(15, 64, 2).eval().to(dev).to(torch.bfloat16)
eng = Engine(2, slots, sims, board_size=15, nn_dtype=torch.bfloat16, device=dev)
run = SelfPlayRunner(eng, model, sims, fused=True)
it = torch.ones(slots, device=dev)
for _ in range(2):
    run.move_round(it)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
t0 = time.perf_counter()
ev[0].record()
noise = run.draw_noise()
eng.begin_search()
ev[1].record()
for _ in range(sims):
    run.sim_step(noise)
ev[2].record()
run.draw_expo()
eng.play_moves(it, run.expo, None, run.actions, run.done)
done = run.done.cpu()
ev[3].record()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"noise+begin {ev[0].elapsed_time(ev[1]):.2f} ms, {sims} sim steps {ev[1].elapsed_time(ev[2]):.2f} ms ({ev[1].elapsed_time(ev[2]) / sims * 1e3:.1f} us each), "
      f"expo+play+done {ev[2].elapsed_time(ev[3]):.2f} ms, wall {1e3 * (t1 - t0):.2f} ms")
