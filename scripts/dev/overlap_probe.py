"""Does running two half-batches on two streams hide the small kernels behind the other half's tower?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from alphazero_amd import Engine
from alphazero_amd.arena import SelfPlayRunner
from alphazero_amd.nets import GomokuNet

torch.manual_seed(0)
dev = torch.device("cuda", 0)
sims = int(os.environ.get("SIMS", "200"))
DT = torch.float32 if os.environ.get("DTYPE", "bf16") == "f32" else torch.bfloat16
model = None  # built in __main__ (nothing touches the GPU at import)


def make(slots, graph):
    eng = Engine(2, slots, sims, board_size=15, nn_dtype=DT, device=dev)
    run = SelfPlayRunner(eng, model, sims, fused=True, use_graph=graph)
    return eng, run


def bench(parts, graph, moves=3):
    slots = 16384 // parts
    ers = [make(slots, graph) for _ in range(parts)]
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    it = [torch.ones(slots, device=dev) for _ in range(parts)]
    def move():
        noises = []
        for (eng, run), s in zip(ers, streams):
            with torch.cuda.stream(s):
                noises.append(run.draw_noise())
                if graph and run._graph is None:
                    run._capture(noises[-1])
                eng.begin_search()
        for _ in range(sims):
            for (eng, run), s, nz in zip(ers, streams, noises):
                with torch.cuda.stream(s):
                    if graph:
                        run._graph.replay()
                    else:
                        run.sim_step(nz)
        for (eng, run), s, t in zip(ers, streams, it):
            with torch.cuda.stream(s):
                run.draw_expo()
                eng.play_moves(t, run.expo, None, run.actions, run.done)
    for _ in range(4):
        move()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(moves):
        move()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for eng, _ in ers:
        eng.check()
    print(f"parts={parts} graph={graph}: {dt / moves / sims * 1e6:.1f} us per simulation step of 16384 trees, {16384 * sims * moves / dt / 1e6:.2f} M sims/s", flush=True)
    del ers


if __name__ == "__main__":
    model = GomokuNet(15, 64, 2).eval().to(dev).to(DT)
    for parts, graph in ((1, False), (2, False), (4, False)):
        bench(parts, graph)
