"""Small configurations (tic-tac-toe / dots-and-boxes / gomoku 9x9): does splitting the games over P engines on P streams (each
replaying its simulation step as a HIP graph) fill the chip better than one engine?  WORKLOAD=dab|tictactoe|gomoku9"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from alphazero_amd import Engine
from alphazero_amd.arena import SelfPlayRunner

dev = torch.device("cuda", 0)
wl = os.environ.get("WORKLOAD", "dab")
game, n, sims, slots_total, _ = bench.WORKLOADS[wl]


def run(parts, moves=6):
    model = bench.make_model(wl).to(dev)
    slots = slots_total // parts
    ers = []
    for _ in range(parts):
        eng = Engine(game, slots, sims, board_size=n, nn_dtype=torch.float32, device=dev)
        # one step per graph: move() replays sims - 1 times
        ers.append((eng, SelfPlayRunner(eng, model, sims, fused=True, use_graph=True, graph_unit=1)))
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    it = [torch.ones(slots, device=dev) for _ in range(parts)]

    def move():
        noises = []
        for (eng, r), s in zip(ers, streams):
            with torch.cuda.stream(s):
                noises.append(r.draw_noise())
                if r._graph is None:
                    r._capture(noises[-1])
                eng.begin_search()
                eng.select_gather(None)
        for _ in range(sims - 1):
            for (eng, r), s in zip(ers, streams):
                with torch.cuda.stream(s):
                    r._graph.replay()
        for (eng, r), s, t, nz in zip(ers, streams, it, noises):
            with torch.cuda.stream(s):
                r.evaluate()
                eng.expand_backup(r.logits, r.values, nz)
                r.draw_expo()
                eng.play_moves(t, r.expo, None, r.actions, r.done)
                eng.reset_games(r.done)

    for _ in range(3):
        move()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(moves):
        move()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for eng, _ in ers:
        eng.check()
    print(f"{wl} parts={parts}: {dt / moves / sims * 1e6:.1f} us per simulation step of {slots * parts} trees, "
          f"{slots * parts * sims * moves / dt / 1e6:.2f} M sims/s", flush=True)


if __name__ == "__main__":
    for p in (1, 2, 3, 4):
        run(p)
