#!/usr/bin/env python3
"""bench.py -- MCTS simulations/s (+ self-play games/s) of the HIP many-tree engine on MI355X.

  python bench.py [--gpus N --steps K --warmup W]            (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json metric: "MCTS sims/sec + self-play games/sec, gomoku 15x15"; configs[3]):
gomoku 15x15, 400 simulations per move, 16,384 concurrent games per GPU, the reference's GomokuModel
(64 ch, 2 residual blocks) with random-init weights (seed 0), synthetic self-play (no dataset needed).
The PRIMARY line runs the network in float32, the reference's own arithmetic (nn_models/base.py:17-19 under
core/MCTS_alphazero.py:161-162, no autocast), on the hand-written f32 MFMA tower (az_tower_f32.hip); the same
workload with the bf16 tower is measured right after it and reported as the object "throughput_mode", with a
measured decision-agreement figure against the f32 search.
One STEP = one move of self-play for every game on the GPU: Dirichlet draw, 400 x (select+gather ->
network forward on the 16,384-position batch -> expand+backup), temperature sampling, game step,
emission of finished games' (state, pi, z) samples into a pinned host ring, slot recycling.
So one step = slots x 400 simulations per GPU.  Inputs live in HBM; only finished samples cross PCIe.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel of the step.  With the fused evaluator that is the conv tower (k_tower_f32 /
                  k_tower, MFMA bound): the convolutions' algorithmic flops for the boards one launch evaluates / mean
                  launch time (HIP events on the launch stream around the kernel alone), vs the dense MFMA peak of the
                  dtype (157.3 TFLOP/s f32, 2.5 PFLOP/s bf16).  Otherwise the slower tree kernel (HBM bound).
  roofline_tree-- the hand-written tree kernel of a simulation step (k_expand_select: backup of simulation i + descent
                  of simulation i+1 in one launch): algorithmic bytes per launch / mean launch time, vs 8 TB/s HBM.
  nn_eval      -- the whole network forward (tower + head linears, or PyTorch/MIOpen with --evaluator model):
                  achieved TFLOP/s on the boards actually evaluated vs the same peak.
  cpu_baseline -- the CPU oracle (oracle/, "port" of the reference path, proven equal to it on the
                  golden fixtures) timed on this box's host cores with the same network on torch-CPU.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (game id, board, sims/move, slots per GPU, net kwargs)
    "gomoku15": (2, 15, 400, 16384, dict(board_size=15, channels=64, blocks=2)),
    "gomoku9": (2, 9, 200, 4096, dict(board_size=9, channels=64, blocks=2)),
    # configs[4]: deep towers (the reference's quick_model scaled: 128 channels x 10 residual blocks = 21 conv layers,
    # 1.39 GFLOP per evaluation; and the same depth at 64 channels), both on the fused MFMA kernel
    "gomoku15_deep": (2, 15, 800, 16384, dict(board_size=15, channels=128, blocks=10)),
    "gomoku15_deep64": (2, 15, 800, 16384, dict(board_size=15, channels=64, blocks=10)),
    # configs[4] as written -- "quick_model scaled up": the reference's QuickModel (nn_models/quick_model.py; nets.QuickNet) on the
    # 15x15 board with its default 128 channels and 5-layer residual blocks, RESNET_BLOCK_NUM 1 -> 4: 21 conv layers like
    # gomoku15_deep, plus the 3x3 head convs; on the tower kernels' QuickModel instantiations (f32 primary, bf16 throughput mode)
    "gomoku15_quick": (2, 15, 800, 16384, dict(board_size=15, channels=128, blocks=4, layers=5)),
    "dab": (1, 3, 80, 3000, {}),
    "tictactoe": (0, 3, 50, 1000, {}),
}


PRIOR_SHARPNESS = 1.0  # --prior-sharpness: the policy head's last Linear layer (weights and bias) is multiplied by this


def make_model(workload: str):
    """the workload's network, random init (seed 0).  A random-init network gives nearly flat priors, hence shallow trees
    (mean selection depth 1.5 on gomoku 15x15); a TRAINED network's priors are peaked and its trees 4-8 levels deep.
    --prior-sharpness s > 1 stands in for that: logits x s (folded into the policy head's last Linear layer, so the
    evaluation costs exactly what it costs at s = 1).  Diagnostic only: the default line runs at s = 1."""
    import torch
    from alphazero_amd import nets

    torch.manual_seed(0)
    game = WORKLOADS[workload][0]
    if game == 2:
        kw = WORKLOADS[workload][4]
        m = (nets.QuickNet(**kw) if "layers" in kw else nets.GomokuNet(**kw)).eval()
    else:
        m = (nets.DotsAndBoxesNet() if game == 1 else nets.TicTacToeNet()).eval()
    if PRIOR_SHARPNESS != 1.0:
        last = [x for x in m._policy_head if isinstance(x, torch.nn.Linear)][-1]
        with torch.no_grad():
            last.weight.mul_(PRIOR_SHARPNESS)
            last.bias.mul_(PRIOR_SHARPNESS)
    return m


# ------------------------------------------------------------------------------ CPU baseline ----
def _cpu_worker(args):
    """One single-threaded self-play worker on the CPU oracle (mirrors _self_player_worker,
    training/alphazero_trainer.py:164-221: torch.set_num_threads(1), own seed)."""
    global PRIOR_SHARPNESS
    workload, seed, budget_s, PRIOR_SHARPNESS = args
    import numpy as np
    import torch

    torch.set_num_threads(1)
    from oracle import az_oracle as O

    game, n, sims, _, _ = WORKLOADS[workload]
    model = make_model(workload)
    d = O.dims(game, n)
    ev = O.torch_eval(model, (d["cin"], d["H"], d["W"]))
    rng = np.random.default_rng(seed)
    A = d["A"]
    done_sims = 0
    games = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        s = np.zeros(d["numel"], np.float32)
        hist = np.zeros((8, d["numel"]), np.float32)
        player = 1
        while time.perf_counter() - t0 < budget_s:
            noise = rng.dirichlet([0.3] * A).astype(np.float32)
            r = O.search(game, n, s, player, hist, sims, ev, noise=noise)
            done_sims += sims
            expo = rng.exponential(1.0, A).astype(np.float32)
            a = int(np.argmax(r["pi"] / expo))
            s, player = O.next_state(game, n, s, a, player)
            hist = np.concatenate([hist[1:], s[None]])
            if O.status(game, n, s)[0]:
                games += 1
                break
    return done_sims, games, time.perf_counter() - t0


def cpu_baseline(workload: str, budget_s: float = 12.0) -> dict:
    import multiprocessing as mp

    from oracle import az_oracle as O

    O.build()
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 4)
    share = min(share, os.cpu_count() or share)
    workers = max(1, share - 2)  # the reference's "auto" rule: cores - 2 (config/train_config.py:35-38)
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_worker, [(workload, 1000 + i, budget_s, PRIOR_SHARPNESS) for i in range(workers)])
    sims = sum(r[0] for r in res)
    wall = max(r[2] for r in res)
    game, n, nsims, _, _ = WORKLOADS[workload]
    return {"value": sims / wall, "unit": "sims/s", "cores": workers, "kind": "port",
            "sample": f"{workers} single-thread workers x {budget_s:.0f} s of {workload} self-play "
                      f"({nsims} sims/move, same net on torch-CPU, batch 1) on the CPU oracle",
            "per_core": sims / wall / workers,
            # the reference's own Python, for orientation (measured in the survey container, 1 process,
            # torch.set_num_threads(1), shimmed reference; it cannot run on the GPU box): labelled constants, not measurements of this run
            "reference_python_1thread_sims_per_s": {"tictactoe": 1230, "dab": 370, "gomoku15": 59}.get(
                "gomoku15" if workload.startswith("gomoku15") else workload),
            "reference_python_where": "survey container, 8 host cores, one single-thread process (SURVEY.md section 6); README "
                                      "wall-clock: tic-tac-toe 1000 games in 59 s, dots-and-boxes 3000 games in 1883 s (i9 + RTX 4070 laptop)"}


# ------------------------------------------------------------------------------ GPU bench -------
PEAK_TFLOPS = {"f32": 157.3, "f32w": 157.3, "bf16": 2500.0, "f16": 2500.0, "f16x2": 2500.0, "f16r": 2500.0, "bf16r": 2500.0}  # dense MFMA peaks, MI355X_MICROARCH.md
# "f16x2" (az_tower_h2.hip): f32 operands split into two f16 halves, THREE f16 MFMAs per product block, f32 accumulation --
# f32-level accuracy on the f16 pipe.  Its roofline counts the flops the MFMA pipe executes (3 x the convolution's) against the
# f16 peak; the convolution's own flops per second are reported beside it as f32_equivalent_tflops.
H2_MFMA_PER_PRODUCT = 3.0


def measure(args, dtype_name, steps, warmup, dev, rank, world, dst, with_probe=True):
    """One timed self-play run of the workload at one network precision -> dict of results (rank 0) / None."""
    import torch
    import torch.distributed as dist

    use_dist = dst is not None
    from alphazero_amd import Engine
    from alphazero_amd.arena import END_TEMPERATURE, START_TEMPERATURE, TEMPERATURE_DECAY, SelfPlayRunner, reference_tao
    from alphazero_amd.engine import copy_to_host
    from alphazero_amd.samples import HostSampleRing, WeightPublisher

    game, n, sims, slots, _ = WORKLOADS[args.workload]
    sims = args.sims or sims
    slots = args.slots or slots
    tdtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32w": torch.float32, "f16x2": torch.float32,
              "f16r": torch.float32, "bf16r": torch.float32}[dtype_name]
    h2 = dtype_name == "f16x2"
    # towers asked for by name on an f32 engine: az_tower_h2.hip ("f16x2"), az_tower_h1.hip (the row-tile tower: "f16r" = f16
    # operands, "bf16r" = bf16 operands)
    named_tower = {"f16x2": "f16x2", "f16r": "f16", "bf16r": "bf16r", "f32w": "f32w"}.get(dtype_name)
    if dtype_name == "f32" and game == 2 and "layers" not in WORKLOADS[args.workload][4]:
        named_tower = "f32"  # the direct-convolution tower by name (left to itself the runner picks the Winograd tower where it fits)
    model = make_model(args.workload).to(dev)
    if tdtype != torch.float32:
        model = model.to(tdtype)
    model = model.to(memory_format=torch.channels_last)
    publisher = WeightPublisher(model, src=0)
    torch.manual_seed(1234 + rank)  # (only the agreement probe's opening moves still draw from torch's generators)
    torch.cuda.manual_seed(1234 + rank)
    # Dirichlet / Exponential rows: the engine's per-game counter-based RNG (az_draw_rng), Philox(seed = 1234) keyed on the
    # GLOBAL game id and the ply -- slot g of rank r starts with game r + world * g, the k-th game a rank
    # starts later is r + world * k: the same games are played whatever the number of GPUs they are spread over
    SEED = 1234

    eng = Engine(game, slots, sims, board_size=n, nn_dtype=tdtype, device=dev, leaves_per_step=max(args.leaves, 1))
    d = eng.dims
    evaluator = args.evaluator
    if evaluator == "auto":
        from alphazero_amd.fused import FusedTowerEvaluator, SmallNetEvaluator

        ok = FusedTowerEvaluator.precision_for(model, tdtype) is not None  # (QuickModel family: f32 tower only)
        ok = ok or (game != 2 and tdtype == torch.float32 and SmallNetEvaluator.supports(model))  # az_smallnet.hip (f32 only)
        evaluator = "fused" if ok else "model"
    run = SelfPlayRunner(eng, model, sims, channels_last=True, fused=(named_tower or evaluator == "fused"), use_graph=bool(args.graph),
                         graph_unit=args.graph_unit, seed=SEED)
    run.game_ids.copy_(rank + world * torch.arange(slots, dtype=torch.int64))
    if evaluator == "hash":  # tree kernels alone (no network): the synthetic evaluator reads the packed leaf batch, as the towers do
        class _PackedOnly:  # (makes the runner pass nn_input = NULL: no dense planes are written)
            kernel_events, generation = None, 0

            @staticmethod
            def overflowed():
                return False

        run.fused = _PackedOnly()
        run.evaluate = lambda: eng.eval_hash(None, run.logits, run.values, args.hash_scale)
    wave = eng.leaves_per_step > 1  # virtual loss / in-tree leaf batching: several leaves per tree per network batch
    ring = HostSampleRing(max(4 * slots, 65536), (d.nn_channels, d.height, d.width), d.n_action)
    esize = {torch.float32: 4, torch.bfloat16: 2, torch.float16: 2}[tdtype]

    copy_stream = torch.cuda.Stream(dev)
    games_started = slots
    ids_started = slots  # game ids this rank has handed out (id = rank + world * k)
    inv_tao_h = torch.tensor([1.0 / reference_tao(0)] * slots, dtype=torch.float32).pin_memory()
    inv_tao = inv_tao_h.to(dev)
    finished_games = 0
    finished_plies = 0

    # Samples leave the GPU as int8 planes (+-1 / 0 are exact) + f32 pi + z: a quarter of the f32 bytes over PCIe (the
    # D2H runs as a copy kernel on this stack and slows the search kernels it overlaps with).  The f32 ring rows are
    # written by the host from the pinned staging buffers one move later, while the GPU runs the next search.
    pending = []

    def finalize_pending():
        while pending:
            ev, s8_h, pis_h, zs_h = pending.pop(0)
            ev.synchronize()
            ring.put_batch(s8_h, pis_h, zs_h)  # copy_ converts int8 -> f32 into the ring

    def collect(done):
        """emit + D2H the samples of the games that just ended, restart their slots"""
        nonlocal games_started, ids_started, finished_games, finished_plies
        plies, _, _, _ = eng.game_info()
        both = torch.stack((done.to(torch.int32), plies)).cpu()  # the move's one host sync
        idx = torch.nonzero(both[0]).reshape(-1)
        if idx.numel() == 0:
            return
        plies_h = both[1][idx]
        states, pis, zs, _ = eng.emit_samples(idx.to(torch.int32), plies_h)
        eng.reset_games(done)
        s8 = states.to(torch.int8)
        copy_stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(copy_stream):
            staged = []
            for t in (s8, pis, zs):
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                copy_to_host(h, t)  # 16 workgroups: a D2H copy_ is a chip-wide copy kernel here and stalls the search
                t.record_stream(copy_stream)
                staged.append(h)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        pending.append((ev, *staged))
        nf = int(idx.numel())
        finished_games += nf
        finished_plies += int(plies_h.sum())
        # per-slot temperature schedule restarts with the new game (reference_tao, vectorised)
        run.game_ids[idx.to(dev)] = (rank + world * (ids_started + torch.arange(nf, dtype=torch.int64))).to(dev)
        ids_started += nf
        k = (games_started + torch.arange(nf)) // slots
        inv_tao_h[idx] = (1.0 / torch.clamp(START_TEMPERATURE * TEMPERATURE_DECAY ** k.double(), min=END_TEMPERATURE)).float()
        games_started += nf
        inv_tao.copy_(inv_tao_h, non_blocking=True)

    # --- pre-roll: stagger game ages so that games finish at a steady rate during the timed steps
    if args.preroll > 0:
        run.num_sims = 2
        R = args.preroll
        ar = torch.arange(slots, device=dev)
        for r in range(R):
            _, done = run.move_round(inv_tao)
            again = (ar % R == r) | (done != 0)
            eng.reset_games(again.to(torch.uint8))
            run.game_ids += again.to(torch.int64) * (world * slots)  # a restarted slot is a new game: a fresh id of this rank's residue class
            ids_started += slots  # (ids below rank + world * ids_started are taken)
        run.num_sims = sims
        run.check()  # the engine's error flag + an f16 tower's overflow flag

    ev_tree, ev_nn, ev_tower = [], [], []
    EVENT_EVERY = 8  # every 8th simulation step of the timed region is bracketed with HIP events

    phase = [0]

    def timed_round():
        """run.move_round with HIP events around each tree kernel and each network forward"""
        ph = phase[0] % EVENT_EVERY  # which simulations are bracketed rotates from move to move: an unbiased sample of the
        phase[0] += 3                # move (its first launches share the chip with the previous move's sample copy)
        noise = run.draw_noise()
        nn_in = None if run.fused is not None else run.nn_input
        eng.begin_search()
        eng.select_gather(nn_in)  # the move's first descent; every later one rides with the previous simulation's backup
        for k in range(sims - 1):
            if (k + ph) % EVENT_EVERY:  # un-bracketed simulation step (an event record costs the stream ~5 us)
                if run.fused is not None:
                    run.fused.kernel_events = None
                run.eval_step(noise)
                continue
            if run.fused is not None:
                run.fused.kernel_events = ev_tower  # events around the tower kernel alone, inside FusedTowerEvaluator
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record()
            run.evaluate()
            b.record()
            eng.expand_select(run.logits, run.values, noise, nn_in)
            c.record()
            ev_nn.append((a, b)); ev_tree.append((b, c))
        if run.fused is not None:
            run.fused.kernel_events = None
        run.evaluate()
        eng.expand_backup(run.logits, run.values, noise)
        run.draw_expo()
        eng.play_moves(inv_tao, run.expo, None, run.actions, run.done)
        return run.done

    def step(timed: bool):
        if publisher.publish(force=args.force_dist) and run.fused is not None:  # learner -> self-play weight push (RCCL broadcast; no-op at N = 1)
            run.fused.refresh()  # re-fold BatchNorm / re-pack the published weights for the fused tower
        done = timed_round() if (timed and not args.graph and not wave) else run.move_round(inv_tao)[1]
        finalize_pending()  # the previous move's samples -> the host ring, under the search just queued
        collect(done)

    for _ in range(warmup):
        step(False)
    finalize_pending()
    run.check()  # the engine's error flag + an f16 tower's overflow flag
    eng.reset_stats()
    finished_games = finished_plies = 0
    torch.cuda.synchronize()
    if use_dist:
        dst.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    finalize_pending()  # the last move's samples belong to the timed region
    torch.cuda.synchronize()
    if use_dist:
        dst.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    run.check()  # the engine's error flag + an f16 tower's overflow flag
    st = eng.read_stats()
    if args.graph and not wave:  # kernel times: one eager, event-bracketed probe step outside the timed region
        eng.reset_stats()
        keep = (finished_games, finished_plies)
        collect(timed_round())
        torch.cuda.synchronize()
        finished_games, finished_plies = keep
        st_probe = eng.read_stats()
    else:
        st_probe = st

    tt = torch.tensor([elapsed, float(st["sims"]), float(finished_games), float(finished_plies)], dtype=torch.float64, device=dev)
    if use_dist:
        tmax = tt.clone()
        dst.all_reduce(tmax, dist.ReduceOp.MAX)
        dst.all_reduce(tt, dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_sims, total_games, total_plies = float(tt[1]), float(tt[2]), float(tt[3])

    # what an empty event pair reads on this stream (record -> record): subtracted from every bracketed launch
    empty = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record(); empty.append((a, b))
    torch.cuda.synchronize()
    ev_overhead = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2]
    tree_ms_raw = sum(a.elapsed_time(b) for a, b in ev_tree) / max(len(ev_tree), 1)
    nn_ms_raw = sum(a.elapsed_time(b) for a, b in ev_nn) / max(len(ev_nn), 1)
    tower_ms_raw = sum(a.elapsed_time(b) for a, b in ev_tower) / max(len(ev_tower), 1) if ev_tower else None
    has_fused = run.fused is not None
    eng_k = eng.leaves_per_step
    del run, ring, pending
    eng.close()
    del eng
    torch.cuda.empty_cache()

    # Games per second.  The games that end inside the timed steps were aged by the pre-roll's cheap moves, so their count
    # says nothing about a steady state.  games/s = plies/s (measured above) / mean plies of a game played start to finish
    # with the full search: measured here on a small population, outside the timed region.
    game_len = None
    if rank == 0 and with_probe and args.game_probe > 0 and evaluator != "hash":
        pe = Engine(game, args.game_probe, sims, board_size=n, nn_dtype=tdtype, device=dev, leaves_per_step=max(args.leaves, 1))
        pr = SelfPlayRunner(pe, model, sims, channels_last=True, fused=(named_tower or evaluator == "fused"), use_graph=bool(args.graph),
                            graph_unit=args.graph_unit, seed=SEED)
        pr.game_ids += 1 << 40  # (games of their own, outside every rank's id range)
        pit = torch.full((args.game_probe,), 1.0 / reference_tao(0), device=dev)
        lengths, moves, tp0 = [], 0, time.perf_counter()
        while len(lengths) < args.game_probe and moves <= d.max_plies and time.perf_counter() - tp0 < 60.0:
            done_h = pr.move_round(pit)[1].cpu().bool()
            moves += 1
            if bool(done_h.any()):
                lengths += pe.game_info()[0].cpu()[done_h].tolist()
        pe.check()
        if lengths:
            game_len = {"mean_plies": sum(lengths) / len(lengths), "games": len(lengths), "of": args.game_probe,
                        "seconds": time.perf_counter() - tp0,
                        "how": f"{args.game_probe} games played start to finish with the same network and {sims} sims/move, tao = 1"}
        del pr
        pe.close()
        del pe

    if rank != 0:
        return None

    def corr(ms):
        return max(ms - ev_overhead, 1e-6)

    if wave:  # no per-kernel brackets in wave mode (the step count of a move is data dependent): throughput only
        return {
            "metric": "mcts_sims_per_sec", "value": total_sims / elapsed, "unit": "sims/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {sims} sims/move, {slots} concurrent games per GPU, evaluator={evaluator} "
                                   f"({dtype_name} network arithmetic), virtual loss with {eng_k} leaves per tree per network batch "
                                   "(az_select_wave / az_expand_wave; NOT the reference's sequential search)",
                       "slots_per_gpu": slots, "sims_per_move": sims, "leaves_per_step": eng_k,
                       "parallelism": f"games sharded x{world}, no data-path collective"},
            "plies_per_sec": slots * world * steps / elapsed,
            "games_per_sec": (slots * world * steps / elapsed / game_len["mean_plies"]) if game_len else None,
            "game_length": game_len, "roofline": None,
            "evals_per_sim": st["evals"] / max(st["sims"], 1), "mean_depth": st["sum_depth"] / max(st["sims"], 1)}

    peak = PEAK_TFLOPS[dtype_name]
    nin = d.nn_channels * d.height * d.width
    launches = sims * (1 if args.graph else steps)  # launches of each per-simulation kernel behind st_probe
    tree_ms, nn_ms = corr(tree_ms_raw), corr(nn_ms_raw)
    st = st_probe  # byte counts must come from the same launches the events bracket
    # algorithmic bytes.  A child record has a light part (prior f32 + action u8 = 5 B), read for every
    # child a descent scans and written at expansion, and a heavy part ({visits, meta} 8 B + value f64 8 B) that exists -- and
    # is read -- only for the children a descent has touched; a first touch writes 30 B (the swap of two light parts, the
    # heavy part's initial values, the block's touched count).  Leaf batch = packed ring 32*SW B (+ the dense planes only
    # when a PyTorch forward consumes them)
    planes = 0.0 if has_fused else nin * esize
    fixed_sel = 4.0 * st["sum_depth"] + st["evals"] * (32.0 * d.state_words + planes) + st["sims"] * (16.0 + 36.0 * d.state_words + 16.0)
    fixed_exp = st["evals"] * (4.0 * d.n_action + 4.0 + 4.0 * d.state_words) + 32.0 * st["sum_depth"] + 36.0 * st["sims"]
    sel_bytes = (5.0 * st["sum_children"] + 16.0 * st["sum_touched"] + 30.0 * st["evals"] + fixed_sel) / launches
    exp_bytes = (5.0 * st["expanded_edges"] + fixed_exp) / launches
    # the same launches priced as round 2's layout would have moved them (20 B per scanned child, 12 B per new record):
    # what the kernel's time is compared with when the question is "how much faster than the old layout's roofline"
    r02_bytes = (20.0 * st["sum_children"] + fixed_sel + 12.0 * st["expanded_edges"] + fixed_exp) / launches
    # the tree work of a simulation step is ONE launch: backup of simulation i + descent of simulation i+1
    dom = ("k_expand_select", sel_bytes + exp_bytes, tree_ms)
    tower_kernel = "k_smallnet_f32" if game != 2 else ("k_tower_f32_t16" if dtype_name == "f32" else "k_tower_f32w" if dtype_name == "f32w" else
                                                       "k_tower_h2" if h2 else "k_tower_h1" if named_tower else "k_tower")

    # HBM traffic from PMC counters cannot be collected inside this run (separate rocprofv3 --pmc passes,
    # scripts/pmc_traffic.sh); the last committed measurement of this same workload is reported, if present
    def pmc_traffic(kernel):
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
        if args.workload != "gomoku15" or evaluator != "fused" or not os.path.exists(tpath):
            return None, None
        try:
            d = json.load(open(tpath))
            t = d[kernel] if kernel in d else d[kernel.replace("_t16", "")]
            return (t["FETCH_SIZE"]["corrected_bytes_per_launch"] + t["WRITE_SIZE"]["corrected_bytes_per_launch"],
                    "bytes per launch, FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, from separate --pmc passes "
                    "(scripts/pmc_traffic.sh -> profiles/pmc_traffic_latest.json)")
        except Exception:
            return None, None

    traffic, traffic_note = pmc_traffic(dom[0])
    achieved = dom[1] / (dom[2] * 1e-3) / 1e9 if dom[2] > 0 else 0.0
    tree_roofline = {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": traffic, "traffic_note": traffic_note, "bytes_per_launch": dom[1],
                     "ms_per_launch": dom[2],
                     "round2_layout": {"bytes_per_launch": r02_bytes, "frac_at_this_time": (r02_bytes / (dom[2] * 1e-3) / 1e9 / 8000.0) if dom[2] > 0 else 0.0,
                                       "note": "the same descents and expansions priced with round 2's 20-byte scanned / 12-byte written child "
                                               "records: what this launch time would be as a fraction of 8 TB/s had the bytes not been cut"},
                     "touched_share_of_scanned": st["sum_touched"] / max(st["sum_children"], 1)}
    evals_per_launch = st["evals"] / launches  # boards the network really evaluates per simulation step
    roofline = tree_roofline
    nn_eval = None
    if evaluator in ("model", "fused"):
        import torch.nn as tnn
        conv_flops = sum(2.0 * d.height * d.width * m.in_channels * m.out_channels * m.kernel_size[0] * m.kernel_size[1]
                         for m in model.modules() if isinstance(m, tnn.Conv2d))
        lin_flops = sum(2.0 * m.in_features * m.out_features for m in model.modules() if isinstance(m, tnn.Linear))
        flops_per_eval = model.flops_per_eval() if hasattr(model, "flops_per_eval") else conv_flops + lin_flops
        fl = flops_per_eval * evals_per_launch
        nn_eval = {"bound": "mfma", "ms_per_batch": nn_ms, "achieved": fl / (nn_ms * 1e-3) / 1e12 if nn_ms else 0.0,
                   "peak": peak, "unit": "TFLOP/s", "frac": (fl / (nn_ms * 1e-3) / 1e12 / peak) if nn_ms else 0.0,
                   "flops_per_eval": flops_per_eval, "boards_per_batch": evals_per_launch,
                   "share_of_step": nn_ms / max(nn_ms + tree_ms, 1e-9),
                   "implementation": (("az_smallnet.hip: k_smallnet_f32 (the whole network, one f32 MFMA kernel)" if game != 2 else
                                       ("az_tower_f32.hip: k_tower_f32_t16<QuickModel> (f32 MFMA conv tower, 16-cell tiles, 3x3 head convs) + k_quick_heads_f32"
                                        if dtype_name == "f32" else "az_tower.hip: k_tower<QuickModel> (fused bf16 MFMA conv tower, 3x3 head convs) + k_quick_heads")
                                       if "layers" in WORKLOADS[args.workload][4] else
                                       "az_tower_f32.hip: k_tower_f32_t16 (f32 MFMA conv tower, direct convolution, 16-cell tiles) + k_heads_f32" if dtype_name == "f32" else
                                       "az_tower_w.hip: k_tower_f32w (f32 MFMA conv tower, 3x3 convolutions by the Winograd transform F(2x2,3x3)) + k_heads_f32" if dtype_name == "f32w" else
                                       "az_tower_h2.hip: k_tower_h2 (f32 operands as two f16 halves, three f16 MFMAs per product, f32 accumulation) + k_heads_f32" if h2 else
                                       f"az_tower_h1.hip: k_tower_h1 (row-tile conv tower, {'f16' if dtype_name == 'f16r' else 'bf16'} MFMA, f32 accumulation) + k_heads_f32" if named_tower else
                                       "az_tower.hip: k_tower (fused bf16 MFMA conv tower) + k_heads (MFMA head linears)")
                                      if evaluator == "fused" else "PyTorch/MIOpen module forward")}
        if evaluator == "fused" and tower_ms_raw is not None:
            # the dominant kernel of the step: the fused conv tower (MFMA bound).  Algorithmic flops = the convolutions'
            # multiply-adds x 2 for the boards this launch evaluates (terminal leaves are skipped, padding is not counted)
            tw_ms = corr(tower_ms_raw)
            tw_fl = (conv_flops if game == 2 else flops_per_eval) * evals_per_launch  # k_smallnet_f32 holds the head layers too
            f32_equiv = tw_fl / (tw_ms * 1e-3) / 1e12
            if h2:
                tw_fl *= H2_MFMA_PER_PRODUCT  # what the f16 pipe executes
            ttraffic, tnote = pmc_traffic("k_tower_h1_f16" if dtype_name == "f16r" else tower_kernel)
            net_roofline = {"bound": "mfma", "kernel": tower_kernel, "achieved": tw_fl / (tw_ms * 1e-3) / 1e12, "peak": peak,
                        "unit": "TFLOP/s", "frac": tw_fl / (tw_ms * 1e-3) / 1e12 / peak, "traffic": ttraffic,
                        "traffic_note": tnote, "flops_per_launch": tw_fl, "boards_per_launch": evals_per_launch,
                        "ms_per_launch": tw_ms, "share_of_step": tw_ms / max(nn_ms + tree_ms, 1e-9)}
            if dtype_name == "f32w":
                # what the MFMA pipe executes: 16 transform points x 64 tiles of 2x2 outputs (9x9: 32 tile slots) x 64 x 64 (stem: 16 of the
                # padded input channels) multiply-adds per layer = (4 + 16 L) k-steps x 256 (128) MFMAs of 2,048 flops per board
                n_l = sum(1 for m in model.modules() if isinstance(m, tnn.Conv2d) and m.kernel_size == (3, 3)) - 1
                ex_fl = (4 + 16 * n_l) * (256 if d.height == 15 else 128) * 2048.0 * evals_per_launch  # (9x9: two tile groups of 16 per board)
                net_roofline["executed"] = {"flops_per_launch": ex_fl, "achieved": ex_fl / (tw_ms * 1e-3) / 1e12, "frac": ex_fl / (tw_ms * 1e-3) / 1e12 / peak}
                net_roofline["note"] = ("achieved / frac price the ALGORITHMIC flops of the network (the direct 3x3 convolutions' multiply-adds x 2, as for "
                                        "every other tower) against the f32 MFMA peak; the kernel computes them by the Winograd transform F(2x2,3x3) in f32, "
                                        "which executes 2.1x fewer multiply-adds, so frac may exceed 1; 'executed' = the MFMA flops really issued "
                                        "(the pipe's utilisation; the rest of the time goes to the transforms' VALU adds, which cannot co-execute with f32 MFMAs)")
            if h2:
                net_roofline["f32_equivalent_tflops"] = f32_equiv
                net_roofline["note"] = ("achieved / peak count the f16 MFMA flops executed (3 per multiply-add of the convolution: w1 x1 + w1 x2 + "
                                        "w2 x1); f32_equivalent_tflops = the convolution's own flops per second, comparable with the f32 tower's 'achieved'")
            # the dominant kernel of the step: the network kernel for the gomoku towers; for the two small nets the tree
            # kernel runs longer than the whole network and stays the headline roofline
            if game == 2 or tw_ms >= tree_ms:
                roofline = net_roofline
            else:
                nn_eval["roofline_network_kernel"] = net_roofline
    out = {
        "metric": "mcts_sims_per_sec", "value": total_sims / elapsed, "unit": "sims/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"bf16r": "bf16", "f16r": "f16", "f32w": "f32"}.get(dtype_name, dtype_name), "data": "synthetic",
        "config": {"workload": f"{args.workload}: {sims} sims/move, {slots} concurrent games per GPU, "
                               f"evaluator={evaluator} ({dtype_name} network arithmetic), one step = one self-play move of every game",
                   "slots_per_gpu": slots, "sims_per_move": sims, "parallelism": f"games sharded x{world}, no data-path collective",
                   **({"prior_sharpness": PRIOR_SHARPNESS} if PRIOR_SHARPNESS != 1.0 else {})},
        "plies_per_sec": slots * world * steps / elapsed,
        "games_per_sec_finished_in_window": total_games / elapsed,
        "games_per_sec": (slots * world * steps / elapsed / game_len["mean_plies"]) if game_len else None,
        "games_per_sec_note": "games_per_sec_finished_in_window counts the games that really ended inside the timed steps (aged "
                              "by the pre-roll); games_per_sec = plies_per_sec / game_length.mean_plies is the steady-state rate",
        "game_length": game_len,
        "finished_in_window": {"games": total_games, "mean_plies": (total_plies / total_games) if total_games else None},
        "roofline": roofline,
        "roofline_tree": tree_roofline,
        "tree_kernels": {"k_expand_select": {"ms": tree_ms, "bytes": sel_bytes + exp_bytes,
                                             "GBps": (sel_bytes + exp_bytes) / max(tree_ms, 1e-9) / 1e6,
                                             "bytes_descent": sel_bytes, "bytes_expand_backup": exp_bytes},
                         "event_pair_overhead_ms": ev_overhead,
                         "timed_with": "probe step after the timed region (timed steps replay a HIP graph)" if args.graph
                         else "HIP events inside the timed region",
                         "mean_depth": st["sum_depth"] / max(st["sims"], 1),
                         "mean_children_per_level": st["sum_children"] / max(st["sum_depth"], 1),
                         "evals_per_sim": st["evals"] / max(st["sims"], 1)},
    }
    if nn_eval is not None:
        out["nn_eval"] = nn_eval
    return out


def decision_agreement(args, dev, n_pos=1024, plies=12, other="bf16"):
    """How often does another evaluator (the bf16 throughput mode; the f16x2 tower) decide like the f32 network?  The same n_pos mid-game positions (reached by
    `plies` moves of cheap self-play), the same Dirichlet rows, the full search with each precision: fraction of roots
    whose most-visited move is the same, whose move sampled with the same exponential row is the same, and whose whole
    visit vector is identical."""
    import torch

    from alphazero_amd import Engine
    from alphazero_amd.arena import SelfPlayRunner

    game, n, sims, _, kw = WORKLOADS[args.workload]
    sims = args.sims or sims
    if game != 2 or kw.get("channels") != 64:
        return None
    torch.manual_seed(99)
    torch.cuda.manual_seed(99)
    res, pos_moves = {}, None
    noise = expo = None
    named = {"f32": "f32", "f32w": "f32w", "f16x2": "f16x2", "f16r": "f16", "bf16r": "bf16r"}  # towers asked for by name on an f32 engine
    for name, dt in (("f32", torch.float32), (other, torch.bfloat16 if other == "bf16" else torch.float32)):
        model = make_model(args.workload).to(dev).to(dt)
        eng = Engine(game, n_pos, sims, board_size=n, nn_dtype=dt, device=dev)
        run = SelfPlayRunner(eng, model, sims, fused=named.get(name, True))
        A = eng.dims.n_action
        it = torch.ones(n_pos, device=dev)
        if pos_moves is None:  # the opening moves, played once (hash evaluator, 8 sims) and replayed for the other precision
            pos_moves = []
            lg, vl = eng.new(n_pos, A), eng.new(n_pos)
            for _ in range(plies):
                eng.begin_search()
                for _ in range(8):
                    eng.select_gather(None)
                    eng.eval_hash(None, lg, vl, 6.0)
                    eng.expand_backup(lg, vl, None)
                a, _ = eng.play_moves(it, torch.empty(n_pos, A, device=dev).exponential_(1))
                pos_moves.append(a.clone())
            noise = torch._sample_dirichlet(torch.full((n_pos, A), 0.3, device=dev))
            expo = torch.empty(n_pos, A, device=dev).exponential_(1)
        else:
            lg, vl = eng.new(n_pos, A), eng.new(n_pos)
            for a in pos_moves:
                eng.begin_search()
                eng.select_gather(None)
                eng.eval_hash(None, lg, vl, 6.0)
                eng.expand_backup(lg, vl, None)
                eng.select_gather(None)
                eng.eval_hash(None, lg, vl, 6.0)
                eng.expand_backup(lg, vl, None)
                eng.play_moves(it, torch.ones(n_pos, A, device=dev), a)
        run.draw_noise = lambda nz=noise, r=run: (r.noise.copy_(nz), r.noise)[1]
        run.search()
        visits = eng.new(n_pos, A, dtype=torch.int32)
        pi = eng.root_policy(visits=visits)
        run.check()  # the engine's error flag + an f16 tower's overflow flag
        _, _, active, _ = eng.game_info()
        res[name] = (visits.clone(), (pi / expo).argmax(1), active.bool().clone())
        del run
        eng.close()
    (v32, s32, a32), (v16, s16, a16) = res["f32"], res[other]
    live = a32 & a16
    nl = max(int(live.sum()), 1)
    return {"positions": nl, "sims_per_move": sims,
            "same_most_visited_move": float((v32.argmax(1) == v16.argmax(1))[live].sum()) / nl,
            "same_sampled_move": float((s32 == s16)[live].sum()) / nl,
            "identical_visit_vectors": float((v32 == v16).all(1)[live].sum()) / nl,
            "mean_abs_pi_difference": float(((v32.float() - v16.float()).abs().sum(1) / max(sims - 1, 1))[live].mean()) / 2,
            "how": f"{nl} positions {plies} plies into a game, same Dirichlet and exponential rows, full search with the direct-convolution f32 tower "
                   f"(az_tower_f32.hip) and with the {other} tower ({'az_tower.hip' if other == 'bf16' else 'az_tower_h2.hip' if other == 'f16x2' else 'az_tower_w.hip' if other == 'f32w' else 'az_tower_h1.hip'}); mean_abs_pi_difference = total variation "
                   "distance between the two root visit distributions"}


# ------------------------------------------------------------------------------ N ranks --------
def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list, dry_run: bool) -> int:
    """`python bench.py --gpus N` without a launcher around it: start N rank processes of this script, one per GPU (the
    reference's scale-out is N spawned self-play workers, training/alphazero_trainer.py:395-456, workers :420-433; here a
    worker owns a GPU).  Runs BEFORE anything imports torch: this parent never touches a GPU, never execs, and never
    restarts a child.  Rank 0's stdout (the JSON line) is relayed; the exit code is the worst of the children's; when a
    rank dies the others are ended (by their own PIDs) instead of waiting in a collective for it."""
    import subprocess
    import threading

    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AZ_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        capture = r == 0 or dry_run
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if capture else subprocess.DEVNULL, text=True))
    lines = [[] for _ in range(n)]

    def relay(r):
        for line in procs[r].stdout:
            lines[r].append(line)
            if not dry_run:
                sys.stdout.write(line)
                sys.stdout.flush()

    readers = [threading.Thread(target=relay, args=(r,), daemon=True) for r in range(n) if procs[r].stdout is not None]
    for t in readers:
        t.start()
    worst = 0
    live = set(range(n))
    while live:
        time.sleep(0.2)
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                worst = worst or rc
                print(f"[bench] rank {r} exited with code {rc}: ending the other ranks", file=sys.stderr, flush=True)
                for o in sorted(live):
                    procs[o].terminate()
    for t in readers:
        t.join(timeout=5.0)
    if dry_run:  # one JSON list: what every rank saw
        print(json.dumps([json.loads(l[-1]) for l in lines if l]))
    return worst


class Dist:
    """the few collectives the bench needs, on RCCL (backend "nccl") or -- for a 2-rank rehearsal on a one-GPU box, where
    RCCL refuses two ranks on one device -- on gloo through host tensors"""

    def __init__(self, backend, rank, world, dev):
        import torch.distributed as dist

        self.dist, self.backend, self.rank, self.world, self.dev = dist, backend, rank, world, dev
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier(self):
        self.dist.barrier()

    def all_reduce(self, t, op):
        """t: a device tensor; reduced in place over the ranks"""
        if self.backend == "nccl":
            self.dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=op)
            t.copy_(h)
        return t

    def gather_objects(self, obj):
        out = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="gomoku15", choices=sorted(WORKLOADS))
    ap.add_argument("--slots", type=int, default=0)
    ap.add_argument("--sims", type=int, default=0)
    ap.add_argument("--dtype", default="auto", choices=["auto", "bf16", "f16", "f32", "f32w", "f16x2", "f16r", "bf16r"],
                    help="network arithmetic of the PRIMARY line.  auto = f32, the reference's own precision (nn_models/base.py:17-19): the "
                         "hand-written f32 kernels for the GomokuModel / QuickModel families and the two small nets, the module's own PyTorch forward otherwise")
    ap.add_argument("--secondary-steps", type=int, default=-1, help="timed steps of the second line at the other precision (bf16 "
                    "throughput mode beside an f32 primary), reported as the object 'throughput_mode'; -1 = min(steps, 4) at one GPU, 0 = skip")
    ap.add_argument("--evaluator", default="auto", choices=["auto", "fused", "model", "hash"],
                    help="fused = hand-written MFMA conv tower (az_tower.hip / az_tower_f32.hip); model = the module's own PyTorch forward; "
                         "auto = fused when a hand-written kernel covers the network at this precision")
    ap.add_argument("--graph", type=int, default=-1, help="-1 (default): 1 for the workloads whose configuration has at most 4,096 games (gomoku9, dab, tictactoe), 0 otherwise.  "
                    "1: the timed steps replay the simulation step as a captured HIP graph "
                    "(what the small, launch-bound games want); the per-kernel HIP-event times then come from one eager probe "
                    "step run after the timed region.  0 (default): eager launches, events inside the timed region")
    ap.add_argument("--graph-unit", type=int, default=0, help="simulation steps captured per HIP graph (one replay = that many steps); "
                    "0 (default) = the whole search of a move")
    ap.add_argument("--leaves", type=int, default=1, help="leaves per tree per network batch: > 1 = virtual loss / in-tree leaf "
                    "batching (az_select_wave / az_expand_wave), which is not the reference's sequential search; reported without roofline")
    ap.add_argument("--preroll", type=int, default=96, help="cheap 2-sim move rounds that stagger game ages")
    ap.add_argument("--game-probe", type=int, default=1024, help="games played start to finish (full search, outside the timed "
                    "region, rank 0) to measure the mean game length that converts plies/s into games/s; 0 = skip")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-agreement", action="store_true", help="skip the f32-vs-bf16 decision-agreement measurement")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank: exercises the "
                    "barrier / all-reduce / weight-broadcast path on a one-GPU box")
    ap.add_argument("--prior-sharpness", type=float, default=1.0, help="multiply the network's logits by this (folded into the policy head's "
                    "last Linear layer): > 1 peaks the priors and deepens the trees the way a trained network does (a random-init network: "
                    "mean depth 1.5).  Diagnostic for the tree kernels; reported in config.prior_sharpness")
    ap.add_argument("--hash-scale", type=float, default=6.0, help="--evaluator hash: the synthetic evaluator's logits span +-scale/2 "
                    "(6: mean depth ~2.3 at 200 sims; 40-100: the 4-8 levels deep trees of a trained network)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI, one rank per GPU (the measured "
                    "configuration).  gloo: the same N-rank code path with the collectives on host tensors, ranks may share a GPU "
                    "(local rank modulo the visible GPUs): a rehearsal of the N > 1 logic on a one-GPU box, not a measurement")
    ap.add_argument("--launch-dry-run", action="store_true", help="every rank prints the RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* it "
                    "was started with and exits before anything touches a GPU (test of the launcher)")
    args = ap.parse_args()
    if args.graph < 0:
        args.graph = 1 if WORKLOADS[args.workload][3] <= 4096 and args.leaves <= 1 and args.evaluator != "hash" else 0
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    global PRIOR_SHARPNESS
    PRIOR_SHARPNESS = float(args.prior_sharpness)

    # `python bench.py --gpus N` with no launcher around it: become the launcher (before torch is imported; this process
    # never touches a GPU) and let N children of this script be the ranks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.launch_dry_run))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or under "
                 f"torch.distributed.run with --nproc-per-node equal to --gpus")
    if args.launch_dry_run:
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "master_addr": os.environ.get("MASTER_ADDR"),
                          "master_port": os.environ.get("MASTER_PORT"), "pid": os.getpid(), "torch_imported": "torch" in sys.modules}))
        return

    # a line on stderr every 45 s: the deep configurations run minutes between two outputs, and a harness that watches for
    # silence would take them for hung
    import threading
    t_start = time.perf_counter()

    def _heartbeat():
        while True:
            time.sleep(45.0)
            print(f"[bench] rank {rank} running, {time.perf_counter() - t_start:.0f} s", file=sys.stderr, flush=True)

    if rank == 0:
        threading.Thread(target=_heartbeat, daemon=True).start()

    cpu = None
    if rank == 0 and not args.no_cpu_baseline and args.cpu_baseline_seconds > 0:
        # before anything touches the GPU (the workers are spawned processes); with several ranks the others wait in the
        # rendezvous below meanwhile, so the sample is kept short
        cpu = cpu_baseline(args.workload, args.cpu_baseline_seconds if world == 1 else min(args.cpu_baseline_seconds, 6.0))

    import torch

    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        sys.exit("bench.py: no GPU visible (there is no CPU path)")
    if local_rank >= n_dev and args.dist_backend == "nccl":
        sys.exit(f"bench.py: local rank {local_rank} has no GPU of its own ({n_dev} visible): RCCL needs one GPU per rank "
                 "(--dist-backend gloo rehearses the N-rank path with ranks sharing a GPU)")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dst = Dist(args.dist_backend, rank, world, dev) if (world > 1 or args.force_dist) else None

    primary = args.dtype
    if primary == "auto":
        # f32, the reference's own arithmetic; every workload has an f32 evaluator (hand-written tower or the module's forward).  The
        # 64-channel GomokuModel networks (15x15 and 9x9) run on the Winograd f32 tower (az_tower_w.hip: what SelfPlayRunner(fused=True) picks
        # for it on an f32 engine); --dtype f32 asks for the direct-convolution f32 tower (az_tower_f32.hip) by name
        primary = "f32w" if args.workload in ("gomoku15", "gomoku15_deep64", "gomoku9") and args.evaluator in ("auto", "fused") else "f32"
    out = measure(args, primary, args.steps, args.warmup, dev, rank, world, dst)
    f32_family = primary in ("f32", "f32w")
    sec_steps = args.secondary_steps if args.secondary_steps >= 0 else (min(args.steps, 4) if world == 1 else 0)
    second = second_f16 = None
    row_tile = False
    direct = None
    if primary == "f32w" and sec_steps > 0 and args.leaves <= 1:  # the same workload on the direct-convolution f32 tower, for comparison
        direct = measure(args, "f32", sec_steps, 1, dev, rank, world, dst, with_probe=False)
    if f32_family and sec_steps > 0 and WORKLOADS[args.workload][0] == 2 and args.evaluator in ("auto", "fused") and args.leaves <= 1:
        # the 15x15 / 64-channel GomokuModel network: the row-tile tower (az_tower_h1.hip), bf16 operands like az_tower.hip's
        row_tile = args.workload in ("gomoku15", "gomoku15_deep64")
        second = measure(args, "bf16r" if row_tile else "bf16", sec_steps, 1, dev, rank, world, dst)
        second_f16 = measure(args, "f16r", sec_steps, 1, dev, rank, world, dst, with_probe=False) if row_tile else None
    third = None
    if f32_family and sec_steps > 0 and args.workload in ("gomoku15", "gomoku9", "gomoku15_deep64") and args.evaluator in ("auto", "fused") \
            and args.leaves <= 1:
        third = measure(args, "f16x2", sec_steps, 1, dev, rank, world, dst, with_probe=False)
    ranks_seen = dst.gather_objects({"rank": rank, "device": dev_index, "pid": os.getpid()}) if dst is not None else None
    if rank == 0:
        out["cpu_baseline"] = cpu
        if cpu:
            cpu["network_precision"] = "f32 (torch-CPU forward of the same network, as the reference runs it)"
            out["speedup_vs_cpu_baseline"] = out["value"] / cpu["value"]
        if primary == "f32w":
            out["network_algorithm"] = ("f32 operands, products and sums (the reference's arithmetic type, nn_models/base.py:17-19); the 3x3 convolutions by the "
                                        "Winograd transform F(2x2,3x3) on f32 MFMAs (az_tower_w.hip) -- the algorithm GPU libraries choose for f32 3x3 "
                                        "convolutions; against an f64 evaluation of the network its logits are as close as the direct-convolution tower's "
                                        "(tests/test_gpu_tower_w.py); f32_direct_mode = the same workload on the direct-convolution f32 tower")
        if direct is not None:
            rd = direct["roofline"]
            out["f32_direct_mode"] = {
                "value": direct["value"], "unit": "sims/s", "ms_per_step": direct["ms_per_step"], "dtype": "f32",
                "roofline": {k: rd[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "ms_per_launch", "boards_per_launch")},
                "note": "az_tower_f32.hip: the same f32 network by direct convolution (rounds 1-2's primary line)",
                "speedup_of_primary": out["value"] / direct["value"],
                "decision_agreement": None if args.no_agreement else decision_agreement(args, dev, other="f32w")}
        if second is not None:
            second["note"] = ("the same workload with the bf16 tower (bf16 operands, f32 accumulation): NOT the reference's precision; "
                              "decisions differ from the f32 search as measured in decision_agreement")
            if cpu:
                second["speedup_vs_cpu_baseline_f32_network"] = second["value"] / cpu["value"]
            if not args.no_agreement:
                second["decision_agreement"] = decision_agreement(args, dev, other="bf16r" if row_tile else "bf16")
            if second_f16 is not None:  # the same kernel on f16 operands (exact power-of-two scalings): 8x the operand precision
                r16 = second_f16["roofline"]
                second["f16_operands"] = {
                    "value": second_f16["value"], "unit": "sims/s", "ms_per_step": second_f16["ms_per_step"], "dtype": "f16",
                    "roofline": {k: r16[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "ms_per_launch", "boards_per_launch")},
                    "note": "az_tower_h1.hip with f16 operands (activations x 16, weights x a power of two per layer, f32 accumulation) "
                            "and the f32 heads: the throughput mode for callers who want its decisions closer to the f32 search's",
                    "decision_agreement": None if args.no_agreement else decision_agreement(args, dev, other="f16r")}
            out["throughput_mode"] = second
        if third is not None:
            third["note"] = ("EXPLORATORY, secondary: the same workload with az_tower_h2.hip -- every f32 operand of the convolutions split into two f16 "
                             "halves, three f16 MFMAs per product, f32 accumulation: f32-LEVEL ACCURACY (tests/test_gpu_tower_h2.py holds it to the f32 "
                             "tower's bars against an f64 evaluation), but not the reference's arithmetic and not bit-equal to the f32 tower: never the headline")
            third["speedup_vs_f32_line"] = third["value"] / out["value"]
            if cpu:
                third["speedup_vs_cpu_baseline_f32_network"] = third["value"] / cpu["value"]
            if not args.no_agreement:
                third["decision_agreement"] = decision_agreement(args, dev, other="f16x2")
            out["f32_accurate_mode"] = third
        if second is None and world > 1:
            out["throughput_mode"] = None
            out["throughput_mode_why"] = ("the bf16 line, the decision-agreement measurement and the game-length probe of the other "
                                          "precision run at one GPU only (--secondary-steps K forces the bf16 line at N > 1)")
        if dst is not None:  # who really took part: one rank per entry, with the device ordinal it ran on
            out["rccl_ranks"] = len(ranks_seen)
            out["dist_backend"] = "rccl (torch.distributed 'nccl')" if args.dist_backend == "nccl" else "gloo (host tensors; rehearsal, ranks may share a GPU)"
            out["devices"] = [r["device"] for r in sorted(ranks_seen, key=lambda r: r["rank"])]
            out["launched_by"] = "bench.py --gpus N (own launcher)" if os.environ.get("AZ_BENCH_LAUNCHED") else "external launcher (torch.distributed.run)"
        print(json.dumps(out), flush=True)
    if dst is not None:
        dst.close()  # rank 0 arrives last (game-length probe, printing): every rank leaves the group together


if __name__ == "__main__":
    main()
