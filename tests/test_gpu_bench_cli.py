"""bench.py end to end at toy sizes: the JSON contract (metric / value / roofline / cpu_baseline / throughput_mode) and the
torch.distributed path (RCCL barrier, all-reduce of the timings, weight broadcast) with one rank under --force-dist."""
import json
import os
import socket
import subprocess
import sys

import pytest

from .conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--preroll", "4", "--game-probe", "16",
           "--cpu-baseline-seconds", "1", *extra]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_line_gomoku_f32_primary_with_bf16_beside_it_under_rccl():
    d = _run(["--workload", "gomoku9", "--slots", "256", "--sims", "24", "--force-dist"])
    assert d["metric"] == "mcts_sims_per_sec" and d["unit"] == "sims/s" and d["n_gpus"] == 1 and d["dtype"] == "f32"
    assert d["value"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # (the Winograd f32 tower: frac prices the network's algorithmic flops and may exceed 1; the executed MFMA flops sit beside it)
    assert r["bound"] == "mfma" and r["kernel"] == "k_tower_f32w" and r["peak"] == 157.3 and r["frac"] > 0 and 0 < r["executed"]["frac"] < 1
    assert d["f32_direct_mode"]["roofline"]["kernel"] == "k_tower_f32_t16" and 0 < d["f32_direct_mode"]["roofline"]["frac"] < 1
    assert d["roofline_tree"]["bound"] == "hbm" and d["roofline_tree"]["peak"] == 8000.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    t = d["throughput_mode"]
    assert t["dtype"] == "bf16" and t["roofline"]["kernel"] == "k_tower" and t["roofline"]["peak"] == 2500.0
    a = t["decision_agreement"]
    assert 0.5 <= a["same_most_visited_move"] <= 1.0 and a["positions"] > 0
    h = d["f32_accurate_mode"]  # the split-operand f16 tower: a third, secondary line
    assert h["dtype"] == "f16x2" and h["roofline"]["kernel"] == "k_tower_h2" and h["roofline"]["peak"] == 2500.0
    assert h["roofline"]["f32_equivalent_tflops"] > 0 and h["decision_agreement"]["same_most_visited_move"] >= 0.9


def test_bench_line_gomoku15_throughput_mode_is_the_row_tile_tower():
    d = _run(["--workload", "gomoku15", "--slots", "128", "--sims", "12", "--no-cpu-baseline"])
    # the primary line: f32 arithmetic on the Winograd tower (az_tower_w.hip); its roofline prices the network's algorithmic
    # (direct-convolution) flops and carries the executed MFMA flops beside them; the direct-convolution tower as a comparison object
    assert d["dtype"] == "f32" and d["roofline"]["kernel"] == "k_tower_f32w" and "az_tower_w.hip" in d["nn_eval"]["implementation"]
    assert 0 < d["roofline"]["executed"]["frac"] < 1 and d["roofline"]["executed"]["flops_per_launch"] < d["roofline"]["flops_per_launch"]
    fd = d["f32_direct_mode"]
    assert fd["dtype"] == "f32" and fd["roofline"]["kernel"] == "k_tower_f32_t16" and fd["value"] > 0
    assert fd["decision_agreement"]["same_most_visited_move"] >= 0.9
    dd = _run(["--workload", "gomoku15", "--slots", "128", "--sims", "12", "--no-cpu-baseline", "--dtype", "f32", "--secondary-steps", "0"])
    assert dd["dtype"] == "f32" and dd["roofline"]["kernel"] == "k_tower_f32_t16" and "f32_direct_mode" not in dd
    t = d["throughput_mode"]
    assert t["dtype"] == "bf16" and t["roofline"]["kernel"] == "k_tower_h1" and "az_tower_h1.hip" in t["nn_eval"]["implementation"]
    assert t["decision_agreement"]["same_most_visited_move"] >= 0.5
    f = t["f16_operands"]  # the same kernel on f16 operands: closer to the f32 search
    assert f["dtype"] == "f16" and f["value"] > 0 and f["roofline"]["kernel"] == "k_tower_h1"
    assert f["decision_agreement"]["same_most_visited_move"] >= t["decision_agreement"]["same_most_visited_move"] - 0.05


def test_bench_line_small_game_fused_and_through_pytorch():
    base = ["--workload", "dab", "--slots", "128", "--sims", "16", "--graph", "1", "--no-cpu-baseline"]
    d = _run(base)  # auto: the one-kernel f32 network of az_smallnet.hip
    assert d["dtype"] == "f32" and d["value"] > 0 and d["cpu_baseline"] is None and "throughput_mode" not in d
    assert "evaluator=fused" in d["config"]["workload"] and "k_smallnet_f32" in d["nn_eval"]["implementation"]
    kernels = {d["roofline"]["kernel"], d["nn_eval"].get("roofline_network_kernel", {}).get("kernel")}
    assert "k_smallnet_f32" in kernels and d["roofline_tree"]["kernel"] == "k_expand_select"
    d = _run(base + ["--evaluator", "model"])  # the module's own PyTorch forward: the tree kernel's HBM roofline leads
    assert d["roofline"]["kernel"] == "k_expand_select" and "PyTorch" in d["nn_eval"]["implementation"]


def test_bench_gpus_2_launches_two_ranks_gloo_rehearsal():
    """`python bench.py --gpus 2` as the driver calls it: the script launches its own two ranks.  On a one-GPU box RCCL refuses
    two ranks on one device, so the collectives run on gloo (--dist-backend gloo) with both ranks sharing cuda:0: this runs
    measure()'s N > 1 branches -- barrier pair, MAX / SUM all-reduce of the timings, weight broadcast + refresh of the fused
    tower before every move, rank-0-only game-length probe while rank 1 waits, the final barrier -- with world = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--slots", "128", "--sims", "16",
           "--steps", "2", "--warmup", "1", "--preroll", "4", "--game-probe", "16", "--cpu-baseline-seconds", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["devices"] == [0, 0] and "gloo" in d["dist_backend"]
    assert d["launched_by"].startswith("bench.py")
    assert d["value"] > 0 and d["scaling"] == "weak" and d["dtype"] == "f32" and d["cpu_baseline"]["value"] > 0
    # whole-job aggregate: both ranks' simulations over the slowest rank's time
    per_rank_sims = 128 * 16 * 2
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - 2 * per_rank_sims) <= 0.02 * 2 * per_rank_sims
    assert d["throughput_mode"] is None and "one GPU" in d["throughput_mode_why"]


def test_bench_under_torch_distributed_run_two_ranks_gloo_rehearsal():
    """The driver's launch line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with two ranks: bench.py takes the ranks it is given (no launcher of its own), rank 0
    prints the one JSON line.  (gloo rehearsal on the one-GPU box, as above; the primary line is the Winograd f32 tower.)"""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--slots", "128",
           "--sims", "16", "--steps", "2", "--warmup", "1", "--preroll", "4", "--game-probe", "16", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["devices"] == [0, 0] and d["launched_by"].startswith("external")
    assert d["value"] > 0 and d["dtype"] == "f32" and d["roofline"]["kernel"] == "k_tower_f32w" and d["throughput_mode"] is None
    # and a world size that differs from --gpus is refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
