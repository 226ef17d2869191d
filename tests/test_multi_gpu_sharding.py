"""The N>1 path on CPU: two ranks over gloo shard the chains, evaluate their own range (here with
the CPU oracle standing in for the device), and the all-gather of per-chain summary records
reproduces the single-process result on every rank."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_chains, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mmid_amd_loader
    mm = mmid_amd_loader.load()
    from mmid_amd import draws, parallel
    import oracle_py
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pb = mm.SEPAIHRDProblem.load(os.path.join(ROOT, "tests", "golden", "reference_test_fixture.json"))
    pb.constraint_mode = mm.CONSTRAINT_REFLECT
    lo, hi = parallel.shard_range(n_chains, rank, world)
    # chain c draws from mt19937(1 + c) wherever it lives
    theta = draws.jitter_draws(pb, parallel.chain_seed(1, lo), hi - lo)
    ll = oracle_py.Oracle(pb).eval_batch(theta, nthreads=1)["loglik"]
    rec = parallel.summary_record(theta[:, None, :], ll[:, None], np.zeros(hi - lo))
    gathered = parallel.all_gather_records(rec, n_chains)
    q = parallel.ensemble_quantiles(gathered)
    np.save(os.path.join(out_dir, f"gathered_{rank}.npy"), gathered.numpy())
    np.save(os.path.join(out_dir, f"quant_{rank}.npy"), q.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    sys.path.insert(0, ROOT)
    import mmid_amd_loader
    mmid_amd_loader.load()
    from mmid_amd import parallel
    for n, w in ((262144, 8), (11, 2), (5, 8), (4096, 3)):
        spans = [parallel.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_allgather_matches_single_process(tmp_path, mm, oracle_py):
    import torch.multiprocessing as mp
    from mmid_amd import draws, parallel
    n_chains, world = 11, 2   # ragged: 6 + 5
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n_chains, str(tmp_path)), nprocs=world, join=True)
    pb = mm.SEPAIHRDProblem.load(os.path.join(ROOT, "tests", "golden", "reference_test_fixture.json"))
    pb.constraint_mode = mm.CONSTRAINT_REFLECT
    theta = draws.jitter_draws(pb, 1, n_chains)
    ll = oracle_py.Oracle(pb).eval_batch(theta, nthreads=1)["loglik"]
    want = parallel.summary_record(theta[:, None, :], ll[:, None], np.zeros(n_chains))
    g0 = np.load(tmp_path / "gathered_0.npy")
    g1 = np.load(tmp_path / "gathered_1.npy")
    assert np.array_equal(g0, want) and np.array_equal(g1, want)
    assert np.array_equal(np.load(tmp_path / "quant_0.npy"), np.load(tmp_path / "quant_1.npy"))


@pytest.mark.gpu
def test_bench_multi_rank_path_end_to_end():
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank, --gpus 2
    --allgather), rehearsed on ONE GPU: SEPAIHRD_BENCH_REHEARSAL=1 puts both ranks on device 0 and swaps RCCL for
    gloo -- the sharded draws, barriers, max-over-ranks timing, the all-gather of the per-chain summary records
    and the rank-0 JSON line are the production code.  Not a measurement (the line says so)."""
    import json
    import subprocess
    env = dict(os.environ, SEPAIHRD_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--allgather", "--chains", "1500", "--sampler-iterations", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["rehearsal"] is True and out["cpu_baseline"] is None
    assert out["unit"] == "evals/s" and out["higher_is_better"] is True and out["dtype"] == "f64"
    # whole-job aggregate: both ranks' chains over the max-over-ranks time
    assert abs(out["value"] - 2 * 1500 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert out["status_counts"][0] == 1500
    ag = out["allgather"]
    assert ag["ranks"] == 2 and ag["blocks_in_rank_order"] is True and ag["bytes_per_rank"] == 1500 * (2 * 62 + 2) * 8
    assert out["roofline"]["bound"] == "fp64_valu" and 0 < out["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_chain_groups_on_distinct_devices(mm, oracle_py, shipped):
    """optimizeChainGroupsOnDevice with one objective per DEVICE (how the C++ sampler shards over the GPUs of a
    node: an objective is bound to a device at construction) == the single-device run.  Needs two visible GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the multi-device grouping needs two")
    pb = shipped.with_(arith=mm.ARITH_STRICT, constraint_mode=1)
    x0 = oracle_py.Oracle(pb).jitter_draws(pb.base_theta, 9, 7, mode=1)
    kw = dict(seed=23, iterations=120, burn_in=40, adaptation_period=30, thinning=5)
    one = mm.HostObjective(pb, device=0).metropolis_hastings(x0, device_state=True, **kw)
    objs = [mm.HostObjective(pb, device=d) for d in range(2)]
    grp = mm.hostabi.metropolis_hastings_groups(objs, x0, **kw)
    for k in ("accept_trace", "accepted", "best_value", "best"):
        assert np.array_equal(grp[k], one[k]), k
