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
