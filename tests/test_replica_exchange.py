"""Host-side replica exchange and the N>1 sharding path (gloo, world_size 2, CPU)."""
import math
import os
import random
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rep_temps_ladder():
    from desirna_amd.replica_exchange import get_rep_temps
    t = get_rep_temps(64, 10, 150)
    assert t[:3] == [10, 12.222, 14.444] and t[-2:] == [147.778, 150.0] and len(t) == 64
    assert get_rep_temps(1, 10, 150) == [150]


def test_mc_delta_rng_consumption():
    from desirna_amd.replica_exchange import mc_delta
    rng = random.Random(3)
    assert mc_delta(1.0, 0.5, 10.0, rng) == (True, True)
    assert rng.random() == random.Random(3).random()          # nothing drawn when the mutant is not worse
    rng = random.Random(3)
    ref = random.Random(3)
    acc, better = mc_delta(1.0, 1.01, 100.0, rng)
    assert better is False and acc == (math.exp(-504.12 / 100.0 * (1.01 - 1.0)) > ref.random())


def test_replica_exchange_swaps_only_temperatures():
    from desirna_amd.replica_exchange import get_rep_temps, replica_exchange
    R = 8
    temps = get_rep_temps(R, 10, 150)
    scores = [5.0, 4.0, 3.0, 2.0, 1.0, 0.5, 0.2, 0.1]         # hotter replicas are better: every pair swaps
    new, acc, better, rej = replica_exchange(temps, scores, global_step=1, rng=random.Random(0))
    assert sorted(new) == sorted(temps) and acc == 4 and better == 4 and rej == 0
    assert new[0] == temps[1] and new[1] == temps[0]
    new2, acc2, _, _ = replica_exchange(temps, scores, global_step=2, rng=random.Random(0))
    assert acc2 == 3 and new2[0] == temps[0] and new2[1] == temps[2] and new2[2] == temps[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_allgather_and_identical_swaps_gloo_world2(tmp_path):
    """Two ranks own interleaved replicas, all-gather their scores (gloo) and must take identical swap decisions."""
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import json, os, random, sys
        sys.path.insert(0, %r)
        import numpy as np
        import torch.distributed as dist
        from desirna_amd.replica_exchange import ReplicaShards, get_rep_temps, replica_exchange
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        R = 13
        sh = ReplicaShards(R, rank, world)
        truth = np.random.default_rng(7).normal(size=R)           # what a single process would have
        full = sh.allgather_scores(truth[sh.local])
        temps = get_rep_temps(R, 10, 150)
        rng = random.Random(2137)
        for step in range(1, 6):
            temps, acc, better, rej = replica_exchange(temps, list(full), step, rng)
        print(json.dumps({"rank": rank, "local": sh.local, "full": full.tolist(), "temps": temps}))
        dist.destroy_process_group()
    """ % ROOT))
    port = _free_port()
    procs = []
    for rk in range(2):
        env = dict(os.environ, RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0
        import json
        outs.append(json.loads(out.strip().splitlines()[-1]))
    truth = np.random.default_rng(7).normal(size=13)
    assert outs[0]["local"] == list(range(0, 13, 2)) and outs[1]["local"] == list(range(1, 13, 2))
    for o in outs:
        assert np.array_equal(np.array(o["full"]), truth)
    assert outs[0]["temps"] == outs[1]["temps"]


def test_sharded_design_run_gloo_world2(tmp_path):
    """The whole design loop with replicas sharded over two ranks (gloo, oracle-backed scorer): both ranks must
    agree on the temperature ladder after every exchange and together hold every replica exactly once."""
    script = tmp_path / "worker_design.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        from types import SimpleNamespace
        import torch.distributed as dist
        from desirna_amd import design, params
        from desirna_amd.energy_scores import parse_scoring_functions
        from desirna_amd.replica_exchange import ReplicaShards
        from oracle.pyoracle import Oracle
        from tests.test_design_driver import OracleScorer, ETE1
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        inp = SimpleNamespace(name="e1", sec_struct=ETE1, seq_restr="N" * 16, seed_seq=None, alt_sec_struct=None, alt_sec_structs=None)
        sc = OracleScorer(Oracle(params.load_blob()), ETE1, parse_scoring_functions("Ed-Epf:1.0"))
        res = design.run_design(inp, replicas=6, exchange=10, steps=3, seed=11, scorer=sc, shards=ReplicaShards(6, rank, world))
        print(json.dumps({"rank": rank, "reps": [s.replica_num for s in res["replicas"]],
                          "temps": [s.temp_shelf for s in res["replicas"]], "scored": res["stats"]["scored"],
                          "acc_re": res["stats"]["acc_re"], "rej_re": res["stats"]["rej_re"]}))
        dist.destroy_process_group()
    """ % ROOT))
    port = _free_port()
    procs = []
    for rk in range(2):
        env = dict(os.environ, RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    import json
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0
        outs.append(json.loads(out.strip().splitlines()[-1]))
    assert sorted(outs[0]["reps"] + outs[1]["reps"]) == [1, 2, 3, 4, 5, 6]
    from desirna_amd.replica_exchange import get_rep_temps
    assert sorted(outs[0]["temps"] + outs[1]["temps"]) == sorted(get_rep_temps(6, 10.0, 150.0))
    assert (outs[0]["acc_re"], outs[0]["rej_re"]) == (outs[1]["acc_re"], outs[1]["rej_re"])   # same swap decisions
    assert outs[0]["scored"] == outs[1]["scored"] == 3 + 3 * 10 * 3


def _run_ranks(script, world=2, timeout=900):
    import json
    port = _free_port()
    procs = []
    for rk in range(world):
        env = dict(os.environ, RANK=str(rk), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=timeout)
        assert p.returncode == 0, out
        outs.append(json.loads(out.strip().splitlines()[-1]))
    return outs


def test_sharded_unseeded_run_and_collective_stop_gloo_world2(tmp_path):
    """with the default seed (0 = unseeded) every rank used to seed its own main stream, and the loop exits
    (time limit, stop-when-solved) were rank-local, so one rank could leave while the other blocked in the all-gather.
    Now rank 0's seed is broadcast and the stop decision comes out of the gathered flags: both ranks end after the same
    number of exchange steps with the same ladder, although only one of them may hold the solved replica."""
    script = tmp_path / "worker_stop.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        from types import SimpleNamespace
        import torch.distributed as dist
        from desirna_amd import design, params
        from desirna_amd.energy_scores import parse_scoring_functions
        from desirna_amd.replica_exchange import ReplicaShards
        from oracle.pyoracle import Oracle
        from tests.test_design_driver import OracleScorer, ETE1
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        inp = SimpleNamespace(name="e1", sec_struct=ETE1, seq_restr="N" * 16, seed_seq=None, alt_sec_struct=None, alt_sec_structs=None)
        sc = OracleScorer(Oracle(params.load_blob()), ETE1, parse_scoring_functions("Ed-Epf:1.0"))
        res = design.run_design(inp, replicas=5, exchange=20, steps=40, seed=0, scorer=sc, stop_when_solved=True,
                                shards=ReplicaShards(5, rank, world))
        print(json.dumps({"rank": rank, "steps": res["steps"], "temps": res["temps"], "solved": res["solved"],
                          "best": res["best"].sequence, "first": res["simulation_data"][0]["sequence"]}))
        dist.destroy_process_group()
    """ % ROOT))
    outs = _run_ranks(script)
    assert outs[0]["steps"] == outs[1]["steps"] < 40 and outs[0]["solved"] and outs[1]["solved"]
    assert outs[0]["temps"] == outs[1]["temps"] and outs[0]["first"] == outs[1]["first"]       # same seed on both ranks
    assert outs[0]["best"] == outs[1]["best"]                                                  # best of the whole job


def test_sharded_fast_driver_equals_single_rank_gloo_world2(tmp_path, oracle):
    """run_design_fast with shards (native proposer / Metropolis with the reference's per-replica MT19937 streams, oracle-
    backed engine stub): two ranks together hold every replica once, end with the same ladder, and reproduce the records
    of the unsharded run replica by replica -- ONE all-gather per exchange step."""
    script = tmp_path / "worker_fast.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        from types import SimpleNamespace
        import torch.distributed as dist
        from desirna_amd import design, params
        from desirna_amd.replica_exchange import ReplicaShards
        from oracle.pyoracle import Oracle
        from tests.test_design_driver import OracleEngine, ETE1
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        calls = [0]
        orig = dist.all_gather_into_tensor
        def counted(*a, **k):
            calls[0] += 1
            return orig(*a, **k)
        dist.all_gather_into_tensor = counted
        inp = SimpleNamespace(name="e1", sec_struct=ETE1, seq_restr="N" * 16, seed_seq=None, alt_sec_struct=None, alt_sec_structs=None)
        res = design.run_design_fast(inp, replicas=7, exchange=12, steps=3, seed=21, engine=OracleEngine(Oracle(params.load_blob())),
                                     native_loop=False, shards=ReplicaShards(7, rank, world))
        print(json.dumps({"rank": rank, "local": res["local"], "temps": res["temps"], "gathers": calls[0],
                          "recs": [[r["replica_num"], r["sim_step"], r["sequence"], r["temp_shelf"]] for r in res["simulation_data"]],
                          "best": res["best"].sequence, "acc_re": res["stats"]["acc_re"]}))
        dist.destroy_process_group()
    """ % ROOT))
    outs = _run_ranks(script)
    from types import SimpleNamespace
    from desirna_amd import design
    from tests.test_design_driver import OracleEngine, ETE1
    inp = SimpleNamespace(name="e1", sec_struct=ETE1, seq_restr="N" * 16, seed_seq=None, alt_sec_struct=None, alt_sec_structs=None)
    one = design.run_design_fast(inp, replicas=7, exchange=12, steps=3, seed=21, engine=OracleEngine(oracle), native_loop=False)
    assert sorted(outs[0]["local"] + outs[1]["local"]) == list(range(7))
    assert outs[0]["temps"] == outs[1]["temps"] == one["temps"]
    assert outs[0]["gathers"] == outs[1]["gathers"] == 3 + 1          # one per exchange step (+ the start-up flag exchange)
    want = sorted([r["replica_num"], r["sim_step"], r["sequence"], r["temp_shelf"]] for r in one["simulation_data"])
    assert sorted(outs[0]["recs"] + outs[1]["recs"]) == want
    assert outs[0]["best"] == outs[1]["best"] == one["best"].sequence and outs[0]["acc_re"] == one["stats"]["acc_re"]


import pytest


@pytest.mark.gpu
def test_sharded_native_fast_driver_two_ranks_one_gpu(tmp_path, eterna_targets):
    """The production fast path (drna_mc_run: the whole inner loop native, scoring on the GPU) with the replicas of one
    design sharded over two ranks that share this box's single GPU (gloo rehearsal; RCCL needs one GPU per rank): union of
    replicas, identical ladders, one all-gather per exchange step, and the records / best of the unsharded run
    (reference utils/replica_exchange_monte_carlo.py:233-271, :113-173)."""
    tg = eterna_targets["eteV1_92.txt"]
    script = tmp_path / "worker_gpu.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        from types import SimpleNamespace
        import torch.distributed as dist
        from desirna_amd import design
        from desirna_amd.replica_exchange import ReplicaShards
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        calls = [0]
        orig = dist.all_gather_into_tensor
        def counted(*a, **k):
            calls[0] += 1
            return orig(*a, **k)
        dist.all_gather_into_tensor = counted
        tg = %r
        inp = SimpleNamespace(name="e92", sec_struct=tg, seq_restr="N" * len(tg), seed_seq=None, alt_sec_struct=None, alt_sec_structs=None)
        res = design.run_design_fast(inp, replicas=16, exchange=20, steps=3, seed=5, shards=ReplicaShards(16, rank, world))
        print(json.dumps({"rank": rank, "local": res["local"], "temps": res["temps"], "gathers": calls[0],
                          "recs": [[r["replica_num"], r["sim_step"], r["sequence"], r["temp_shelf"]] for r in res["simulation_data"]],
                          "best": res["best"].sequence, "acc_re": res["stats"]["acc_re"], "scored": res["stats"]["scored"]}))
        dist.destroy_process_group()
    """ % (ROOT, tg)))
    outs = _run_ranks(script)
    from types import SimpleNamespace
    from desirna_amd import design
    inp = SimpleNamespace(name="e92", sec_struct=tg, seq_restr="N" * len(tg), seed_seq=None, alt_sec_struct=None, alt_sec_structs=None)
    one = design.run_design_fast(inp, replicas=16, exchange=20, steps=3, seed=5)
    assert sorted(outs[0]["local"] + outs[1]["local"]) == list(range(16))
    assert outs[0]["temps"] == outs[1]["temps"] == one["temps"]
    assert outs[0]["gathers"] == outs[1]["gathers"] == 3 + 1
    want = sorted([r["replica_num"], r["sim_step"], r["sequence"], r["temp_shelf"]] for r in one["simulation_data"])
    assert sorted(outs[0]["recs"] + outs[1]["recs"]) == want
    assert outs[0]["best"] == outs[1]["best"] == one["best"].sequence
    assert outs[0]["scored"] + outs[1]["scored"] == one["stats"]["scored"] and outs[0]["acc_re"] == one["stats"]["acc_re"]


def test_shard_puzzles_balances_cubes():
    from desirna_amd.replica_exchange import shard_puzzles
    lengths = [400, 12, 36, 200, 104, 104, 380, 16, 90, 250]
    own = shard_puzzles(lengths, 4)
    load = [sum(l ** 3 for l, o in zip(lengths, own) if o == r) for r in range(4)]
    assert sorted(set(own)) == [0, 1, 2, 3]
    assert max(load) <= 400 ** 3 + 36 ** 3 + 16 ** 3 + 12 ** 3          # the longest puzzle sits (almost) alone
    assert shard_puzzles(lengths, 1) == [0] * len(lengths)


def test_puzzle_set_sharded_gloo_world2(tmp_path):
    """Config 4's multi-rank path on CPU: three small puzzles over two ranks (oracle-backed scorer), results gathered on both."""
    script = tmp_path / "worker_set.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        from types import SimpleNamespace
        import torch.distributed as dist
        from desirna_amd import design, params
        from desirna_amd.energy_scores import parse_scoring_functions
        from oracle.pyoracle import Oracle
        from tests.test_design_driver import OracleScorer
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        orc = Oracle(params.load_blob())
        targets = ["(((((......)))))", "((((....))))", "(((((((....)))))))..."]
        inputs = [SimpleNamespace(name="p%%d" %% k, sec_struct=t, seq_restr="N" * len(t), seed_seq=None, alt_sec_struct=None,
                                  alt_sec_structs=None) for k, t in enumerate(targets)]
        def driver(inp, **kw):
            sc = OracleScorer(orc, inp.sec_struct, parse_scoring_functions("Ed-Epf:1.0"))
            return design.run_design(inp, scorer=sc, **kw)
        res = design.run_puzzle_set(inputs, rank, world, driver=driver, replicas=4, exchange=10, steps=4, seed=5)
        print(json.dumps({"rank": rank, "res": {str(k): v for k, v in res.items()}}))
        dist.destroy_process_group()
    """ % ROOT))
    port = _free_port()
    procs = []
    for rk in range(2):
        env = dict(os.environ, RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    import json
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0
        outs.append(json.loads(out.strip().splitlines()[-1]))
    assert outs[0]["res"] == outs[1]["res"] and sorted(outs[0]["res"]) == ["0", "1", "2"]
    ranks = {v["rank"] for v in outs[0]["res"].values()}
    assert ranks == {0, 1}                                       # both ranks worked
    for k, v in outs[0]["res"].items():
        assert len(v["sequence"]) == len(v["mfe_ss"])
