"""CPU tests of the host logic and of the C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blob_matches_reference_file_when_present(blob):
    ref = "/root/reference/rna_turner1999.par"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present on this box")
    from desirna_amd import params
    assert (params.load_par_file(ref) == blob).all()


def test_blob_spot_values(blob):
    # stack[CG][CG] = -240, stack[GC][GC] = -340 (rna_turner1999.par '# stack')
    stack = blob[3:3 + 64].reshape(8, 8)
    assert stack[1, 1] == -240 and stack[2, 2] == -340 and stack[1, 2] == -330


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from desirna_amd import engine
    hdr = open(os.path.join(ROOT, "include", "desirna_amd.h")).read()
    declared = set(re.findall(r"\b(drna_[a-z_]+)\s*\(", hdr))
    assert declared == set(engine.EXPORTS)
    lib = ctypes.CDLL(engine.LIB_PATH)
    for name in declared:
        assert getattr(lib, name) is not None


def test_simscore_mirror_matches_oracle_and_goldens(oracle, traj_golden, example_inputs):
    from desirna_amd.sim_score import SimScore
    for r in traj_golden[::7]:
        tgt = example_inputs[r["run"]]["sec_struct"][0].replace("&", "Ee")
        q = r["mfe_ss"].replace("&", "Ee")
        s = SimScore(tgt, q)
        s.find_basepairs()
        s.cofusion_matrix()
        (mcc, rec, prec), conf = oracle.simscore(tgt, q)
        assert s.conf_mat == conf
        assert (s.mcc(), s.recall(), s.precision()) == (mcc, rec, prec)
        assert 1 - s.mcc() == float(r["one_minus_mcc"])


def test_simscore_all_unpaired_special_case():
    from desirna_amd.sim_score import SimScore
    s = SimScore("....", "....")
    s.find_basepairs()
    s.cofusion_matrix()
    assert s.mcc() == 1.0 and s.recall() == 0.0 and s.precision() == 0.0


def test_parse_scoring_functions_first_term_quirk():
    from desirna_amd.energy_scores import parse_scoring_functions
    assert parse_scoring_functions("Ed-Epf:0.9,1-MCC:0.05,Edef:0.01") == [("Ed-Epf", 0.9)]
    assert parse_scoring_functions("Ed-Epf:0.9,1-MCC:0.05", first_term_only=False) == [("Ed-Epf", 0.9), ("1-MCC", 0.05)]
    with pytest.raises(ValueError):
        parse_scoring_functions("Ed-Epf")


def test_engine_fails_loudly_without_library(tmp_path):
    from desirna_amd import engine
    with pytest.raises(FileNotFoundError):
        engine.load_library(str(tmp_path / "nope.so"))


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without WORLD_SIZE starts N ranks itself (torch.distributed.run's environment contract),
    relays rank 0's line and fails when a rank fails; with WORLD_SIZE set (the driver's torchrun launch) it starts nothing."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DRNA_BENCH_ECHO_RANK"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["RANK"] == "0" and line["WORLD_SIZE"] == "3" and line["MASTER_ADDR"] == "127.0.0.1" and line["gpus"] == 3
    ranks = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(3)]
    assert [x["LOCAL_RANK"] for x in ranks] == ["0", "1", "2"] and len({x["MASTER_PORT"] for x in ranks}) == 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, DRNA_BENCH_ECHO_FAIL="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "ranks failed" in r.stderr
    for f in tmp_path.iterdir():
        f.unlink()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"],
                       env=dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and sorted(x.name for x in tmp_path.iterdir()) == ["rank1.json"]      # one process: this rank only


def test_bench_launcher_stops_the_other_ranks_when_one_dies(tmp_path):
    """A rank that exits non-zero at start-up must not leave the others in the rendezvous: the launcher ends them at once and
    reports which rank failed (advisor r3: the parent used to block on rank 0's stdout until the backend's own timeout)."""
    import os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DRNA_BENCH_ECHO_RANK=str(tmp_path), DRNA_BENCH_ECHO_FAIL="1", DRNA_BENCH_ECHO_HANG="0")     # rank 1 dies, rank 0 would sit for 10 minutes
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "ranks failed" in r.stderr and "rank 1 exited with code 3" in r.stderr
    assert time.time() - t0 < 60
    # ... and a deadline ends ranks that all hang
    env.update(DRNA_BENCH_ECHO_FAIL="-", DRNA_BENCH_ECHO_HANG="1", DRNA_BENCH_LAUNCH_TIMEOUT="3")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "deadline reached" in r.stderr and time.time() - t0 < 60


def test_bench_strong_scaling_shards(tmp_path):
    """--scaling strong: --replicas is the TOTAL (the reference's R), replica r on rank r mod N: the shards
    partition the replicas, their sizes differ by at most one, and every launched rank reports its own share."""
    import json, os, subprocess, sys
    import bench
    from desirna_amd import replica_exchange as rx
    for R, N in ((64, 1), (64, 2), (64, 4), (64, 8), (10, 4), (7, 8)):
        shards = [bench.strong_shard(R, k, N) for k in range(N)]
        assert sorted(x for sh in shards for x in sh) == list(range(R))
        assert max(len(sh) for sh in shards) - min(len(sh) for sh in shards) <= 1
        assert shards == [rx.ReplicaShards(R, k, N).local for k in range(N)]        # the design driver's sharding
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["DRNA_BENCH_ECHO_RANK"] = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--scaling", "strong", "--replicas", "64"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(3)]
    assert [g["local_replicas"] for g in got] == [22, 21, 21] and all(g["scaling"] == "strong" for g in got)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["local_replicas"] == 64      # weak: --replicas per GPU


def test_tile_product_schedule_never_reads_an_unfinished_operand():
    """fold_pf_strip.hpp deals the chunks of a tile's far range to the 16 steps before the tile is due, from the middle outward
    (tools/pkt_schedule.py restates the plan): every chunk exactly once, and only when both operands are final and visible
    (diagonals <= step - 2) -- for the shipped constants and the smallest lead that works; one diagonal less must fail."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pkt_schedule", os.path.join(ROOT, "tools", "pkt_schedule.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    src = open(os.path.join(ROOT, "desirna_amd", "csrc", "fold_pf_strip.hpp")).read()
    import re
    W = int(re.search(r"#define DRNA_PKT_W (\d+)", src).group(1)); L = int(re.search(r"#define DRNA_PKT_L (\d+)", src).group(1))
    assert m.check(W, L) == 0 and m.check(W, 3) == 0 and m.check(8, 3) == 0
    assert m.check(W, 2) > 0


def test_tile_geometry_of_the_partition_function_counts_every_split_point_once():
    """fold_pf_lds.hpp cuts a cell's multiloop split points into the far range of its 4 x 4 tile (one matrix-product chain, by the
    helper workgroup or by the main workgroup's sweep waves) and at most PKE + 3 near ones at either end (tools/kt_geometry.py
    restates the index arithmetic): every split point exactly once, the near slots fit their four rows, a tile's operands are
    final when its first step comes, and the helper's flag covers the cells the main workgroup reads after it -- for the shipped
    slack and for lengths that leave ragged tiles; a far range that starts one split point early must be caught."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kt_geometry", os.path.join(ROOT, "tools", "kt_geometry.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    src = open(os.path.join(ROOT, "desirna_amd", "csrc", "fold_pf_lds.hpp")).read()
    pke = int(re.search(r"#define DRNA_PKE (\d+)", src).group(1))
    assert pke >= 4
    assert "const int c = a + B, m_lo = 4 * a + 9 + PKE, m_hi = 4 * c - 3 - PKE;" in src          # what far_range() restates
    assert "nl = 4 * a + 4 + PKE - i; nh = j - 4 * c - 1 + PKE;" in src                            # ... and near_counts()
    for n in (200, 199, 64, 37, 21, 9):
        assert m.check(n, pke) == 0, n
    good = m.far_range
    m.far_range = lambda a, B, p: (good(a, B, p)[0] - 1, good(a, B, p)[1])
    assert m.check(64, pke) > 0
    m.far_range = good
    assert m.check(64, 3) > 0          # less slack than the main workgroup's four-step window needs
