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
