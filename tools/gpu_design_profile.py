"""GPU box: where the end-to-end design step goes (native batched driver, L=200, R=64)."""
import cProfile, pstats, io, os, sys, time
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from desirna_amd import design
tg = bench.load_target("eteV1_69.txt")
inp = SimpleNamespace(name="ete69", sec_struct=tg, seq_restr="N" * len(tg), seed_seq=None, alt_sec_struct=None, alt_sec_structs=None)
design.run_design_fast(inp, replicas=64, exchange=20, steps=1, seed=1)     # warm-up
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
res = design.run_design_fast(inp, replicas=64, exchange=100, steps=5, seed=1)
pr.disable()
dt = time.time() - t0
print("scored/s", res["stats"]["scored"] / dt, "ms per MC iteration", 1e3 * dt / 500)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
print(s.getvalue())
