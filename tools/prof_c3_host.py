import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(sys.path[0], "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
dev = torch.device("cuda", 0)
b.march_composite_rate(dev, iters=5)
pr = cProfile.Profile(); pr.enable()
r = b.march_composite_rate(dev, iters=200)
pr.disable()
print(r)
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
