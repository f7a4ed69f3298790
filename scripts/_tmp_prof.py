import cProfile, pstats, sys, os, io
sys.argv = ["x", "100000", "50", "480", "270"]
src = open(os.path.join(os.path.dirname(__file__), "model_step_bench.py")).read()
head = src.split("for _ in range(25):")[0]
g = {"__name__": "__main__", "__file__": os.path.join(os.path.dirname(__file__), "model_step_bench.py")}
exec(compile(head, "msb", "exec"), g)
import torch
step = g["step"]
for _ in range(30): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
