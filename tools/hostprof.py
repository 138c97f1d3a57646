"""Host-side profile of the training step (GPU box): where does the Python time of one step go?
    python tools/hostprof.py [steps]"""
import cProfile, io, os, pstats, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["bench.py", "--steps", sys.argv[1] if len(sys.argv) > 1 else "10", "--warmup", "3", "--no-cpu-baseline", "--no-probe"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
