"""Where the host's time per training step goes: cProfile over bench.py's timed steps (run on the GPU box)."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-parity", "--no-torch-adam", "--no-probe"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
