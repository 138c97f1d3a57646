"""A/B of two builds of libru3d.so on the same box: python tools/ab.py libA.so libB.so [reps]  (child processes, alternating)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a, b = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "3d-unet-renal-anatomy-extraction_amd"))
import _native as N, _ops as ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
out = []
for n, cin, cout, s in [(2, 32, 32, 128), (2, 64, 64, 64), (2, 128, 128, 32)]:
    x = torch.randn(n, s, s, s, cin, device=dev).bfloat16().permute(0, 4, 1, 2, 3)
    dy = torch.randn(n, s, s, s, cout, device=dev).bfloat16().permute(0, 4, 1, 2, 3)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    pw = ops.pack_weight(w, N.ROLE_CONV_FWD, torch.bfloat16, 1)
    out.append("%%d->%%d@%%d fwd %%.4f fwd_in %%.4f wgrad %%.4f" %% (cin, cout, s, timeit(lambda: ops.conv_fwd(x, pw, None, cout, 3, 1)),
               timeit(lambda: ops.conv_fwd_in(x, pw, None, cout, 3, 1)), timeit(lambda: ops.conv_wgrad(x, dy, 3, 1))))
print(" | ".join(out))
''' % (ROOT, ROOT)
for r in range(reps):
    for tag, lib in (("A", a), ("B", b)):
        env = dict(os.environ, RU3D_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(tag, p.stdout.strip() or p.stderr.strip()[-300:], flush=True)
