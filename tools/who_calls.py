"""Developer aid: which Python lines call torch.zeros / full / ones_like / zeros_like / empty during a bench step.
usage (GPU box): python tools/who_calls.py --workload train --n-rand 384 --steps 3 --warmup 2 --extra off --cpu-rays 0"""
import collections, os, runpy, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
counts = collections.Counter()
def wrap(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        st = [f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in traceback.extract_stack()[:-1] if ROOT in fr.filename and "who_calls" not in fr.filename]
        counts[(name, " <- ".join(st[-3:]))] += 1
        return orig(*a, **k)
    setattr(mod, name, f)
for n in ("zeros", "full", "ones_like", "zeros_like", "cat", "stack", "tensor", "as_tensor", "clone", "empty_like"):
    wrap(torch, n)
for n in ("copy_", "clone", "contiguous", "to", "float", "detach", "index_select", "mul", "__mul__", "__getitem__"):
    wrap(torch.Tensor, n)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    for (n, st), c in sorted(counts.items(), key=lambda kv: -kv[1]):
        sys.stderr.write(f"{c:5d}  torch.{n}  {st}\n")
