#!/usr/bin/env python3
"""Where the HOST time of a training step goes (cProfile over bench.py's own step loop): tools/host_profile.py [bench.py flags]
The device runs behind the host at 384 rays per rank only if the host enqueues a step faster than the GPU executes it."""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"] + sys.argv[1:]
import bench  # noqa: E402

if os.environ.get("ANERF_PROFILE_SINGLE_THREAD") == "1":
    # run the autograd engine on the calling thread: the backward's Python (the custom Functions' backward bodies, the
    # data-parallel hooks) then shows up in this profile instead of hiding inside `run_backward`
    import torch
    torch.autograd.set_multithreading_enabled(False)

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
sys.stderr.write(s.getvalue())
