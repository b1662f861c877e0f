"""Which event queries does the HIP runtime refuse while a stream capture is open?  (The process group's watchdog thread died of one:
profiles/r06_watchdog_vs_global_capture.txt.)  For capture mode in {global, thread_local}: events recorded BEFORE the capture on
(a) an unrelated stream, (b) a side stream that later joins the capture, (c) the capturing stream itself -- each queried from a
second thread and from the capturing thread while the capture is open.  Run on the GPU box."""
import threading
import time

import torch

dev = torch.device("cuda", 0)
x = torch.zeros(1024, device=dev)


def attempt(fn):
    try:
        fn()
        return "ok"
    except RuntimeError as e:
        return "REFUSED (" + str(e).split("\n")[0][:70] + ")"


for mode in ("global", "thread_local"):
    main, side, other = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    ev = {}
    for name, s in (("unrelated stream", other), ("side stream that joins the capture", side), ("the capturing stream", main)):
        with torch.cuda.stream(s):
            x.add_(1)
            e = torch.cuda.Event()
            e.record()
            ev[name] = e
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    res = {}
    with torch.cuda.stream(main):
        ctx = torch.cuda.graph(g, stream=main, capture_error_mode=mode)
        ctx.__enter__()
        x.add_(1)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            x.mul_(2)
        torch.cuda.current_stream().wait_stream(side)

        def second_thread():
            torch.cuda.set_device(0)
            for name, e in ev.items():
                res[("second thread", name)] = attempt(e.query)
            fresh = torch.cuda.Event()
            res[("second thread", "an event never recorded")] = attempt(fresh.query)
        th = threading.Thread(target=second_thread)
        th.start()
        th.join()
        ok = True
        try:
            ctx.__exit__(None, None, None)
        except Exception as e:          # the capturing thread's own queries below would invalidate it; done after the capture instead
            ok = False
            res[("capture", "end")] = "FAILED " + str(e).split("\n")[0][:80]
    print(f"capture mode {mode}: capture {'completed' if ok else 'failed'}")
    for (who, name), r in res.items():
        print(f"   {who:14s} query of an event recorded earlier on {name:38s}: {r}")
    torch.cuda.synchronize()
    time.sleep(0.2)
