"""Helpers for the tests that run a rank in a spawned process (one-rank RCCL communicators on the GPU box)."""


def await_worker(p, q, timeout):
    """the worker's answer, or None as soon as the worker is gone without one (a rank that dies -- std::terminate on a helper thread
    -- never answers: waiting out the whole timeout cost the suite ten minutes when that happened)"""
    import queue
    import time
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            return q.get(timeout=2)
        except queue.Empty:
            if not p.is_alive():
                try:
                    return q.get(timeout=2)
                except queue.Empty:
                    return None
    return None
