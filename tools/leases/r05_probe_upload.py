"""probe: cost of one collated batch's upload -- pageable .to() vs pin_memory() + non_blocking -- and of a producer thread (GPU box)"""
import importlib, os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
dataset = importlib.import_module("a-nerf_amd.dataset")
d = cases.dataset_dict("surreal_full")
up = lambda a: np.kron(a, np.ones((1, 6, 5, 1), a.dtype))
for k in ("imgs", "masks", "sampling_masks", "bkgds"):
    d[k] = up(d[k])
p = os.path.join(tempfile.mkdtemp(), "x_train_h5py.npz")
dataset.write_npz_twin(p, d)
q = [0, 3, 5, 7, 9, 11, 13, 20]
for dev in ("cpu", "cuda"):
    ds = dataset.H5PoseData(p, device=dev, kind="surreal")
    for _ in range(20):
        ds.sample_batch(q, 128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        b = ds.sample_batch(q, 128)
    torch.cuda.synchronize()
    print(dev, "sample_batch ms", (time.perf_counter() - t0) / 200 * 1e3)
cols = {k: v.cpu().numpy() for k, v in b.items() if k != "rays"}
for name, fn in (("pageable .to", lambda a: torch.as_tensor(a).to("cuda")),
                 ("pin_memory + non_blocking", lambda a: torch.from_numpy(a).pin_memory().to("cuda", non_blocking=True))):
    for _ in range(20):
        [fn(a) for a in cols.values()]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        [fn(a) for a in cols.values()]
    torch.cuda.synchronize()
    print(name, "ms per batch of", len(cols), "tensors:", (time.perf_counter() - t0) / 200 * 1e3)
# one flat pinned staging buffer, one copy
tot = sum(a.nbytes for a in cols.values())
stage = torch.empty(tot, dtype=torch.uint8).pin_memory()
devbuf = torch.empty(tot, dtype=torch.uint8, device="cuda")
def flat():
    o = 0
    sv = stage.numpy()
    for a in cols.values():
        sv[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
        o += a.nbytes
    devbuf.copy_(stage, non_blocking=True)
for _ in range(20):
    flat()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    flat(); torch.cuda.synchronize()
print("one pinned staging buffer + one copy (+sync) ms:", (time.perf_counter() - t0) / 200 * 1e3, "bytes", tot)
