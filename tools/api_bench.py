import importlib, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
synth = importlib.import_module("a-nerf_amd.synth"); render_mod = importlib.import_module("a-nerf_amd.render")
import test_hip_backward as T
from cases import build
c = build("eval_s32"); caster = T.make_caster(c).eval()
sc = synth.make_scene(0, 512, 512, 600.0); dev = T.dev
n = len(sc["rays_o"]); ro, rd = dev(sc["rays_o"]), dev(sc["rays_d"])
kp = dev(sc["pose"]["kp"])[None].expand(n, 24, 3); skts = dev(sc["pose"]["skts"])[None].expand(n, 24, 4, 4)
bones = dev(sc["pose"]["bones"])[None].expand(n, 24, 3); cyl = dev(sc["cyl"])[None].expand(n, 5)
pk = {"density_scale": 1.0, "density_fn": torch.nn.functional.relu}
for prec in ["fp32", "bf16x3"]:
    caster.render_precision = prec
    for chunk in [4096, 32768, 1 << 20]:
        f = lambda: render_mod.render(512, 512, 600.0, chunk=chunk, rays=(ro, rd), use_viewdirs=True, ray_caster=caster, cams=None, subject_idxs=None,
                                      N_samples=64, N_importance=0, perturb=0.0, raw_noise_std=0.0, preproc_kwargs=pk, kp_batch=kp, skts=skts, cyls=cyl, bones=bones)
        with torch.no_grad():
            f(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): f()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"{prec:7s} render() chunk={chunk:8d}: {dt*1e3:7.1f} ms/frame  {n/dt/1e6:.3f} M rays/s")
