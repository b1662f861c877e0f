"""GPU: the drop-in surface end to end, as run_nerf.py drives the reference (run_nerf.py:504-640) -- a dataset in the reference's
on-disk layout (written from a teacher's renders: there is no dataset in the image), the dataset reader's sampling and collate,
create_raycaster from its data_attrs, Trainer.train_batch replayed from the captured hipGraph, a checkpoint in the reference's
format, reload, render_path of a held-out camera (tools/train_synthetic.py).  The student must actually learn (loss down, PSNR up)
and the reloaded checkpoint must render the same image."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_dataset_to_trainer_to_checkpoint_to_render_path(tmp_path):
    spec = importlib.util.spec_from_file_location("train_synthetic", os.path.join(ROOT, "tools", "train_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.main(["--iters", "150", "--hw", "64", "--n-kps", "3", "--n-cams", "3", "--n-rand", "512", "--n-sample-images", "4",
                  "--graph", "on", "--out", str(tmp_path)])
    assert r["graph"] and r["graphs"]["eager"] == 2 and r["graphs"]["replays"] == 148 and r["graphs"]["captures"] >= 1
    assert r["last"][1] < 0.6 * r["first"][1], r                      # the loss fell
    assert r["psnr_gain_db"] > 2.0, r                                  # ... and the PSNR against the teacher's pixels rose
    assert r["held_out_psnr_db"] > 10.0, r                             # a camera the student never saw
    assert r["testset_frames"] == 9 and r["testset_psnr_db"] > 15.0, r      # the dataset's render subset through render_path (run_nerf's test render)
    assert r["reload_max_abs_diff"] == 0.0                             # checkpoint round trip: the same image, bit for bit
    assert os.path.exists(r["checkpoint"]) and os.path.exists(r["dataset"])


@pytest.mark.gpu
def test_pose_refinement_recovers_perturbed_poses(tmp_path):
    """A-NeRF's own use: the dataset's poses are estimates (every joint rotation off by N(0, 0.05 rad)), the images show the true
    poses.  The subject is an analytic ball-and-stick body, so that shape and colours follow the pose.  create_popt builds the pose
    layer / its Adam / the anchors from the dataset's attributes (run_nerf.py:523), FusedAdam.from_torch puts both optimisers in one
    bucket, the Trainer refines poses and networks together from captured graphs -- the photometric loss must pull the poses
    back: mean per-joint error (scene units / ext_scale, the reference's MPJPE convention) more than halved in 400 iterations
    after the subject was learnt on the true poses (measured: 85 -> 12.6 mm; 800 iterations: profiles/r05_pose_refine.txt)."""
    spec = importlib.util.spec_from_file_location("train_synthetic", os.path.join(ROOT, "tools", "train_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.main(["--subject", "spheres", "--pose-noise", "0.05", "--pretrain", "1000", "--iters", "400", "--pose-step", "1",
                  "--graph", "on", "--out", str(tmp_path)])
    p = r["pose_refinement"]
    assert r["pretrain"]["last"][2] > r["pretrain"]["first"][2] + 6.0, r          # the subject was learnt (PSNR up by > 6 dB)
    assert 60.0 < p["mpjpe_mm_start"] < 120.0 and p["mpjpe_mm_end"] < 0.5 * p["mpjpe_mm_start"], p
    assert p["pose_steps"] == 400 and p["reloaded_pose_adam_steps"] == 400 and p["reloaded_layer_identical"], p
    # (each new variant -- another number of distinct poses in the batch -- runs eagerly once before its capture)
    assert 390 <= r["graphs"]["replays"] <= 398 and r["graphs"]["replays"] + r["graphs"]["eager"] == 400 and r["reload_max_abs_diff"] == 0.0, r
