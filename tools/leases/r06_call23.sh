#!/bin/bash
# round 6, call 23: soak -- pose refinement end to end (fused input-gradient kernel, kp_loss_add, per-variant graph warm-up), graph vs eager
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
for g in on off; do
  python tools/train_synthetic.py --subject spheres --pose-noise 0.05 --pretrain 600 --iters 2000 --pose-step 4 --graph $g --out /tmp/soak_$g 2>&1 | tail -4 | cut -c1-900 > $O/r06_soak_$g.txt
  tail -1 $O/r06_soak_$g.txt | cut -c1-700
done
python - <<'P'
import json
a=[json.loads(open(f"gpurun_out/r06_soak_{g}.txt").read().strip().splitlines()[-1]) for g in ("on","off")]
keys=[k for k in a[0] if k in a[1] and k not in ("graph","graphs","it_per_s","seconds","wall_s","checkpoint","dataset")]
diff={k:(a[0][k],a[1][k]) for k in keys if a[0][k]!=a[1][k]}
print("graph vs eager: differing keys:", list(diff)[:10] if diff else "none (identical)"); print("graphs:", a[0].get("graphs"), "it/s", a[0].get("it_per_s"), a[1].get("it_per_s"))
P
