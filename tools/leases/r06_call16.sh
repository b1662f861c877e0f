cd $GRAFT_REPO_ROOT; python tools/diag/gate_bones_dskts.py 2>&1 | tail -12
