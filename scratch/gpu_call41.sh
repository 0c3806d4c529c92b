cd $GRAFT_REPO_ROOT
timeout -k 10 200 python scratch/dbg_graph_grads.py 2>&1 | grep replay | cut -c1-600
