cd $GRAFT_REPO_ROOT
TAG=default timeout -k 10 200 python scratch/dbg_train_nan.py 2>&1 | grep iter
TAG=eager UG=0 timeout -k 10 200 python scratch/dbg_train_nan.py 2>&1 | grep iter
TAG=nopre FDYN_NO_PRE=1 timeout -k 10 200 python scratch/dbg_train_nan.py 2>&1 | grep iter
TAG=nomfma FDYN_MFMA_TRAIN=0 FDYN_NO_PRE=1 timeout -k 10 200 python scratch/dbg_train_nan.py 2>&1 | grep iter
TAG=notrunk FDYN_NO_TRUNK=1 timeout -k 10 200 python scratch/dbg_train_nan.py 2>&1 | grep iter
