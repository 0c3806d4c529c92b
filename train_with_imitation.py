#!/usr/bin/env python3
"""`python train_with_imitation.py [--residual] [--rl-steps S] [--n-envs N]` -- the reference's
learned_controllers/train_with_imitation.py entry point over the HIP path (see hcrl_amd/train_with_imitation.py).

    python train_with_imitation.py --residual --rl-steps 50000000 --n-envs 16384
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hcrl_amd  # noqa: E402,F401
from hcrl_amd.train_with_imitation import main  # noqa: E402

if __name__ == "__main__":
    main()
