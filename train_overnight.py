#!/usr/bin/env python3
"""`python train_overnight.py --config <overnight yaml> [--resume CKPT] [--skip-demos] [--skip-bc]` -- the reference's
learned_controllers/train_overnight.py entry point over the HIP path (see hcrl_amd/train_overnight.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hcrl_amd  # noqa: E402,F401
from hcrl_amd.train_overnight import main  # noqa: E402

if __name__ == "__main__":
    main()
