#!/usr/bin/env python3
"""`python recover_training.py [--checkpoint-step S] [--output PATH] [--checkpoint-dir D] [--eval-dir D]` -- the
reference's learned_controllers/recover_training.py entry point (see hcrl_amd/recover_training.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hcrl_amd  # noqa: E402,F401
from hcrl_amd.recover_training import main  # noqa: E402

if __name__ == "__main__":
    main()
