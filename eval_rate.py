#!/usr/bin/env python3
"""`python eval_rate.py --model <checkpoint.pt> [--compare-pid] [--n-episodes N]` -- the reference's
learned_controllers/eval_rate.py entry point over the HIP path (see hcrl_amd/eval_rate.py).

    python eval_rate.py --pid-only --n-episodes 4096 --difficulty hard
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hcrl_amd  # noqa: E402,F401
from hcrl_amd.eval_rate import main  # noqa: E402

if __name__ == "__main__":
    main()
