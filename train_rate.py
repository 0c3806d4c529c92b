#!/usr/bin/env python3
"""`python train_rate.py --config <yaml> [--no-lstm] [--bf16] [--bc-pretrain N]` -- the reference's
learned_controllers/train_rate.py entry point over the HIP path (see hcrl_amd/train_rate.py).

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_rate.py --bf16
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hcrl_amd  # noqa: E402,F401
from hcrl_amd.train_rate import main  # noqa: E402

if __name__ == "__main__":
    main()
