#!/usr/bin/env python3
"""`dswx_hls_batch.py --gpus N rc1.yaml rc2.yaml ...`: many tiles over the GPUs of a node."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteus_amd.batch import main   # noqa: E402

if __name__ == '__main__':
    sys.exit(main())
