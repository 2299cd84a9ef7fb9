#!/usr/bin/env python3
"""bench.py's training leg for ONE trainer, alone (for rocprofv3 --kernel-trace --stats):
    python3 tools/train_line.py vits|matcha|matcha_mas|fs2 [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "vits"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
line = bench.train_step_line(torch.device("cuda:0"), steps, kind)
line.pop("ms_per_step_all", None)
print(json.dumps(line))
