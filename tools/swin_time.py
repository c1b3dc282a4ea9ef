"""SwinTRN (BASELINE configs[3]) training step alone: ms/step + the per-family profile (run on the GPU box; under
`rocprofv3 --kernel-trace --stats` for the per-kernel table)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

r = bench.swin_report(torch.device("cuda:0"))
print(json.dumps(r))
