"""GPU (tools): slam_logmel_fwd at the C3 batch (31 clips x 30 s, 128 mels) and at 1 clip; median of 20 launches."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_llm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for B, n_mels in ((31, 128), (31, 80), (1, 128)):
    audio = torch.randn(B, 480000, device=dev) * 0.1
    ops.logmel(audio, n_mels)
    ts = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.logmel(audio, n_mels)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    us = statistics.median(ts)
    frames = B * 3000
    flops = frames * 2.0 * 208 * 416      # the folded products as issued (13 bin tiles x 104 k-steps x 16 x 16 x 4 x 2 per 16 frames)
    print(json.dumps(dict(B=B, n_mels=n_mels, us=round(us, 1), fp32_mfma_TFs=round(flops / us / 1e6, 1),
                          out_TBs=round(frames * n_mels * 4 * 3 / us / 1e6, 3))), flush=True)
