#!/usr/bin/env python3
"""Developer: run the attention kernel a few times (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
crops = 512; T = crops * 257
qkv = torch.randn(T, 3072, device=dev).to(torch.bfloat16)
o = torch.empty(T, 1024, device=dev, dtype=torch.bfloat16)
for _ in range(10):
    lib.clipenc_op_attention(qkv.data_ptr(), o.data_ptr(), crops, 257, 1024, 16, st)
torch.cuda.synchronize()
