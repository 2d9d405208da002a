"""Bytes per TCP_TCC_READ_REQ on this GPU: a streaming read of a known size under
rocprofv3 --pmc TCP_TCC_READ_REQ_sum (run: rocprofv3 --pmc TCP_TCC_READ_REQ_sum -d out -- python3 l2_request_size.py)."""
import torch
x = torch.ones(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
torch.cuda.synchronize()
for _ in range(3):
    y = x.sum()
torch.cuda.synchronize()
print("read 3 x %d bytes" % (x.numel() * 4), float(y))
