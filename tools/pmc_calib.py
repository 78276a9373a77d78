"""Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950: copy a known number of bytes
(far larger than the 256 MiB Infinity Cache) and compare the counters with the byte count."""
import torch
n = 1 << 28  # 1 GiB of float32
x = torch.ones(n, device="cuda")
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
print("copied", 4 * n, "bytes per dispatch")
