"""The torch plumbing bench.py uses for N > 1 (one rank per GPU): a zero-copy torch view of a device buffer the library owns
(`bench.device_tensor`, __cuda_array_interface__) and an RCCL reduce on it.  One rank is enough to prove the wrapping works on
this ROCm build; the sharded image arithmetic itself is covered on CPU with gloo (test_multi_rank_cpu.py)."""
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_tensor_view_and_rccl_reduce(built):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x = torch.arange(1024, dtype=torch.float32, device="cuda")
        t = bench.device_tensor(x.data_ptr(), x.numel(), torch)
        assert t.is_cuda and t.data_ptr() == x.data_ptr()
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert float(t.sum()) == 1023 * 1024 / 2
        t += 1
        assert float(x[0]) == 1.0      # a view of the same memory, not a copy
    finally:
        dist.destroy_process_group()
