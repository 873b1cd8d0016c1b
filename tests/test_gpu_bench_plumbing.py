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


def test_bench_two_ranks_on_one_device(built, tmp_path):
    """bench.py's N > 1 path end to end on the 1-GPU box: two gloo ranks share the device (RCCL refuses two ranks on one device), each
    renders its own tiles, rank 0 gathers the peer's tiles and reads the frame back.  One JSON line, both ranks' samples counted, image
    finite; the gathered frame's mean equals the one-rank run's (same seed => same frame)."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BENCH_DIST_BACKEND="gloo")
    common = ["--steps", "6", "--warmup", "2", "--width", "640", "--height", "360", "--triangles", "20000", "--no-pmc", "--no-cpu-baseline"]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d2 = json.loads(lines[0])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=dict(os.environ), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d2["n_gpus"] == 2 and d2["image"]["finite"] and d2["value"] > 0
    assert d2["counters"]["numRays"] == d1["counters"]["numRays"] and d2["counters"]["numShadowRays"] == d1["counters"]["numShadowRays"]
    assert d2["image"]["mean_per_pass"] == d1["image"]["mean_per_pass"]
    # a line says which switches changed its work: the two-rank run names its backend and BENCH_DIST_BACKEND, the one-rank run nothing of the kind
    assert d2["config"]["env"].get("BENCH_DIST_BACKEND") == "gloo" and d2["config"]["dist_backend"] == "gloo" and d2["config"]["emulated_shard"] is None
    assert "BENCH_DIST_BACKEND" not in d1["config"]["env"] and d1["config"]["emulated_shard"] is None and "EMULATED" not in d1["metric"]


def test_bench_line_names_every_switch_that_changed_its_work(built):
    """Round-4 review item 7: an emulated-shard line (BENCH_EMULATE_SHARD=N renders 1/N of the frame) must not be readable as a whole-frame line, and any
    RTGPU_* / BENCH_* variable in the environment goes into config.env -- also on the reduced BENCH_TIMED_ONLY line."""
    import json
    import subprocess
    common = ["--steps", "4", "--warmup", "2", "--width", "640", "--height", "360", "--triangles", "20000", "--no-pmc", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if not k.startswith(("BENCH_", "RTGPU_"))}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, env=dict(env, BENCH_EMULATE_SHARD="4", RTGPU_PACKET="0"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["emulated_shard"] == [0, 4] and "EMULATED SHARD: 1/4 of the frame" in d["metric"] and d["config"]["parallelism"].startswith("emulated shard 0 of 4")
    assert d["config"]["env"] == {"BENCH_EMULATE_SHARD": "4", "RTGPU_PACKET": "0"} and d["n_gpus"] == 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, env=dict(env, BENCH_TIMED_ONLY="1", BENCH_EMULATE_SHARD="2"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["timed_only"] and d["config"]["emulated_shard"] == [0, 2] and d["config"]["env"] == {"BENCH_EMULATE_SHARD": "2", "BENCH_TIMED_ONLY": "1"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["env"] == {} and d["config"]["emulated_shard"] is None and "EMULATED" not in d["metric"]
    # the frame's PCIe read-back is timed behind the region and reported beside `value`, never inside it (measurement contract)
    hr = d["host_readback"]
    assert hr["in_value"] is False and hr["ms"] > 0 and hr["bytes"] == 640 * 360 * 12 and 0 < hr["value_incl_host_readback"] < d["value"]
    assert "HBM" in d["config"]["timed_region"] and "host_readback" in d["config"]["timed_region"]
    # the roofline of a --no-pmc line: launch time and the calibration are there, but a 4-wide walk without its own counts (they come from a child run the
    # flag skips) claims no fraction -- SURVEY 8(d)'s model of the reference's BINARY walk exceeds the HBM peak and is kept as a work measure only
    r = d["roofline"]
    assert r["kernel"] == "k_trace_wide" and r["avg_launch_ms"] > 0 and "FETCH_SIZE_true_bytes_per_reported_byte" in r["calibration"]
    assert r["frac"] is None and r["achieved"] is None and r["algorithmic_bytes_per_launch"] is None and str(r["algorithmic_model"]).startswith("n/a")
    assert r["traffic"] is None and r["reference_walk_bytes_per_launch"] > 0
