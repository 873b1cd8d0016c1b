"""N > 1 path on CPU: two gloo ranks each render the tiles they own (with the CPU oracle standing in for the
device pass -- same shard arithmetic, same per-pass constants from the same seed), rank 0 gathers the peers' owned
tiles with bench.py's own TileGather (the class the RCCL run uses, on CPU tensors here), and the result must equal
the unsharded image bit for bit -- including -0.0 and the untouched pixels of rank 0."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = sys.argv[1]; out = sys.argv[2]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as ra
from raytracer_amd import scenes
import oracle_lib
import bench
# bench.py's own set-up of the process group: RCCL as asked, gloo (host-staged exchange) if the communicator does not come up -- which is the case here (no GPU)
backend, why = bench.init_process_group(dist, os.environ.get("WORKER_BACKEND", "gloo"), timeout_s=int(os.environ.get("WORKER_TIMEOUT_S", "60")))
if os.environ.get("BENCH_TEST_FAIL_NCCL_ON_RANK") is not None:
    # a ONE-SIDED failure: the other rank's group would have come up (gloo stands in for RCCL; it waits for its peer inside init until the group's time-out, where
    # RCCL's lazy communicator would return at once and learn of the failure from the peer's flag); both ranks must end on the fallback and say why
    assert backend == "gloo" and why and "nccl process group failed" in why, (backend, why)
elif os.environ.get("BENCH_TEST_NCCL_STAND_IN"):
    assert backend == "gloo" and why is None        # (the stand-in group came up on every rank: no fallback, the agreement said yes)
elif os.environ.get("WORKER_BACKEND") == "nccl":
    assert backend == "gloo" and why and "nccl process group failed" in why, (backend, why)
else:
    assert backend == "gloo" and why is None
rank, world = dist.get_rank(), dist.get_world_size()
w, h = 200, 136
scene, camera = scenes.cornell_box(w / h)
bn = ra.load_blue_noise(); scene.desc.contents.blueNoise = bn.ctypes.data
vp = ra.Viewport(w, h, seed=5, max_ray_depth=4)         # same seed on every rank => same per-pass constants
img = np.zeros((h, w, 3), dtype=np.float32)
cnt = np.zeros(16, dtype=np.uint64)
for _ in range(2):
    p = vp.next_pass_params(camera)
    oracle_lib.render_pass(scene.desc, p, w, h, img, None, cnt, shard=(rank, world), threads=2)
t = torch.from_numpy(img)
own = bench.owned_pixel_indices(w, h, rank, world)
others = np.setdiff1d(np.arange(w * h), own)
assert not img.reshape(-1, 3)[others].any()            # a shard never touches a pixel it does not own
gather = bench.TileGather(torch, dist, w, h, rank, world, t.view(-1), device="cpu")
mode = gather.choose_mode()                            # the bench's warm-up: tries the collective, agrees on gather / grouped send-recv across the ranks
assert mode == os.environ.get("EXPECT_MODE", "gather"), (mode, gather.mode_reason)
gather.run()
gather.run()                                           # idempotent: the bench warms the collective with one extra call
rays = torch.tensor([int(cnt[0])], dtype=torch.int64)
dist.all_reduce(rays)
if rank == 0:
    np.save(out, np.concatenate([t.numpy().reshape(-1), np.array([float(rays.item())], dtype=np.float32)]))
dist.barrier()
dist.destroy_process_group()
'''


import pytest


@pytest.mark.parametrize("exchange", ["gather", "send_recv", "nccl_fallback", "nccl_one_sided_failure", "nccl_agreed"])
def test_two_rank_tile_sharding_matches_single_rank(built, tmp_path, exchange):
    """exchange = send_recv: the fallback bench.py takes when the gather collective does not come up on a fabric (BENCH_GATHER=send_recv forces it):
    grouped isend / irecv of the same packed tiles -- the same frame.  nccl_fallback: the ranks ask for RCCL where there is none (this container has no
    GPU): bench.init_process_group falls back to gloo on every rank, says why, and the frame is the same.  nccl_one_sided_failure / nccl_agreed: the explicit
    agreement of round 6 (a flag per rank in a side store before the first collective), with one rank failing and with none."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import raytracer_amd as ra
    from raytracer_amd import scenes
    import oracle_lib
    w, h = 200, 136
    scene, camera = scenes.cornell_box(w / h)
    bn = ra.load_blue_noise()
    scene.desc.contents.blueNoise = bn.ctypes.data
    vp = ra.Viewport(w, h, seed=5, max_ray_depth=4)
    whole = np.zeros((h, w, 3), dtype=np.float32)
    cnt = np.zeros(16, dtype=np.uint64)
    for _ in range(2):
        oracle_lib.render_pass(scene.desc, vp.next_pass_params(camera), w, h, whole, None, cnt, threads=4)

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / "reduced.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", EXPECT_MODE="send_recv" if exchange == "send_recv" else "gather")
    if exchange == "send_recv":
        env["BENCH_GATHER"] = "send_recv"
    if exchange == "nccl_fallback":
        env["WORKER_BACKEND"] = "nccl"
    if exchange == "nccl_one_sided_failure":
        # round 6 (advisor): rank 1's RCCL group fails, rank 0's would come up (gloo stands in for RCCL where there is no GPU) and waits for its peer inside
        # init -- until the group's time-out (12 s here); the flags in the side store then send BOTH ranks to the gloo fallback, which rendezvouses through the
        # side store under its own prefix
        env.update(WORKER_BACKEND="nccl", BENCH_TEST_NCCL_STAND_IN="gloo", BENCH_TEST_FAIL_NCCL_ON_RANK="1", WORKER_TIMEOUT_S="12")
    if exchange == "nccl_agreed":
        env.update(WORKER_BACKEND="nccl", BENCH_TEST_NCCL_STAND_IN="gloo")   # every rank's group comes up: the agreement says yes, no fallback
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                           "--master-port", {"gather": "29541", "send_recv": "29543", "nccl_fallback": "29545", "nccl_one_sided_failure": "29547", "nccl_agreed": "29549"}[exchange], str(script), ROOT, out], env=env, timeout=600)
    data = np.load(out)
    reduced = data[:-1].reshape(h, w, 3)
    assert np.array_equal(reduced.view(np.uint32), whole.view(np.uint32))
    assert int(data[-1]) == int(cnt[0])
