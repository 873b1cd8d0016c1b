#!/usr/bin/env python3
"""bench.py -- Msamples/s (paths x bounces = RayTracingCounters::numRays) of the MI355X path-tracing core.

Workload (BASELINE.json configs[2]): Sponza-class triangle mesh (~262k triangles, procedural stand-in: the
reference's sponza.obj is not shipped), PathTracerMIS, 8 bounces, 1920x1080, background + delta directional
light, LightSamplingStrategy::Single.  One "step" = one pass = one sample per pixel of the whole frame.

  python bench.py --gpus 1 --steps 256 --warmup 16        (the defaults: BASELINE config 3 renders 256 samples per pixel)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --workload bdpt-glass                   (BASELINE configs[4]: rough-glass slab, renderer "VCM" without merging)

Timed region: the K passes + (N > 1) the gather + the final Viewport::GetSumBuffer (read_sum): the clock stops when the accumulated
float3 frame lies in the viewport's page-locked host bitmap.  N > 1: the frame's 64x64 tiles are interleaved across ranks (tile % N == rank,
identical scene on every GPU, no data-path collective); the passes are followed by ONE gather of the owned tiles to rank 0 over RCCL (packed tile
pixels, 24.9 MB / N per peer; the collective is warmed up before the timed region), inside the timed region.  Total work is fixed => "scaling":
"strong".  The read-back (24.9 MB over PCIe, ~0.45 ms; the bitmaps are page-locked by a warm-up read-back in front of the region) is the region's
last step on rank 0; the rate without it -- the frame complete in rank 0's HBM, what BENCH_r05.json's `value` was -- is printed beside `value` as
`frame_in_hbm`, the copy's own time as `host_readback`, and `config.value_definition` names the definition ("survey-8d/r6").

The roofline block is measured in the run itself: HIP-event launch times from a serial (one batch lane) replay of the same
passes; `roofline.frac` = HBM-side traffic of the dominant kernel from two `rocprofv3 --pmc` child runs of the same passes
(FETCH_SIZE, WRITE_SIZE, calibrated per kernel class; separate
passes, kernel-trace only).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# What `value` times.  "survey-8d/r6": SURVEY 8(d)'s region -- K x Viewport::Render (+ the gather, N > 1) + the final Viewport::GetSumBuffer -- with the
# viewport's bitmaps page-locked by a warm-up read-back in front (BENCH_r01..r04: the same region, but with that one-time page-locking inside it;
# BENCH_r05: the region ended with the frame complete in HBM -- this line's frame_in_hbm.value).
VALUE_DEFINITION = "survey-8d/r6"
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth, same guide
SEED = 20260928


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)   # the 256 spp of BASELINE config 3: the whole frame of the metric's configuration
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--triangles", type=int, default=262144)
    ap.add_argument("--workload", default="sponza", choices=["sponza", "sponza-textured", "sponza-all", "cornell", "sphere", "zoo", "bdpt-glass"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child runs (roofline.traffic = null)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the bounded CPU-baseline sample")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # internal: the profiled child of a bench run
    ap.add_argument("--walk-diag-child", action="store_true", help=argparse.SUPPRESS)   # internal: the instrumented-walk child (RTGPU_WIDE_DIAG=3)
    return ap.parse_args()


def build_scene(args, aspect):
    from raytracer_amd import scenes
    if args.workload in ("sponza", "sponza-textured", "sponza-all"):
        return scenes.sponza_class(aspect, args.triangles, textured=args.workload == "sponza-textured")
    if args.workload == "zoo":   # every light type x every BSDF on analytic shapes: the scene that mixes all hit kinds (tests/scene_zoo.py)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        import scene_zoo
        return scene_zoo.all_lights_scene(aspect)
    if args.workload == "cornell":
        return scenes.cornell_box(aspect)
    if args.workload == "bdpt-glass":
        return scenes.rough_glass_slab(aspect)
    return scenes.sphere_area_light(aspect)


def make_viewport(ra, args, scene, device, shard=None):
    from raytracer_amd import scenes
    if args.workload == "sponza-all":   # SURVEY 8(d)'s path-exact run of config 3: LightSamplingStrategy::All with dimensions = 128 (PathTracerMIS.cpp:141-147)
        vp = ra.Viewport(args.width, args.height, seed=SEED, max_ray_depth=args.depth, dimensions=128, light_sampling_all=True)
    else:
        vp = ra.Viewport(args.width, args.height, seed=SEED, max_ray_depth=args.depth)
    if args.workload == "bdpt-glass":
        vp.set_renderer(scene, name="VCM", device=device, intersection_counters=False)
        vp.set_vcm(**scenes.ROUGH_GLASS_SLAB_VCM)
    else:
        vp.set_renderer(scene, device=device, intersection_counters=False)   # the library's default (the reference's): counters off
    if shard is not None and shard[1] > 1:
        vp.set_shard(*shard)
    return vp


def open_side_store(dist, connect_timeout_s=120):
    """The side TCPStore of init_process_group's backend agreement: MASTER_PORT + 1 (BENCH_SIDE_PORT overrides), rank 0 serves.  main() opens it FIRST -- before
    rank 0 builds the library, which can take a minute while the other ranks are already waiting -- so that every rank finds it within seconds.  None if it
    cannot be set up (the port is taken: rank 0 cannot bind, the others cannot connect): every rank then sees the same thing and the agreement is skipped."""
    import datetime
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    try:
        port = int(os.environ.get("BENCH_SIDE_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
        side = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=connect_timeout_s), wait_for_workers=False)
        side.set("bench_side_hello_%d" % rank, "1")     # (a foreign service on that port fails here, not in the middle of the agreement)
        return side
    except Exception as e:   # noqa: BLE001
        sys.stderr.write("[bench] rank %d: no side store for the backend agreement (%r): the fallback is decided by exceptions only\n" % (rank, e))
        return None


def init_process_group(dist, backend, device=None, timeout_s=300, side=None):
    """The N > 1 run's process group: RCCL (`nccl`) as asked, and -- if the communicator does not come up on this node -- gloo with host-staged tile exchange
    instead of no measurement at all: the exchange is 24.9 MB once per timed region, the passes do not communicate.  Returns (backend in use, reason for a
    fallback or None).

    Round 6 (advisor): the ranks AGREE on the fallback explicitly instead of finding out through a barrier's time-out.  A side TCPStore on MASTER_PORT + 1
    (BENCH_SIDE_PORT overrides; rank 0 serves) carries one flag per rank -- "my RCCL group came up" or the exception -- written BEFORE the first collective; a
    rank whose peer reported a failure (or did not report within the time-out) never enters the barrier it would hang in.  The gloo group of the fallback
    rendezvouses through that side store under its own prefix, so nothing a half-initialised RCCL group left in the default store can be in its way.  If the
    side store itself cannot be set up (the port is taken), every rank sees that alike (rank 0 cannot serve, the others cannot connect) and the round-5
    behaviour remains: try, and fall back on whatever raises.  The first barrier still decides (a communicator can fail inside it): its time-out is the
    group's, `timeout_s`."""
    import datetime
    if backend != "nccl":
        dist.init_process_group(backend)
        return backend, None
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if side is None and world > 1:
        side = open_side_store(dist)
    ok, reason = True, None
    stand_in = os.environ.get("BENCH_TEST_NCCL_STAND_IN")   # test hook (tests/test_multi_rank_cpu.py): a backend that comes up where there is no GPU plays RCCL's part
    try:
        if os.environ.get("BENCH_TEST_FAIL_NCCL_ON_RANK") == str(rank):   # test hook: a one-sided failure
            raise RuntimeError("simulated RCCL failure on rank %d (BENCH_TEST_FAIL_NCCL_ON_RANK)" % rank)
        if stand_in:
            dist.init_process_group(stand_in, timeout=datetime.timedelta(seconds=timeout_s))
        else:
            dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=timeout_s))
    except Exception as e:   # noqa: BLE001 -- whatever the communication library throws: the fallback decides, the reason goes into the JSON line
        ok, reason = False, "nccl process group failed on rank %d: %r" % (rank, e)
    if side is not None:
        try:
            side.set("bench_nccl_%d" % rank, "ok" if ok else reason[:400])
            for r in range(world):
                try:
                    side.wait(["bench_nccl_%d" % r], datetime.timedelta(seconds=timeout_s + 60))
                    flag = side.get("bench_nccl_%d" % r).decode(errors="replace")
                except Exception:   # noqa: BLE001
                    flag = "rank %d did not report its RCCL group within %d s" % (r, timeout_s + 60)
                if flag != "ok" and ok:
                    ok, reason = False, "nccl process group failed on a peer: " + flag
        except Exception as e:   # noqa: BLE001 -- the side store died: decide alone, as before
            sys.stderr.write("[bench] rank %d: backend agreement failed (%r)\n" % (rank, e))
    if ok:
        try:
            dist.barrier()
            return (stand_in or "nccl"), None
        except Exception as e:   # noqa: BLE001
            ok, reason = False, "nccl process group failed in its first barrier on rank %d: %r" % (rank, e)
    sys.stderr.write("[bench] %s -- falling back to gloo (host-staged tile exchange)\n" % reason)
    try:
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:   # noqa: BLE001
        pass
    # torch names the default group after a per-process counter that every ATTEMPT advances: a rank whose RCCL attempt got as far as taking a name and a rank whose
    # attempt failed earlier would look for each other under different store keys and wait for ever.  Same counter on every rank before the fallback.
    try:
        dist.distributed_c10d._world.group_count = 1
    except Exception:   # noqa: BLE001 -- a torch without that field: the symmetric case (every rank failed alike) still works
        pass
    if side is not None:
        dist.init_process_group("gloo", store=dist.PrefixStore("bench_gloo_fallback", side), rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    else:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))
    dist.barrier()
    return "gloo", reason[:500]


def device_tensor(ptr, num_floats, torch):
    """torch view of a hipMalloc'ed float buffer owned by librtgpu (plumbing for the RCCL gather)."""
    class _Buf:
        pass
    b = _Buf()
    b.__cuda_array_interface__ = {"shape": (int(num_floats),), "typestr": "<f4", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(b, device="cuda")


def owned_pixel_indices(width, height, rank, world):
    """Row-major pixel indices of the 64x64 tiles rank `rank` of `world` owns (rtgpu_set_shard: tile % world == rank)."""
    ty, tx = np.meshgrid(np.arange(height) // 64, np.arange(width) // 64, indexing="ij")
    owned = ((ty * ((width + 63) // 64) + tx) % world) == rank
    return np.nonzero(owned.reshape(-1))[0].astype(np.int64)


class TileGather:
    """The final exchange of the multi-GPU path: every peer sends the pixels of its own tiles (packed float3) to rank 0, which
    writes them into its sum buffer -- bit-identical to the 1-GPU image because each pixel was accumulated by exactly one rank."""

    def __init__(self, torch, dist, width, height, rank, world, sum_tensor, device="cuda"):
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.image = sum_tensor.view(-1, 3)
        counts = [len(owned_pixel_indices(width, height, r, world)) for r in range(world)]
        self.pad = max(counts)
        self.own = torch.from_numpy(owned_pixel_indices(width, height, rank, world)).to(device)
        self.send = torch.zeros((self.pad, 3), dtype=torch.float32, device=device)
        self.recv = [torch.zeros((self.pad, 3), dtype=torch.float32, device=device) for _ in range(world)] if rank == 0 else None
        self.peers = [torch.from_numpy(owned_pixel_indices(width, height, r, world)).to(device) for r in range(world)] if rank == 0 else None

    mode = "gather"            # "gather": one dist.gather to rank 0; "send_recv": grouped point-to-point transfers (the fallback)
    mode_reason = "default"

    def choose_mode(self):
        """Outside the timed region: tries the gather collective once on every rank; if ANY rank's attempt fails, all ranks fall back to grouped
        isend / irecv (dist.batch_isend_irecv), and the reason is kept for the JSON line.  BENCH_GATHER=send_recv forces the fallback."""
        forced = os.environ.get("BENCH_GATHER", "")
        ok, why = 1.0, ""
        if forced == "send_recv":
            ok, why = 0.0, "BENCH_GATHER=send_recv"
        else:
            try:
                self._gather()
                if self.send.is_cuda:
                    self.torch.cuda.synchronize()
            except Exception as e:   # a collective that does not come up on this fabric must not take the measurement down
                ok, why = 0.0, "dist.gather failed on rank %d: %r" % (self.rank, e)
        flag = self.torch.tensor([ok], dtype=self.torch.float32, device=self.send.device if self.dist.get_backend() == "nccl" else "cpu")
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        if float(flag.item()) < 1.0:
            TileGather.mode = self.mode = "send_recv"
            self.mode_reason = why or "another rank's dist.gather failed"
            sys.stderr.write("[bench] rank %d: tile exchange falls back to grouped send / recv (%s)\n" % (self.rank, self.mode_reason))
        return self.mode

    def _gather(self):
        if self.send.is_cuda and self.dist.get_backend() != "nccl":
            # BENCH_DIST_BACKEND=gloo (the N > 1 code path exercised on a 1-GPU box): gloo gathers host tensors
            recv = [t.cpu() for t in self.recv] if self.rank == 0 else None
            self.dist.gather(self.send.cpu(), recv, dst=0)
            if self.rank == 0:
                for r in range(1, self.world):
                    self.recv[r].copy_(recv[r])
        else:
            self.dist.gather(self.send, self.recv, dst=0)

    def _send_recv(self):
        host = self.send.is_cuda and self.dist.get_backend() != "nccl"
        if self.rank == 0:
            bufs = [(t.cpu() if host else t) for t in self.recv]
            ops = [self.dist.P2POp(self.dist.irecv, bufs[r], r) for r in range(1, self.world)]
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
            if host:
                for r in range(1, self.world):
                    self.recv[r].copy_(bufs[r])
        else:
            for w in self.dist.batch_isend_irecv([self.dist.P2POp(self.dist.isend, self.send.cpu() if host else self.send, 0)]):
                w.wait()

    def run(self):
        self.send[:len(self.own)] = self.image.index_select(0, self.own)
        if self.mode == "send_recv":
            self._send_recv()
        else:
            self._gather()
        if self.rank == 0:
            for r in range(1, self.world):
                self.image.index_copy_(0, self.peers[r], self.recv[r][:len(self.peers[r])])


def algorithmic_bytes(c):
    """per interior-node visit two 32-byte children, 36 bytes per triangle test, 176 B per mesh hit,
    192 B per analytic hit, 24 B film read-modify-write per path (+12 B on even passes => x1.5)."""
    trace_closest = 32 * c["numRayBoxTests"] + 36 * c["numRayTriangleTests"]
    trace_shadow = 32 * c["numShadowRayBoxTests"] + 36 * c["numShadowRayTriangleTests"]
    shade = 176 * c["numMeshHits"] + 192 * c["numAnalyticHits"]
    film = 36 * c["numPrimaryRays"]
    # ("tail": the fused tail of a batch, rt_tail.hip, does trace and shade work of the late bounces -- the per-class counters do not separate it:
    #  its share is inside "trace" and "shade", which therefore overstate those two classes when a tail runs)
    return {"trace": trace_closest + trace_shadow, "shade": shade, "accumulate": film, "generate": 0, "retrace": 0, "tail": 0}


# kernel-name prefix (as rocprofv3 reports it) -> kernel class of rtgpu_get_kernel_times
KERNEL_CLASS_PREFIXES = (("k_trace_monster", None), ("k_tail", "tail"), ("k_trace_wide", "trace"), ("k_trace_packet", "trace"), ("k_trace_quant", "trace"), ("k_trace", "trace"), ("k_shade", "shade"),
                         ("k_vcm_light_finish", "accumulate"), ("k_vcm_camera_finish", "accumulate"), ("k_vcm_emit", "generate"), ("k_vcm_", "shade"),
                         ("k_lt_shade", "shade"), ("k_generate", "generate"), ("k_accumulate", "accumulate"))


def kernel_class(name, reencoded_walk=False):
    """reencoded_walk: the run's traversal kernel is k_trace_wide / k_trace_quant; the binary-tree k_trace then only re-traces the few rays
    those hand over (the library's class "retrace")."""
    base = name.replace("void ", "").strip()
    for prefix, cls in KERNEL_CLASS_PREFIXES:
        if base.startswith(prefix):
            if prefix == "k_trace" and reencoded_walk:
                return "retrace"
            return cls
    return None


def pmc_child_sums(args, counter, timeout_s):
    """pmc_child_sums_once with ONE retry: a rocprofv3 child that dies or writes no database (seen once in a few dozen runs on the pool's boxes) must not cost
    the line its roofline; the error of the last attempt is what is reported."""
    sums, err = pmc_child_sums_once(args, counter, timeout_s)
    if sums is None:
        sys.stderr.write("[bench] rocprofv3 child failed (%s); trying once more\n" % err)
        sums, err = pmc_child_sums_once(args, counter, timeout_s)
    return sums, err


def pmc_child_sums_once(args, counter, timeout_s):
    """Runs THIS command's passes (same workload, size, --steps, --warmup; one batch lane, intersection counters off) in a child
    process under `rocprofv3 --kernel-trace --pmc <counter ...>` and returns {kernel class: (sum of the counter, launches, ns)} -- or, when
    `counter` is a list (counters that fit one pass), {kernel class: {counter: sum, ..., "launches": n, "ns": ns}}."""
    counters = [counter] if isinstance(counter, str) else list(counter)
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="rtbench_pmc_", dir="/tmp")
    label = " ".join(counters)
    try:
        cmd = [rocprof, "--kernel-trace", "--pmc"] + counters + ["-d", tmp, "-o", "r", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child",
               "--steps", str(args.steps), "--warmup", str(args.warmup), "--width", str(args.width), "--height", str(args.height),
               "--depth", str(args.depth), "--triangles", str(args.triangles), "--workload", args.workload]
        env = dict(os.environ, TMPDIR="/tmp", RTGPU_LANES="1")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (label, r.returncode, (r.stderr or "")[-300:])
        cur = sqlite3.connect(dbs[0]).cursor()
        rows = None
        for query in ("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name",
                      "select k.name, p.counter_name, count(*), sum(p.value) from pmc_events p join kernels k on p.event_id = k.id group by k.name, p.counter_name"):
            try:
                rows = [row for row in cur.execute(query).fetchall() if row[1] in counters]
                if rows:
                    break
            except sqlite3.Error:
                rows = None
        durations = {}
        try:
            durations = {n: (int(c), int(d)) for n, c, d in cur.execute("select name, count(*), sum(duration) from kernels group by name").fetchall()}
        except sqlite3.Error:
            pass
        if not rows:
            return None, "no %s rows in the rocprofv3 database" % label
        reencoded = any(n.replace("void ", "").strip().startswith(("k_trace_wide", "k_trace_quant", "k_trace_packet")) for n, _, _, _ in rows)
        multi = {}
        for name, cname, launches, total in rows:
            cls = kernel_class(name, reencoded)
            if cls:
                a = multi.setdefault(cls, {})
                a[cname] = a.get(cname, 0.0) + float(total)
                seen = a.setdefault("_names", {})
                seen[name] = (int(launches), int(durations.get(name, (0, 0))[1] or 0))
        for cls, a in multi.items():
            names = a.pop("_names")
            a["launches"] = sum(v[0] for v in names.values()); a["ns"] = sum(v[1] for v in names.values())
        if isinstance(counter, str):
            return {cls: [a.get(counter, 0.0), a["launches"], a["ns"]] for cls, a in multi.items()}, None
        return multi, None
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 --pmc %s timed out" % label
    except Exception as e:   # a profiler problem must not take the benchmark down
        return None, "rocprofv3 --pmc %s: %r" % (label, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# what tools/microbench/cadence.hip measured on this chip (profiles/r03_microbench_cadence_l1.txt): a wave64 VALU instruction occupies its
# SIMD for 2.1 clocks with >= 5 resident waves (SIMD-32), and the walk's fetch pattern -- every lane its own 64-byte node, 16 bytes per
# access -- peaks at 1.8 L1 accesses per clock and CU from an L1 / L2-resident table
VALU_CLOCKS_PER_WAVE_INSTRUCTION = 2.1
L1_DIVERGENT_ACCESSES_PER_CLOCK_PER_CU = 1.8
PIPE_COUNTERS = ["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE",
                 "TCP_TOTAL_CACHE_ACCESSES_sum"]


# the divergent-fetch microbenchmark at 5 waves per SIMD and 36 of 64 lanes active (the walk's interior loop runs at 0.52-0.57 lane utilisation): L1
# accesses per clock and CU against the size of the table the nodes are drawn from (profiles/r03_microbench_cadence_l1.txt).  8 waves per SIMD give
# the same or less at every size: past the L2 these are THROUGHPUT figures of the cache hierarchy for this access pattern, not latency figures.
FETCH_PEAK_BY_TABLE_BYTES = [(8 << 10, 1.58), (1 << 20, 1.70), (8 << 20, 1.22), (64 << 20, 0.66), (1 << 30, 0.60)]


def fetch_peak_at(footprint_bytes):
    """Log-linear interpolation of FETCH_PEAK_BY_TABLE_BYTES."""
    import math
    t = FETCH_PEAK_BY_TABLE_BYTES
    if not footprint_bytes or footprint_bytes <= t[0][0]:
        return t[0][1]
    for (b0, p0), (b1, p1) in zip(t, t[1:]):
        if footprint_bytes <= b1:
            return p0 + (p1 - p0) * (math.log(footprint_bytes / b0) / math.log(b1 / b0))
    return t[-1][1]


def measure_pipes(args, num_cus, footprint_bytes=None):
    """Which ceiling is a kernel near?  A third --pmc child run of the same passes (SQ / GRBM / TCP counters that fit one pass), per
    launch and kernel class:
      valu_issue  -- wave-level VALU instructions x 2.1 clocks (the microbenchmark's cadence) / SIMD clocks available (4 SIMDs per CU x the
                     kernel's clocks, GRBM_GUI_ACTIVE summed over the 8 XCDs / 8); lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU);
      l1_access   -- vector-L1 accesses per clock and CU against the 1.8 the walk's access pattern peaks at;
      wave_time   -- where the resident waves' time goes: parked on s_waitcnt (memory), stalled at issue, issuing (SQ_WAIT_ANY /
                     SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES)."""
    budget = max(120.0, 6.0 * (args.steps + args.warmup))
    sums, err = pmc_child_sums(args, PIPE_COUNTERS, budget)
    if sums is None:
        return None, err
    out = {}
    for cls, a in sums.items():
        n = a.get("launches", 0)
        clocks = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0      # per launch population: kernel clocks (one XCD's worth)
        if not n or clocks <= 0 or not a.get("SQ_INSTS_VALU"):
            continue
        simd_clocks = 4.0 * num_cus * clocks
        wave = a.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        out[cls] = {
            "launches": n,
            "clock_GHz": clocks / (a["ns"] * 1e-9) / 1e9 if a.get("ns") else None,
            "valu_issue": {"wave_instructions_per_launch": a["SQ_INSTS_VALU"] / n, "clocks_per_instruction": VALU_CLOCKS_PER_WAVE_INSTRUCTION,
                           "frac": a["SQ_INSTS_VALU"] * VALU_CLOCKS_PER_WAVE_INSTRUCTION / simd_clocks,
                           "lane_utilisation": a.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * a["SQ_INSTS_VALU"])},
            "l1_access": {"accesses_per_launch": a.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / n,
                          "per_clock_per_cu": a.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / (num_cus * clocks),
                          "peak_per_clock_per_cu": L1_DIVERGENT_ACCESSES_PER_CLOCK_PER_CU,
                          # the same microbenchmark at 5 waves per SIMD when the table does not fit the caches (profiles/r03_microbench_cadence_l1.txt): what the
                          # memory system sustains for this access pattern depends on where the nodes live -- the walk's 27 MB of nodes, triangles and leaf
                          # boxes sit between the 8 MB and the 64 MB row
                          "peak_by_table_size_at_5_waves": {"1 MB (L2)": 1.86, "8 MB": 1.24, "64 MB (Infinity Cache)": 0.70, "1 GB (HBM)": 0.62},
                          "frac": a.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / (num_cus * clocks) / L1_DIVERGENT_ACCESSES_PER_CLOCK_PER_CU,
                          # ... and interpolated at the bytes THIS scene's walk fetches from (rtgpu_get_walk_info): the ceiling that applies
                          "walk_footprint_bytes": footprint_bytes,
                          "peak_at_footprint": fetch_peak_at(footprint_bytes) if footprint_bytes else None,
                          "frac_at_footprint": (a.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / (num_cus * clocks) / fetch_peak_at(footprint_bytes)) if footprint_bytes else None},
            "wave_time": {"waiting_for_memory": a.get("SQ_WAIT_ANY", 0.0) / wave, "issue_stalled": a.get("SQ_WAIT_INST_ANY", 0.0) / wave,
                          "issuing": a.get("SQ_ACTIVE_INST_ANY", 0.0) / wave},
        }
    return out, None


# FETCH_SIZE / WRITE_SIZE on gfx950, calibrated on known byte counts in THIS library's access patterns (tools/microbench/fetch_calib.hip over a 2 GB table,
# profiles/r05_fetch_calibration.json): FETCH_SIZE = fabric read requests x 64 B whatever their size.  A wide coalesced stream (16 B per lane, consecutive)
# makes 128-byte requests and is reported at HALF its bytes -- the guide's factor 2; a random 64-byte node (four 16-byte loads of one half line), a random
# 16-byte load and a random 72-byte triangle pair make 64-byte requests and are reported at 1.00 / 1.00 / 1.05 of the lines they touch.  WRITE_SIZE reports
# coalesced 16-byte stores exactly and a random 16-byte store as 32 bytes.  So: the traversal classes (random node / triangle / box fetches; their streamed
# ray records corrected separately from the walk's own counts) take FETCH_SIZE as it is, the streaming classes (path records) take it doubled.
CALIBRATION = {
    "source": "profiles/r05_fetch_calibration.json (tools/microbench/fetch_calib.hip, tools/fetch_calib.sh; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over known byte counts, 2 GB table)",
    "FETCH_SIZE_true_bytes_per_reported_byte": {"coalesced 16 B/lane stream": 2.0, "random 64-byte node (4 x 16 B)": 1.0, "random 16 B load (per 64-byte line touched)": 1.0,
                                                "random 72-byte triangle pair (per 64-byte line touched)": 1.05},
    "WRITE_SIZE_true_bytes_per_reported_byte": {"coalesced 16 B/lane stores": 1.0, "random 16 B store": 0.5},
    "fetch_factor_by_class": {"trace": 1.0, "retrace": 1.0, "tail": 1.0, "shade": 2.0, "generate": 2.0, "accumulate": 2.0},
}


def measure_traffic(args):
    """Fabric-side bytes per launch of every kernel class from two separate --pmc child runs (the guide's recipe: FETCH_SIZE and WRITE_SIZE do not fit
    one pass; rocprofv3 reports both in KiB), corrected per kernel class by CALIBRATION.  Infinity-Cache hits are included in both counters (they are L2 <->
    fabric requests): an upper bound of the DRAM bytes."""
    budget = max(120.0, 6.0 * (args.steps + args.warmup))
    fetch, err_f = pmc_child_sums(args, "FETCH_SIZE", budget)
    write, err_w = pmc_child_sums(args, "WRITE_SIZE", budget)
    if fetch is None or write is None:
        return None, err_f or err_w
    out = {}
    for cls in fetch:
        if cls in write and fetch[cls][1] == write[cls][1] and fetch[cls][1] > 0:
            n = fetch[cls][1]
            factor = CALIBRATION["fetch_factor_by_class"].get(cls, 2.0)
            out[cls] = {"launches": n, "fetch_size_bytes": 1024.0 * fetch[cls][0] / n, "write_size_bytes": 1024.0 * write[cls][0] / n, "fetch_factor": factor,
                        "hbm_bytes": (factor * 1024.0 * fetch[cls][0] + 1024.0 * write[cls][0]) / n,
                        # rocprofv3's own kernel durations of the same launches: the shorter of the two counter passes (collecting counters slows a kernel by
                        # up to ~3 % on these boxes; a trace without counters -- profiles/rNN_kernel_stats_serial.txt -- agrees with the HIP events within 1 %)
                        "profiled_avg_launch_ms": (min(x for x in (fetch[cls][2], write[cls][2]) if x) / n / 1e6) if (fetch[cls][2] or write[cls][2]) else None}
    return out, None


def walk_byte_model(args):
    """The bytes the 4-wide walk ITSELF asks for (round-4 review, item 2b): a child process renders this command's warm-up + timed passes on one batch lane
    with the walk's diagnostic instantiation (RTGPU_WIDE_DIAG=3: every bounce through k_trace_wide, which counts interior visits, leaf visits, exact-box
    fetches and hit records written through), priced at what the kernel loads per event:
      interior visit 64 B (one node = four 16-byte child records), leaf visit 72 B (the leaf's two triangle slots), exact box 32 B, a closest-hit ray's
      refill 32 B (origin + direction records), an any-hit ray's 36 B (queue entry + shading point + direction / length records), a hit record 20 B.
    (Not counted: the 16 / 4 bytes a ray writes when it ends without a hit / occluded, the 4-byte entries of the few rays handed to the re-trace.)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--walk-diag-child", "--steps", str(args.steps), "--warmup", str(args.warmup), "--width", str(args.width),
           "--height", str(args.height), "--depth", str(args.depth), "--triangles", str(args.triangles), "--workload", args.workload]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, RTGPU_WIDE_DIAG="3", RTGPU_LANES="1"), capture_output=True, text=True, timeout=max(120.0, 6.0 * (args.steps + args.warmup)))
        lines = [l for l in r.stdout.splitlines() if l.startswith('{"walk_diag"')]
        if r.returncode != 0 or not lines:
            return None, "walk-diag child failed (rc %d): %s" % (r.returncode, (r.stderr or "")[-300:])
        d = json.loads(lines[-1])["walk_diag"]
        if not d["interior_visits"]:
            return None, "this scene's launches are not served by k_trace_wide (no diagnostic counts)"
        # sanity of the device counts (64-bit slots each since round 6): an exact box is fetched inside a leaf visit, a hit record is written behind an exact box
        if not (d["hit_records_written"] <= d["exact_box_fetches"] <= d["leaf_visits"]):
            return None, "walk-diag counters inconsistent (hit records %d, exact boxes %d, leaf visits %d)" % (d["hit_records_written"], d["exact_box_fetches"], d["leaf_visits"])
        per_event = {"interior_visits": 64, "leaf_visits": 72, "exact_box_fetches": 32, "closest_rays": 32, "shadow_rays": 36, "hit_records_written": 20}
        d["bytes_per_event"] = per_event
        d["bytes"] = sum(per_event[k] * d[k] for k in per_event)
        rays = d["closest_rays"] + d["shadow_rays"]
        d["per_ray"] = {"interior_visits": d["interior_visits"] / max(1, rays), "leaf_visits": d["leaf_visits"] / max(1, rays), "bytes": d["bytes"] / max(1, rays)}
        return d, None
    except subprocess.TimeoutExpired:
        return None, "walk-diag child timed out"
    except Exception as e:
        return None, "walk-diag child: %r" % (e,)


def work_changing_env():
    """Every environment switch that changes what a bench line measured: the bench's own aids (BENCH_*) and any RTGPU_* knob of the library that is
    set -- a line produced under one of them must say so (config.env), and an emulated shard is not the whole frame (config.emulated_shard, metric)."""
    return {k: os.environ[k] for k in sorted(os.environ) if k.startswith(("BENCH_", "RTGPU_")) or k in ("GPU_MAX_HW_QUEUES",)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if args.workload == "bdpt-glass" and world > 1:
        raise SystemExit("the bidirectional integrator needs the whole frame on one device (light paths splat anywhere): replicas only, run with --gpus 1")
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    # (N > 1, RCCL asked for) the side store of the backend agreement, before rank 0's build keeps the other ranks waiting
    side_store = open_side_store(dist) if (world > 1 and os.environ.get("BENCH_DIST_BACKEND", "nccl") == "nccl") else None
    if rank == 0 and not args.pmc_child and not args.walk_diag_child:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):   # stdout carries the one JSON line and nothing else
            entry.build()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one rank per GPU; BENCH_DIST_BACKEND=gloo with ranks sharing a device exists only to exercise the N>1 code path
    # on a 1-GPU box (RCCL refuses two ranks on one device)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver supports dmabuf IPC only: RCCL's peer buffers need it (exported on the GPU boxes; kept here for any other launcher)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    backend_fallback = None
    if local_rank >= torch.cuda.device_count():
        # fewer visible devices than ranks: BENCH_DIST_BACKEND=gloo on a 1-GPU box (the N > 1 code path exercised with ranks sharing a device), or a launcher
        # that shows every rank only its own device (HIP_VISIBLE_DEVICES per rank: the rank's device is then index 0)
        sys.stderr.write("[bench] rank %d: local rank %d but %d visible device(s): using device %d\n" % (rank, local_rank, torch.cuda.device_count(), local_rank % torch.cuda.device_count()))
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    init_s = 0.0
    if world > 1:
        t_init = time.perf_counter()
        backend, backend_fallback = init_process_group(dist, backend, torch.device("cuda", local_rank), side=side_store)
        dist.barrier()
        torch.cuda.synchronize()
        init_s = time.perf_counter() - t_init    # communicator set-up + the first barrier: outside the timed region, reported per rank
        props = torch.cuda.get_device_properties(local_rank)
        try:
            peer0 = bool(torch.cuda.can_device_access_peer(local_rank, 0)) if local_rank != 0 else True
        except Exception:
            peer0 = None
        sys.stderr.write("[bench] rank %d / %d: device %d (%s, %d CUs, %.0f GB), peer access to device 0: %s, backend %s, process group up in %.2f s\n"
                         % (rank, world, local_rank, props.name, props.multi_processor_count, props.total_memory / 2**30, peer0, backend, init_s))
    import raytracer_amd as ra

    w, h = args.width, args.height
    scene, camera = build_scene(args, w / h)
    emulate = int(os.environ.get("BENCH_EMULATE_SHARD", "0"))   # tuning aid on a 1-GPU box: only the tiles rank 0 of N would own
    shard = (rank, world) if world > 1 else ((0, emulate) if emulate > 1 else None)
    vp = make_viewport(ra, args, scene, local_rank, shard)
    lib = ra.rtgpu_lib()
    host = ra.host_lib()
    ctx = vp.device_context()

    if args.pmc_child or args.walk_diag_child:
        # the profiled child: exactly the parent's passes, strictly serial kernels, nothing else
        lib.rtgpu_set_concurrency(ctx, 1)
        lib.rtgpu_set_intersection_counters(ctx, 0)
        vp.render(camera, args.warmup)
        vp.counters()            # the parent's replay reads the counters here (a synchronising call: the pass batches start over)
        vp.render(camera, args.steps)
        lib.rtgpu_synchronize(ctx)
        if args.walk_diag_child:
            # RTGPU_WIDE_DIAG=3 (set by the parent): the 4-wide walk counted its own fetches in the spare counters (rt_trace_wide.inl)
            c = vp.counters()
            print(json.dumps({"walk_diag": {"interior_visits": c["numUntrustedRays"], "leaf_visits": c["diag2"], "exact_box_fetches": c["numStackOverflowRays"],
                                            "hit_records_written": c["numShadowRayTriangleTests"],   # (the diagnostic walk's own 64-bit slot, rt_trace_wide.inl)
                                            "closest_rays": c["numRays"], "shadow_rays": c["numShadowRays"],
                                            "retraced_rays": c["numRetracedRays"]}}), flush=True)
        return

    def sync_all():
        lib.rtgpu_synchronize(ctx)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # every rank draws the same per-pass constants (same seed => same Halton / AA offsets).
    # The box / triangle test counters are instrumentation (a compile-time debug switch in the reference,
    # RT_ENABLE_INTERSECTION_COUNTERS, off by default): they are OFF in the timed region and collected afterwards
    # by replaying exactly the same passes (same seed => same rays) with the counters on.  Per-kernel event timing is off too.
    lib.rtgpu_set_intersection_counters(ctx, 0)
    vp.render(camera, args.warmup)
    host_sum = np.zeros((h, w, 3), dtype=np.float32)
    gather = None
    if world > 1:
        sum_ptr, sec_ptr, nfl = C.c_void_p(), C.c_void_p(), C.c_size_t()
        lib.rtgpu_get_device_sum(ctx, C.byref(sum_ptr), C.byref(sec_ptr), C.byref(nfl))
        gather = TileGather(torch, dist, w, h, rank, world, device_tensor(sum_ptr.value, nfl.value, torch))
        sync_all()
        t_warm = time.perf_counter()
        gather.choose_mode()    # warm-up of the collective (channel set-up) outside the timed region, and the decision gather / grouped send-recv
        gather.run()            # (it moves the warm-up image)
        torch.cuda.synchronize()
        gather_warmup_s = time.perf_counter() - t_warm
    if rank == 0:
        # warm-up of the read-back: the FIRST Viewport::GetSumBuffer registers the viewport's bitmaps as page-locked memory (1.6-3.5 ms, once per viewport;
        # profiles/r05_readback_cost.txt) -- a later one is the copy alone (~0.45 ms for 24.9 MB).  Rounds 1-5 had that first call inside the timed region.
        host.rth_viewport_fetch_sum(vp._h)
    sync_all()
    c0 = vp.counters()

    sync_all()
    t0 = time.perf_counter()
    vp.render(camera, args.steps)
    gather_s = 0.0
    if world > 1:
        lib.rtgpu_synchronize(ctx)
        render_s = time.perf_counter() - t0     # this rank's own passes (the ranks finish at different times: load balance)
        gather.run()
        torch.cuda.synchronize()
        gather_s = time.perf_counter() - t0 - render_s   # on rank 0 this includes waiting for the slowest peer
    sync_all()                               # every rank's passes have run and rank 0's sum buffer (HBM) holds the whole frame
    elapsed_hbm = time.perf_counter() - t0   # reported beside `value` (frame_in_hbm): the region without the PCIe copy
    if rank == 0:
        host.rth_viewport_fetch_sum(vp._h)   # Viewport::GetSumBuffer = the final read_sum of SURVEY 8(d): the frame is in the viewport's (page-locked) host bitmap afterwards
    elapsed = time.perf_counter() - t0       # the timed region of SURVEY 8(d): render_pass x K (+ gather) + the final read_sum
    readback_s = elapsed - elapsed_hbm
    if rank == 0:
        host.rth_viewport_read_sum(vp._h, host_sum.ctypes.data_as(C.POINTER(C.c_float)), None)   # a copy of that bitmap for the checks below

    c1 = vp.counters()
    delta = {k: c1[k] - c0[k] for k in c1}
    image_ok = bool(np.isfinite(host_sum).all()) if rank == 0 else True
    image_mean = float(host_sum.mean()) / max(1, args.steps + args.warmup) if rank == 0 else 0.0

    if os.environ.get("BENCH_TIMED_ONLY"):   # profiling aid (tools/concurrency.py): the trace ends with the timed region, no replays behind it
        if rank == 0:
            print(json.dumps({"value": delta["numRays"] / elapsed / 1e6, "unit": "Msamples/s", "ms_per_step": 1000.0 * elapsed / max(1, args.steps), "timed_only": True,
                              "host_readback_ms": 1000.0 * readback_s, "value_frame_in_hbm": delta["numRays"] / elapsed_hbm / 1e6,
                              "ms_per_step_frame_in_hbm": 1000.0 * elapsed_hbm / max(1, args.steps), "value_definition": VALUE_DEFINITION,
                              "config": {"env": work_changing_env(), "emulated_shard": [0, emulate] if emulate > 1 else None}}))
        return
    def kernel_times(context):
        ms = (C.c_double * 8)(); launches = (C.c_uint64 * 8)(); names = (C.c_char_p * 8)()
        lib.rtgpu_get_kernel_times(context, ms, launches, names)
        return {names[i].decode(): (ms[i], int(launches[i])) for i in range(8) if names[i]}

    def replay(lanes, intersection_counters, timing):
        """Renders exactly the same passes again (same seed => same rays) on a fresh viewport; returns the counter
        deltas of the `steps` passes, the counter totals of warm-up + steps and, if asked, the per-kernel-class HIP-event
        times of warm-up + steps (every launch of the replay: the population the --pmc children and
        `RTGPU_LANES=1 rocprofv3 --kernel-trace --stats` average over)."""
        v = make_viewport(ra, args, scene, local_rank, shard)
        vctx = v.device_context()
        lib.rtgpu_set_concurrency(vctx, lanes)
        lib.rtgpu_set_intersection_counters(vctx, 1 if intersection_counters else 0)
        lib.rtgpu_enable_timing(vctx, 1 if timing else 0)
        v.render(camera, args.warmup)
        a0 = v.counters()
        v.render(camera, args.steps)
        a1 = v.counters()
        times = kernel_times(vctx) if timing else None
        return {k: a1[k] - a0[k] for k in a1}, dict(a1), times

    serial, _, ktimes = replay(1, False, True)
    assert serial["numRays"] == delta["numRays"] and serial["numShadowRays"] == delta["numShadowRays"], "serial replay diverged from the timed run"
    # instrumented replay for the intersection counters (not timed)
    counted, counted_totals, _ = replay(1, True, False)
    assert counted["numRays"] == delta["numRays"] and counted["numShadowRays"] == delta["numShadowRays"], "replay diverged from the timed run"
    for k in ("numRayBoxTests", "numPassedRayBoxTests", "numRayTriangleTests", "numPassedRayTriangleTests", "numShadowRayBoxTests",
              "numShadowRayTriangleTests"):
        delta[k] = counted[k]
    own_counts = dict(delta)

    scaling_report = None
    if world > 1:
        # what a first measured curve needs to explain itself: every rank's share of the work and of the time, the exchange, and the proof that the
        # assembled frame is the one-GPU frame
        mine = torch.tensor([float(own_counts["numRays"]), float(own_counts["numShadowRays"]), render_s, gather_s, elapsed, init_s, gather_warmup_s, float(local_rank),
                             float(torch.cuda.can_device_access_peer(local_rank, 0)) if local_rank != 0 else 1.0], dtype=torch.float64, device="cuda")
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        per_rank = [t.tolist() for t in per_rank]
        scaling_report = {
            "per_rank_numRays": [int(r[0]) for r in per_rank], "per_rank_numShadowRays": [int(r[1]) for r in per_rank],
            "per_rank_render_ms": [round(1000.0 * r[2], 3) for r in per_rank], "per_rank_gather_ms": [round(1000.0 * r[3], 3) for r in per_rank],
            "per_rank_timed_region_ms": [round(1000.0 * r[4], 3) for r in per_rank],
            "load_imbalance_numRays": max(r[0] for r in per_rank) / (sum(r[0] for r in per_rank) / world),
            "gather_bytes_per_peer": int(gather.pad * 12),
            "per_rank_device": [int(r[7]) for r in per_rank], "per_rank_peer_access_to_device_0": [bool(r[8]) for r in per_rank],
            "per_rank_process_group_init_s": [round(r[5], 3) for r in per_rank], "per_rank_gather_warmup_s": [round(r[6], 3) for r in per_rank],
            "exchange": {"mode": gather.mode, "reason": gather.mode_reason, "backend": backend, "backend_fallback": backend_fallback},
        }
        if rank == 0:
            # the same warm-up + timed passes on ONE device, whole frame: the gathered frame must be that frame, bit for bit
            try:
                whole = make_viewport(ra, args, scene, local_rank, None)
                whole.render(camera, args.warmup + args.steps)
                one_gpu = whole.sum_buffer()
                del whole
                scaling_report["frame_check"] = {"equal_to_one_gpu_replay": bool(np.array_equal(one_gpu.view(np.uint32), host_sum.view(np.uint32))),
                                                 "mean_gathered": float(host_sum.mean()), "mean_one_gpu": float(one_gpu.mean()),
                                                 "max_abs_diff": float(np.abs(one_gpu - host_sum).max())}
            except Exception as e:   # the check must not take the measurement down
                scaling_report["frame_check"] = {"equal_to_one_gpu_replay": None, "error": repr(e)}
            if scaling_report["frame_check"]["equal_to_one_gpu_replay"] is False:
                # reported, not fatal: the line below still carries the measurement, with the failed check in it for whoever reads the curve
                sys.stderr.write("WARNING: the gathered frame differs from the one-GPU replay: %r\n" % (scaling_report["frame_check"],))
        t = torch.tensor([elapsed, elapsed_hbm], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_hbm = (float(x) for x in t.tolist())
        keys = sorted(delta)
        tc = torch.tensor([delta[k] for k in keys], dtype=torch.int64, device="cuda")
        dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        delta = {k: int(v) for k, v in zip(keys, tc.tolist())}

    if rank == 0:
        value = delta["numRays"] / elapsed / 1e6
        if args.workload == "bdpt-glass":
            label = "configs[4]: rough-glass slab over a diffuse ground under a rect light (materials_test.json style), renderer VCM with merging off (BDPT), max path length 8, %dx%d" % (w, h)
            metric = "Msamples/sec (path segments of camera + light sub-paths) at %dx%d, rough-glass BDPT" % (w, h)
        else:
            what = ("procedural Sponza-class mesh (%d triangles, 8 diffuse materials)" % scene.desc.contents.numTriangles) if args.workload.startswith("sponza") else args.workload
            which = {"sponza": "configs[2]", "sponza-all": "configs[2] (path-exact variant)", "sponza-textured": "configs[2] (textured variant)", "cornell": "configs[0] scene (Cornell box)",
                     "sphere": "configs[1]", "zoo": "test scene (every light x every BSDF; not a BASELINE config)"}.get(args.workload, args.workload)
            label = "%s: %s, PathTracerMIS, %d bounces, %dx%d, LightSamplingStrategy::%s" % (which, what, args.depth, w, h, "All, dimensions 128" if args.workload == "sponza-all" else "Single")
            if args.workload == "sponza-textured":
                label += " + albedo / normal maps on all materials, HDR environment map"
            metric = "Msamples/sec (paths x bounces) at %dx%d, Sponza-class PT-MIS" % (w, h)
        out = {
            "metric": metric, "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / max(1, args.steps), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "spp_timed": args.steps, "parallelism": "tile-interleaved x%d" % world,
                       "timed_region": "SURVEY 8(d): K x Viewport::Render + (N > 1) the gather of owned tiles to rank 0 + the final Viewport::GetSumBuffer (read_sum: 24.9 MB to page-locked host memory); frame_in_hbm is the region without that copy",
                       "value_definition": VALUE_DEFINITION,
                       # what changed this line's work, if anything: environment switches of the bench (BENCH_*) and of the library (RTGPU_*)
                       "env": work_changing_env(), "emulated_shard": [0, emulate] if emulate > 1 else None,
                       "dist_backend": backend if world > 1 else None},
            "counters": {k: delta[k] for k in ("numRays", "numPrimaryRays", "numShadowRays", "numRayBoxTests", "numRayTriangleTests",
                                               "numShadowRayBoxTests", "numShadowRayTriangleTests", "numMeshHits", "numAnalyticHits")},
            "mrays_per_s_incl_shadow": (delta["numRays"] + delta["numShadowRays"]) / elapsed / 1e6,
            "intersection_counters": "off in the timed region (reference default); counts from an identical instrumented replay",
            "image": {"finite": image_ok, "mean_per_pass": image_mean},
            # the region without its last step: the frame complete in rank 0's HBM (what BENCH_r05.json called `value`; rounds 1-4 and this round: read-back inside)
            "frame_in_hbm": {"value": delta["numRays"] / elapsed_hbm / 1e6, "ms_per_step": 1000.0 * elapsed_hbm / max(1, args.steps)},
            "host_readback": {"ms": 1000.0 * readback_s, "bytes": int(w) * int(h) * 12, "in_value": True},
        }
        if emulate > 1:
            # NOT the whole frame: the tiles rank 0 of `emulate` ranks would own, rendered alone on this device (a tuning aid for the N > 1 path)
            out["metric"] += " -- EMULATED SHARD: 1/%d of the frame (rank 0's tiles of %d), one device" % (emulate, emulate)
            out["config"]["parallelism"] = "emulated shard 0 of %d (tile-interleaved), 1 device" % emulate
        if scaling_report:
            out["multi_gpu"] = scaling_report
        # roofline of the dominant kernel class (rank 0's own launches and rank 0's own counters)
        abytes = algorithmic_bytes(own_counts)
        abytes_replay = algorithmic_bytes(counted_totals)   # warm-up + timed passes: what the replay's launches processed
        dom = max(ktimes, key=lambda k: ktimes[k][0]) if ktimes else None
        if dom and ktimes[dom][0] > 0:
            traffic, traffic_error = (None, "skipped (--no-pmc)") if (args.no_pmc or world > 1) else measure_traffic(args)
            launches = ktimes[dom][1]
            per_launch_bytes = abytes_replay[dom] / max(1, launches)
            per_launch_s = ktimes[dom][0] / 1000.0 / max(1, launches)
            algorithmic_gbs = per_launch_bytes / per_launch_s / 1e9
            kernel_name = "k_" + dom
            if dom == "trace" and c1.get("numRetracedRays", 0) > 0:
                kernel_name = "k_trace_wide"   # the 4-wide walk served the launches (it hands a few rays to k_trace: class "retrace"); bounce 0's launch of a batch is k_trace_packet, the same tree walked one 8 x 8 pixel block per wave (one launch in ten of the class)
            # Round 6 (review item 2): the block says what it measures.
            #   achieved / frac          the HBM roofline: calibrated fabric bytes per launch (`traffic`: FETCH_SIZE + WRITE_SIZE of two --pmc child runs of the same
            #                            launches) / average launch time, over 8 TB/s.  A fraction of a roof: 0 < frac <= 1 at any batch size (asserted by
            #                            tests/test_gpu_bench_plumbing.py at 20 and at 256 passes); null without PMC (--no-pmc, N > 1).
            #   request_rate_*           the ALGORITHMIC bytes the kernel asks for / launch time.  trace served by k_trace_wide: the walk's OWN fetches, counted on the
            #                            device by its diagnostic instantiation (walk_byte_model); binary walk / shade / accumulate: SURVEY 8(d)'s per-unit figures x the
            #                            replay's counters; classes without a byte model (tail, retrace, generate): "n/a".  These are requests, answered by L1 / L2 /
            #                            Infinity Cache four times out of five -- not bounded by the HBM peak (1.06 x at 256 passes), so the field is not called frac.
            #   reference_walk_*         SURVEY 8(d)'s literal model (32 B per box test + 36 B per triangle test of the REFERENCE'S binary walk): a work measure,
            #                            non-physical as a rate (> 1 x peak: this kernel walks another tree -- 17 visits where the binary walk makes 29 -- out of the caches).
            #   bound / ceilings / frac_of_binding_ceiling   which measured ceiling (HBM, VALU issue, cache fetch) the kernel is nearest to, and the counters behind it.
            roof = {"bound": "hbm", "roof": "hbm", "kernel": kernel_name, "peak": HBM_PEAK_GBS, "unit": "GB/s", "definition": "r6: frac = measured HBM traffic fraction (BENCH_r05's frac is request_rate_over_hbm_peak here)",
                    "avg_launch_ms": per_launch_s * 1000.0, "launches": launches,
                    "reference_walk_bytes_per_launch": per_launch_bytes, "reference_walk_GBs": algorithmic_gbs,
                    "calibration": CALIBRATION,
                    "measured": "launch time: HIP events around every launch of a one-lane (serial kernels) replay of the warm-up and timed passes; "
                                "algorithmic bytes: the walk's own event counts (a child run of the same passes with its diagnostic instantiation) x bytes per event; "
                                "traffic: two rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE) of the same passes, fetch_factor x FETCH_SIZE + WRITE_SIZE per launch "
                                "(factor per kernel class from the calibration run; the traversal class adds the uncounted half of its streamed ray records); "
                                "ceilings: a third child run (SQ / GRBM / TCP counters), cadence and L1 peak from tools/microbench/cadence.hip"}
            t = traffic.get(dom) if traffic else None
            if t and t["launches"] != launches and t["profiled_avg_launch_ms"]:
                # a kernel class whose event pairs span several kernels (the bidirectional integrator's shading stages): bytes and time both
                # from the profiled child, which is one self-consistent population
                roof["avg_launch_ms"] = t["profiled_avg_launch_ms"]; roof["launches"] = launches = t["launches"]
                per_launch_s = t["profiled_avg_launch_ms"] / 1000.0
                per_launch_bytes = abytes_replay[dom] / max(1, launches)
                roof["reference_walk_bytes_per_launch"] = per_launch_bytes
                roof["measured"] += "; this class's event pairs span several kernels, so launch time and count are the profiled child's"
            walk_model, walk_error = (None, "skipped (--no-pmc)")
            if dom == "trace" and kernel_name == "k_trace_wide" and not args.no_pmc and world == 1:
                walk_model, walk_error = walk_byte_model(args)
            if walk_model:
                algorithmic = walk_model["bytes"] / max(1, launches)
                roof["algorithmic_model"] = walk_model
            elif dom in ("tail", "retrace", "generate") or abytes_replay.get(dom, 0) == 0 or (dom == "trace" and kernel_name == "k_trace_wide"):
                # (the 4-wide walk without its own counts -- --no-pmc, N > 1 -- claims nothing: SURVEY 8(d)'s model of the reference's BINARY walk is not what it fetches)
                algorithmic = None
                roof["algorithmic_model"] = "n/a: class '%s' has no byte model here (%s)" % (dom, walk_error)
            else:
                algorithmic = per_launch_bytes
                roof["algorithmic_model"] = "SURVEY 8(d) per-unit bytes x the replay's counters (%s)" % (walk_error if dom == "trace" else "class " + dom)
            roof["algorithmic_bytes_per_launch"] = algorithmic
            roof["request_rate_GBs"] = algorithmic / per_launch_s / 1e9 if algorithmic else None
            roof["request_rate_over_hbm_peak"] = roof["request_rate_GBs"] / HBM_PEAK_GBS if algorithmic else None
            roof["request_rate_reads_as"] = ("algorithmic bytes the lanes ask for per second over the HBM peak; the caches answer most of them (traffic_over_algorithmic), "
                                             "so this may exceed 1 -- it is not a fraction of a roof; `frac` is")
            roof["reference_walk_frac"] = algorithmic_gbs / HBM_PEAK_GBS
            roof["reference_walk_note"] = ("SURVEY 8(d) literal: 32 B x box tests + 36 B x triangle tests of the reference's BINARY walk per launch / launch time / 8 TB/s; "
                                           "non-physical (> 1 possible): the timed kernel walks a different, cache-resident tree")
            roof["achieved"] = None; roof["frac"] = None   # the HBM roofline proper: set from the PMC traffic below, null without it
            # compulsory_*: the bytes one launch cannot avoid moving across the fabric -- each ray's record in, each hit record out, the walked tree's
            # footprint once -- the floor `traffic` is to be read against (above it: nodes re-fetched because 22 MB of tree do not fit a 4 MB L2).
            if t and t["launches"] == launches:
                hbm_bytes = t["hbm_bytes"]
                if walk_model:
                    # the ray records a refill reads are a coalesced stream inside a class whose FETCH_SIZE is otherwise taken as it is: add the half the counter missed
                    hbm_bytes += 0.5 * (32.0 * walk_model["closest_rays"] + 32.0 * walk_model["shadow_rays"]) / max(1, launches)
                roof.update({"traffic": hbm_bytes, "traffic_GBs": hbm_bytes / per_launch_s / 1e9, "traffic_frac": hbm_bytes / per_launch_s / 1e9 / HBM_PEAK_GBS,
                             "traffic_over_algorithmic": hbm_bytes / algorithmic if algorithmic else None,
                             "traffic_fetch_size_bytes": t["fetch_size_bytes"], "traffic_write_size_bytes": t["write_size_bytes"], "traffic_fetch_factor": t["fetch_factor"],
                             "profiled_avg_launch_ms": t["profiled_avg_launch_ms"]})
                roof["achieved"] = roof["traffic_GBs"]; roof["frac"] = roof["traffic_frac"]
                roof["frac_is"] = "traffic / avg launch time / 8 TB/s: calibrated FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes) of the same launches -- the share of the HBM roof in use"
                roof["traffic_over_reference_walk_bytes"] = hbm_bytes / per_launch_bytes if per_launch_bytes else None
                out["traffic_per_launch"] = {k: {"launches": v["launches"], "hbm_bytes": v["hbm_bytes"],
                                                 "GBs": (v["hbm_bytes"] / (ktimes[k][0] / 1000.0 / max(1, ktimes[k][1])) / 1e9) if k in ktimes and ktimes[k][0] > 0 else None,
                                                 "reference_walk_bytes": abytes_replay.get(k, 0) / max(1, v["launches"])}
                                             for k, v in traffic.items()}
                class WalkInfo(C.Structure):
                    _fields_ = [("kernel", C.c_uint32), ("reserved", C.c_uint32), ("nodeBytes", C.c_uint64), ("leafBoxBytes", C.c_uint64), ("triangleBytes", C.c_uint64)]
                wi = WalkInfo()
                lib.rtgpu_get_walk_info(ctx, C.byref(wi))
                footprint = int(wi.nodeBytes + wi.leafBoxBytes + wi.triangleBytes) if dom in ("trace",) else None
                out["walk"] = {"kernel": ["k_trace (binary tree)", "k_trace_wide", "k_trace_wide2"][wi.kernel], "node_bytes": int(wi.nodeBytes),
                               "leaf_box_bytes": int(wi.leafBoxBytes), "triangle_bytes": int(wi.triangleBytes)}
                if walk_model and footprint:
                    compulsory = (32.0 * walk_model["closest_rays"] + 36.0 * walk_model["shadow_rays"] + 20.0 * walk_model["hit_records_written"]) / max(1, launches) + footprint
                    roof["compulsory_bytes_per_launch"] = compulsory
                    roof["traffic_over_compulsory"] = hbm_bytes / compulsory
                pipes, pipes_error = measure_pipes(args, torch.cuda.get_device_properties(local_rank).multi_processor_count, footprint)
                if pipes and dom in pipes:
                    pd = pipes[dom]
                    fracs = {"hbm": roof["traffic_frac"], "valu_issue": pd["valu_issue"]["frac"],
                             "cache_fetch": pd["l1_access"]["frac_at_footprint"] if pd["l1_access"]["frac_at_footprint"] is not None else pd["l1_access"]["frac"]}
                    nearest = max(fracs, key=lambda k: fracs[k])
                    # cache_fetch: the kernel's vector-L1 accesses per clock and CU over what the divergent-fetch microbenchmark sustains from a table of
                    # the walk's footprint -- a throughput ceiling of L1 / L2 / Infinity Cache for one-node-per-lane fetches (more waves per SIMD do not
                    # raise it; 3, 4 or 5 traversal blocks per CU give the same end-to-end rate, profiles/r03_streaming_hints_and_sweeps.txt).  The launch
                    # average includes the refill, leaf and drain phases; the interior loop alone (52 % of the wave time, ~80 % of the accesses) runs at it.
                    # Below 0.6 of every ceiling the kernel is called latency-bound.
                    roof["bound"] = nearest if fracs[nearest] >= 0.6 else "latency"
                    roof["ceilings"] = {"fracs": fracs, "nearest": nearest, **pd}
                    roof["frac_of_binding_ceiling"] = fracs[nearest]
                    roof["binding_ceiling"] = {"name": nearest, "counters": {
                        "hbm": "FETCH_SIZE, WRITE_SIZE (calibrated per kernel class) / launch time / 8 TB/s",
                        "valu_issue": "SQ_INSTS_VALU x 2.1 clocks / (4 SIMDs x CUs x GRBM_GUI_ACTIVE / 8 XCDs); lane utilisation SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU)",
                        "cache_fetch": "TCP_TOTAL_CACHE_ACCESSES_sum / (CUs x GRBM_GUI_ACTIVE / 8) over the divergent-fetch microbenchmark's rate at the walk's footprint"}[nearest]}
                    out["pipes_per_kernel_class"] = {k: {"valu_issue_frac": v["valu_issue"]["frac"], "lane_utilisation": v["valu_issue"]["lane_utilisation"],
                                                         "l1_access_frac": v["l1_access"]["frac"], "waiting_for_memory": v["wave_time"]["waiting_for_memory"]}
                                                     for k, v in pipes.items()}
                else:
                    roof["ceilings_error"] = pipes_error or "no counters for the dominant kernel class"
            else:
                roof.update({"traffic": None, "traffic_error": traffic_error or ("launch population mismatch: %s" % (t,))})
            out["roofline"] = roof
            out["kernel_time_ms"] = {k: round(v[0], 3) for k, v in ktimes.items()}
            out["kernel_launches"] = {k: v[1] for k, v in ktimes.items()}
            tot_bytes = sum(abytes.values())
            out["whole_pass_reference_walk_GBs"] = tot_bytes / elapsed / 1e9

        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, scene, camera, ra)
        line = json.dumps(out)
    else:
        line = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(line, flush=True)   # after the process group is gone: nothing the communication library prints can follow it


def cpu_baseline(args, scene, camera, ra):
    """CPU baseline on the GPU box's host cores, on a BOUNDED sample of the same workload.
    Preferred: oracle/_ref/ref_render -- the reference's own AVX2/FMA translation units (traversal, intersection, shapes, BSDFs, lights,
    sampler, math: everything a ray does) under a restated pass loop (kind "reference-partial", see oracle/ref_harness).  Otherwise the
    scalar CPU restatement of the algorithm (kind "port")."""
    ref = reference_baseline(args, scene, camera, ra)
    if ref is not None:
        return ref
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    w, h = args.width, args.height
    cores = os.cpu_count() or 1
    desc = scene.desc
    bn = ra.load_blue_noise()
    desc.contents.blueNoise = bn.ctypes.data
    if args.workload == "bdpt-glass":
        # the oracle's VertexConnectionAndMerging restatement is single-threaded (film splats are ordered): a band of tiles of one pass
        from raytracer_amd import scenes
        vp = ra.Viewport(w, h, seed=77)
        vcm = oracle_lib.Vcm(**scenes.ROUGH_GLASS_SLAB_VCM)
        img = np.zeros((h, w, 3), dtype=np.float32); light = np.zeros((h, w, 3), dtype=np.float32)
        cnt = np.zeros(16, dtype=np.uint64)
        p = vp.next_pass_params(camera)
        t0 = time.perf_counter()
        vcm.render_pass(desc, p, w, h, img, None, light, cnt, shard=(0, 64))
        dt = time.perf_counter() - t0
        return {"value": int(cnt[0]) / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port", "sample": "1/64 of the tiles of one pass (single thread)",
                "seconds": round(dt, 2), "numRays": int(cnt[0])}
    vp = ra.Viewport(w, h, seed=77, max_ray_depth=args.depth)
    img = np.zeros((h, w, 3), dtype=np.float32)
    cnt = np.zeros(16, dtype=np.uint64)
    # calibration on a thin band (every 16th 64x64 tile), then whole passes within the budget
    p = vp.next_pass_params(camera)
    t0 = time.perf_counter()
    oracle_lib.render_pass(desc, p, w, h, img, None, cnt, shard=(0, 16), threads=cores)
    t_band = time.perf_counter() - t0
    rays_band = int(cnt[0])
    rate = rays_band / t_band
    sample = "1/16 of the tiles of one pass"
    total_rays, total_t = rays_band, t_band
    est_pass = t_band * 16.0
    passes = int(max(0.0, args.cpu_seconds - t_band) // max(est_pass, 1e-9))
    if passes >= 1:
        passes = min(passes, 4)
        cnt[:] = 0
        t0 = time.perf_counter()
        for _ in range(passes):
            p = vp.next_pass_params(camera)
            oracle_lib.render_pass(desc, p, w, h, img, None, cnt, threads=cores)
        total_t = time.perf_counter() - t0
        total_rays = int(cnt[0])
        rate = total_rays / total_t
        sample = "%d full pass(es) of the %dx%d frame" % (passes, w, h)
    return {"value": rate / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port", "sample": sample,
            "seconds": round(total_t, 2), "numRays": total_rays}


def reference_baseline(args, scene, camera, ra):
    """oracle/_ref/ref_render (built by oracle/ref_harness from the reference's own sources where /root/reference exists; the binary
    travels, the sources do not).  Returns None when it is not there or does not support the workload."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_render")
    # every PathTracerMIS workload: the reference's objects render textured materials, an environment map and LightSamplingStrategy::All too
    # (bdpt-glass: VertexConnectionAndMerging.cpp does not build here)
    if not os.path.exists(exe) or args.workload not in ("sponza", "sponza-textured", "sponza-all", "cornell", "sphere", "zoo"):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import ref_render
        sampling_all = args.workload == "sponza-all"
        return ref_render.timed_baseline(exe, args, scene, camera, ra, dimensions=128 if sampling_all else 64, light_sampling_all=sampling_all)
    except Exception as e:
        sys.stderr.write("reference baseline unavailable: %r\n" % (e,))
        return None


if __name__ == "__main__":
    main()
