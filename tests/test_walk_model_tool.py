"""tools/wide8 -- the CPU step model of the traversal formats (round 6) -- stays buildable and self-consistent: on a small mesh it must reproduce the
facts the round's decision rested on (the numbers for the benchmark mesh are in profiles/r06_wide8_step_model.txt):
  * every tree format and visiting order finds the oracle's hit distance for (nearly) every recorded closest-hit ray -- the model walks real trees;
  * an 8-wide collapse of the same binary tree makes fewer node visits than the 4-wide one, but not the third the paper estimate assumed;
  * any-hit rays are cheaper FARTHEST child first than nearest first (what the product kernels do since round 6).
CPU only (the oracle records the rays); test infrastructure, nothing of it is on the product path."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_step_model_builds_and_reproduces_the_orders(built, tmp_path):
    out = str(tmp_path / "walk")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wide8", "dump_walk_inputs.py"), out, "61", "6000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    exe = str(tmp_path / "walk_model")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "wide8", "walk_model.cpp"), "-o", exe])
    r = subprocess.run([exe, out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    variants = {}
    name = None
    for line in r.stdout.splitlines():
        if line and not line.startswith(" ") and ":" in line and line[0] == "W":
            name = line.split(":")[0]
        m = re.match(r"\s+(closest-hit|any-hit):\s+interior\s+([\d.]+) per ray.*?leaf visits\s+([\d.]+).*?(?:hit distance differs from the oracle's for (\d+) rays)?$", line)
        if m and name:
            variants.setdefault(name, {})[m.group(1)] = (float(m.group(2)), float(m.group(3)), int(m.group(4)) if m.group(4) else None)
    rays = int(re.search(r"rays: (\d+) closest-hit", r.stdout).group(1))
    assert rays > 500 and len(variants) >= 10, (rays, sorted(variants))
    w4 = variants["W4  16-bit planes, distance order (round 5)"]
    w8 = variants["W8  8-bit planes, octant order (the 64-byte node)"]
    far = variants["W4  16-bit planes, FARTHEST child first"]
    # the model's walks are real: at most a handful of rays disagree with the oracle's hit distance (ties between coplanar neighbours)
    for name, v in variants.items():
        if v["closest-hit"][2] is not None:
            assert v["closest-hit"][2] <= max(3, rays // 200), (name, v["closest-hit"][2], rays)
    assert 0.55 < w8["closest-hit"][0] / w4["closest-hit"][0] < 0.95, (w8, w4)      # fewer fetches, not a third fewer
    assert far["any-hit"][0] < 0.85 * w4["any-hit"][0] and far["any-hit"][1] < w4["any-hit"][1], (far, w4)   # the order the kernels walk any-hit rays in
    assert far["closest-hit"][0] > w4["closest-hit"][0]                              # ... and why closest-hit rays keep nearest first
