"""Soak test: the device (library defaults: dense path state, the 4-wide walks, four lanes, intersection counters off in half of the cases)
against the CPU oracle on random scenes, cameras, sizes and settings -- bit-identical sum buffers and ray counters expected.
usage: python tools/oracle_fuzz.py [seconds] [seed] [kinds] [only this scene kind]      (tests/test_gpu_fuzz.py runs a bounded slice of it under `pytest -m gpu`)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import scenes
import oracle_lib, scene_zoo


def run(budget=300.0, seed=1, min_cases=0, log=print, kinds=8, only=None):
    """Random cases until `budget` seconds are used up (and at least `min_cases`); returns (cases, mismatches, default-walk cases).  The case stream and the
    rendering of a case live in tools/oracle_fuzz_replay.py, which can replay any single case of a seed (`kinds`: 7 since the end of round 6 -- the zoo of every
    light x every BSDF joined the five scene kinds of rounds 1-6, and the two-level scenes get random cameras too; 8: tests/scene_zoo.py random_scene as a seventh scene
    kind; a mismatch line names seed, index and kinds).  `only`: render just the cases of that scene kind ("random", "zoo", ...)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import oracle_fuzz_replay as replay
    t_end = time.time() + budget
    cases = bad = default_walk = 0
    for case in replay.stream(seed, None, kinds):
        if not (time.time() < t_end or cases < min_cases):
            break
        if only and case[4][0] != only:
            continue
        same, differing = replay.render(case, quiet=True)
        cases += 1
        default_walk += 0 if case[8] else 1
        if not same:
            bad += 1
            log("MISMATCH seed %d index %d FUZZ_KINDS=%d" % (seed, case[0], kinds), case[1:], differing)
    return cases, bad, default_walk


if __name__ == "__main__":
    cases, bad, default_walk = run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                                   kinds=int(sys.argv[3]) if len(sys.argv) > 3 else 8, only=sys.argv[4] if len(sys.argv) > 4 else None)
    print("cases %d (%d with the default walk), mismatches %d" % (cases, default_walk, bad))
    sys.exit(1 if bad else 0)
