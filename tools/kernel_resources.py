#!/usr/bin/env python3
"""Per-kernel VGPR / SGPR / scratch / LDS table from hipcc's assembly listing:  python tools/kernel_resources.py [file.s]
(without an argument: compiles the device translation units of raytracer_amd/csrc for gfx950 to /tmp first; UNITS=rt_trace,rt_tail restricts them)"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/rt_trace_resources.s"
s = ""
if len(sys.argv) <= 1:
    # the library's device translation units, with the flags __graft_entry__.build() gives them
    no_sink = ["-mllvm", "-simplifycfg-sink-common=false"]
    units = [("rt_trace.hip", []), ("rt_shade.hip", no_sink), ("rt_tail.hip", no_sink)]
    if os.environ.get("UNITS"):
        units = [(u, e) for u, e in units if u.split(".")[0] in os.environ["UNITS"].split(",")]
    for unit, extra in units:
        if not os.path.exists(os.path.join(ROOT, "raytracer_amd/csrc", unit)):
            continue
        out = "/tmp/%s_resources.s" % unit.split(".")[0]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"] + extra + ["-S", "--cuda-device-only",
                               os.path.join(ROOT, "raytracer_amd/csrc", unit), "-o", out], stderr=subprocess.DEVNULL)
        s += open(out).read()
else:
    s = open(path).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.vgpr_count:\s+(\d+)', s, re.S):
    name, body = m.group(1), m.group(2)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split('(')[0].replace("void ", "")
    g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, body) or [None, "?"])[1]
    print('%-60s vgpr %3s sgpr %3s scratch %4s lds %6s' % (dn[:60], m.group(3), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
