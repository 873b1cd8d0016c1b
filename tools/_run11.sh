cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kat.py -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-200
python - <<'PY'
import sys, ctypes as C, numpy as np, glob, os
sys.path.insert(0, "tests")
import kat_io, raytracer_amd as ra
lib = ra.rtgpu_lib(); c = C.c_void_p(); lib.rtgpu_create(0, C.byref(c))
for p in sorted(glob.glob("tests/golden/*.kat")):
    name = os.path.basename(p)
    if name.startswith("host_"): continue
    func, inputs, expected = kat_io.load_kat(name)
    out = np.zeros_like(expected)
    lib.rtgpu_kat(c, C.c_uint32(func), inputs.ctypes.data_as(C.c_void_p), C.c_uint32(inputs.shape[1]), out.ctypes.data_as(C.c_void_p), C.c_uint32(expected.shape[1]), C.c_uint32(len(inputs)))
    bad = kat_io.bit_mismatch(expected, out)
    if bad.any():
        signed_zero = bad & (expected == 0.0) & (out == 0.0)
        other = bad & ~signed_zero
        print(name, "mismatches", int(bad.sum()), "of which +-0:", int(signed_zero.sum()), "cols", np.unique(np.nonzero(signed_zero)[1]), "| other:", int(other.sum()), "cols", np.unique(np.nonzero(other)[1]))
PY
