"""Debug helper: run each padded shape in its own process so an abort pinpoints the shape."""
import subprocess
import sys

SHAPES = [(100, 77), (129, 130), (200, 256)]

CODE = r'''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import dcdf_amd, oracle_lib as O
rows, cols, which = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(rows * 1000 + cols)
a = rng.integers(-9, 9, size=(5, rows, cols)).astype(np.int64)
a[2] = a[1]; a[3] = a[1] + 4; a[4, : rows // 2] = a[1, : rows // 2]
b = rng.integers(-40000, 40000, size=(4, rows, cols)).astype(np.int32)
b[2, :, : max(1, cols // 2)] = b[0, :, : max(1, cols // 2)] - 3
x = a if which == "a" else b
r = dcdf_amd.build_batch([x])[0]
if isinstance(r, Exception):
    print("ERR", r); sys.exit(3)
ref = O.chunk_build(x)
print("same" if r.data.write_to() == ref else "DIFF", len(ref))
'''
for rows, cols in SHAPES:
    for which in "ab":
        p = subprocess.run([sys.executable, "-c", CODE, str(rows), str(cols), which], capture_output=True, text=True)
        tail = (p.stderr or "").strip().splitlines()[-3:]
        print(rows, cols, which, "rc=%d" % p.returncode, p.stdout.strip(), "|", " / ".join(t[:160] for t in tail) if p.returncode else "", flush=True)
