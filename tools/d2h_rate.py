import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
pkg = load_pkg()
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")])
meta = json.loads(subprocess.check_output([os.path.join(ROOT, "tools/_build/synth_bam"), "/dev/shm/d2h.bam", "65536", "42", "16"]).decode())
prov = pkg.BamTableProvider("/dev/shm/d2h.bam", None, True, None, index_path="")
plan = prov.scan()
for rep in range(3):
    t0 = time.perf_counter()
    st = plan.execute_device(0, 8192)
    t1 = time.perf_counter()
    rows = 0; nbytes = 0
    for b in plan.execute(0, 8192):
        rows += b.num_rows; nbytes += b.nbytes
    t2 = time.perf_counter()
    print(f"rep {rep}: device-only {1e3*(t1-t0):.1f} ms; with D2H + Arrow import of {rows} rows / {nbytes/1e9:.2f} GB: {1e3*(t2-t1):.1f} ms -> {rows/(t2-t1)/1e6:.1f} Mrec/s, D2H-inclusive")
os.unlink("/dev/shm/d2h.bam")
