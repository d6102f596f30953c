"""dev experiment: do the HBM-bound stages of one scan overlap the (issue-bound) inflate of another when K1's persistent
grid leaves room on the CUs?  Two threads, each scanning its own 32768-member file N times, against one thread doing 2N
scans.  BIOSCAN_K1_WAVES_PER_CU sets K1's grid (default: what the occupancy API allows, 20 per CU)."""
import os, subprocess, sys, threading, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg(); pkg.load_library()
synth = os.path.join(ROOT, "tools", "_build", "synth_bam")
if not os.path.exists(synth): subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
paths = []
for i in range(2):
    p = f"/dev/shm/ovl_{os.getpid()}_{i}.bam"
    subprocess.check_output([synth, p, "32768", str(42 + i), "16"])
    paths.append(p)
provs = [pkg.BamTableProvider(p) for p in paths]
plans = [pv.scan(projection=None, target_partitions=1) for pv in provs]
N = int(os.environ.get("N", "6"))
def run(k, n):
    for _ in range(n): plans[k].execute_device(0, 8192)
for k in range(2): run(k, 1)
t0 = time.perf_counter(); run(0, N); run(1, N); t_seq = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(k, N)) for k in range(2)]
[t.start() for t in th]; [t.join() for t in th]
t_par = time.perf_counter() - t0
print(json.dumps({"k1_waves_per_cu": os.environ.get("BIOSCAN_K1_WAVES_PER_CU", "default"), "sequential_ms_per_scan": round(1e3 * t_seq / (2 * N), 3),
                  "two_threads_ms_per_scan": round(1e3 * t_par / (2 * N), 3)}))
for p in paths: os.unlink(p)
