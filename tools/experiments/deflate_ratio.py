"""dev experiment: compression ratio, block types and kernel time of W2 (k_bgzf_deflate) on BAM payload, against zlib -6"""
import os, subprocess, sys, zlib, struct, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg(); pkg.load_library()
synth = os.path.join(ROOT, "tools", "_build", "synth_bam")
if not os.path.exists(synth): subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
p = "/dev/shm/ratio_%d.bam" % os.getpid()
subprocess.check_output([synth, p, "4096", "42", "16"])
src = open(p, "rb").read(); os.unlink(p)
for name, data in (("synthetic config-2 members", src), ("multi_chrom_large.bam", open(os.path.join(ROOT, "tests/golden/multi_chrom_large.bam"), "rb").read())):
    payload, _ = pkg.bgzf_inflate(data)
    out, ms = pkg.bgzf_deflate(payload)
    types = {0: 0, 1: 0, 2: 0}
    o = 0
    while o < len(out):
        bs = struct.unpack_from("<H", out, o + 16)[0] + 1
        if bs > 28: types[(out[o + 18] >> 1) & 3] += 1
        o += bs
    back, _ = pkg.bgzf_inflate(out)
    assert back == payload
    z = sum(len(zlib.compress(payload[i:i + 65280], 6)) - 6 + 26 for i in range(0, len(payload), 65280))
    print(json.dumps({"input": name, "payload_MB": round(len(payload) / 1e6, 1), "source_ratio": round(len(data) / len(payload), 4),
                      "w2_ratio": round(len(out) / len(payload), 4), "zlib6_ratio": round(z / len(payload), 4), "block_types": types,
                      "kernel_ms": round(ms, 2), "GB_per_s": round(len(payload) / ms / 1e6, 2)}))
