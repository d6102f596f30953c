python - <<'PY'
import subprocess,os,sys,json
sys.path.insert(0,'.')
subprocess.check_output(['tools/_build/synth_bam','/tmp/s2048.bam','2048','7','8'])
for ov in (0,96):
    subprocess.check_call('touch datafusion-bio-formats_amd/csrc/inflate_v2.hip; make -C datafusion-bio-formats_amd/csrc EXTRA="-DV2_OV_BITS=%d" >/dev/null 2>&1'%ov, shell=True)
    env=dict(os.environ, BIOSCAN_DBG_BLOCK='1192')
    r=subprocess.run([sys.executable,'-c','''
import sys; sys.path.insert(0,".")
import __graft_entry__ as ge
pkg=ge._load_pkg()
d=open("/tmp/s2048.bam","rb").read()
try:
    pkg.bgzf_inflate(d)
    print("OK")
except Exception as e: print("ERR",e)
'''],env=env,capture_output=True,text=True)
    open('gpurun_out/dbg_ov%d.txt'%ov,'w').write(r.stdout+r.stderr)
    print(ov, r.stdout[-200:].strip().split("\n")[-1])
PY
