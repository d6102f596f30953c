for a in 0 1 3; do
  BIOSCAN_V2_ABLATE=$a python - <<'PY'
import os,sys,json,subprocess
sys.path.insert(0,'.')
import __graft_entry__ as ge
pkg=ge._load_pkg()
path='/dev/shm/ab.bam'
if not os.path.exists(path):
    subprocess.check_output(['tools/_build/synth_bam',path,'65536','42','16'])
prov=pkg.BamTableProvider(path,index_path="")
prov.make_resident()
lib=pkg.load_library()
import ctypes as C
# time inflate only via bgzf-level stats: use execute_device on count projection (decode is forced each call)
plan=prov.scan(projection=[])
ms=[]
for i in range(3):
    try:
        st=plan.execute_device(0)
        ms.append(st['ms_inflate'])
    except Exception as e:
        ms.append(str(e)[:60])
print('ablate',os.environ.get('BIOSCAN_V2_ABLATE'),ms)
PY
done
