set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
{
for v in "" wdbg2; do
for ctr in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pw
  (cd /tmp && CLSLAM_TOOL_LIB=$v BENCH_WGRAD=0 BENCH_LAYERS=0,3 rocprofv3 --pmc $ctr -d /tmp/pw -o run -- python $OLDPWD/tools/bench_conv.py 10 40 > /tmp/pw.log 2>&1)
  echo "== variant '$v' $ctr  (B=10: layer1 64ch@48x160 output 19.66 MB, input 19.66 MB, U 0.26 MB; layer4 512ch@6x20 output 2.46 MB, input 2.46 MB, U 16.8 MB)"
  python tools/pmc_summary.py $(ls /tmp/pw/*/*.db /tmp/pw/*.db 2>/dev/null | head -1) wino8 2>&1 | tail -2
  python - <<'PY'
import sqlite3, glob
db = (glob.glob('/tmp/pw/*/*.db') + glob.glob('/tmp/pw/*.db'))[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute('pragma table_info(counters_collection)').fetchall()]
ix = {n: i for i, n in enumerate(cols)}
vals = [r[ix['value']] for r in c.execute('select * from counters_collection') if 'wino8' in (r[ix['kernel_name']] if 'kernel_name' in ix else r[ix['name']])]
print('   per launch (KB), in launch order:', [round(v) for v in vals][:60])
PY
done; done
} > $O/c12_wino_pmc.txt 2>&1
echo done
