cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY'
import subprocess, csv, glob, collections, os, shutil
def blits(cmd):
    shutil.rmtree('/tmp/kt_b', ignore_errors=True)
    subprocess.run(['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', '/tmp/kt_b', '-o', 'r', '--'] + cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    f = glob.glob('/tmp/kt_b/*kernel_trace.csv')[0]
    return collections.Counter(r['Kernel_Name'][:40] for r in csv.DictReader(open(f)) if 'rocclr' in r['Kernel_Name'])
for name, base in (('medformer', ['python', 'tools/medformer_step.py']),):
    a = blits(base + ['10', 'bf16', 'graph']); b = blits(base + ['30', 'bf16', 'graph'])
    print(name, 'per replay:', {k: (b[k] - a[k]) / 20 for k in set(a) | set(b)})
PY
