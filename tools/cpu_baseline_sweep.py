import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, os
sys.path.insert(0, "%s"); sys.path.insert(0, os.path.join("%s", "gan-reverser_amd"))
import bench
t = int(sys.argv[1]); b = int(sys.argv[2])
print(bench.cpu_baseline((1, 32, 32), 32, b, t))
''' % (ROOT, ROOT)
for t in (16, 32, 64, 128, 256):
    for b in (32, 64):
        env = dict(os.environ, OMP_NUM_THREADS=str(t), OMP_PROC_BIND="spread", OMP_PLACES="cores")
        out = subprocess.run([sys.executable, "-c", code, str(t), str(b)], env=env, capture_output=True, text=True)
        print(t, b, out.stdout.strip()[-160:], out.stderr.strip()[-200:] if out.returncode else "")
