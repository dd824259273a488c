import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_full_size.py'), '-x', '-q'], capture_output=True, text=True, cwd=ROOT)
print(r.stdout.strip().splitlines()[-1])
