import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = open(os.path.join(ROOT, "scripts", "exp_ablate.py")).read().split("code = r'''")[1].split("''' % ROOT")[0] % ROOT
for N in (256, 512):
    for noxcd in ("1", "0"):
        env = dict(os.environ, N=str(N), PRALINE_NO_XCD=noxcd)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("NO_XCD=%s" % noxcd, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
