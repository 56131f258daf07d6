import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = open(os.path.join(ROOT, "scripts", "exp_ablate.py")).read().split("code = r'''")[1].split("''' % ROOT")[0] % ROOT
for N in (256, 512):
    for G in ("0", "4", "8", "16", "32"):
        env = dict(os.environ, N=str(N), PRALINE_XCD_GROUP=G)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("G=%s" % G, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
