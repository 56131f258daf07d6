import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = open(os.path.join(ROOT, "scripts", "exp_ablate.py")).read().split("code = r'''")[1].split("''' % ROOT")[0] % ROOT
for N in (256, 512):
    for wpb in (1, 2, 4):
        for e in (0, 7):
            env = dict(os.environ, PRALINE_EXP=str(e), N=str(N), PRALINE_WPB=str(wpb))
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            print("wpb=%d" % wpb, (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
