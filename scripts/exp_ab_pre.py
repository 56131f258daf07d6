"""A/B of two library builds on the C3 preprofile stage (run each in its own process: PRALINE_LIB is read at import)."""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default": env["PRALINE_LIB"] = os.path.join(ROOT, lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "exp_preprofile_c3.py")], env=env, capture_output=True, text=True).stdout
        print(lib, "|", " | ".join(l.split("): ")[1] + " " + l.split()[2] for l in out.strip().splitlines()[2:]), flush=True)
