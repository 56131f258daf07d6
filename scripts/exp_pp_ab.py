"""A/B of library variants (PRALINE_LIB) on build_preprofiles, all of C3.  usage: exp_pp_ab.py [variant.so ...]"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for lib in (sys.argv[1:] or [""]):
        env = dict(os.environ)
        if lib:
            env["PRALINE_LIB"] = os.path.join(ROOT, lib)
        print("==", lib or "default", flush=True)
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "exp_pp_rate.py")], env=env, check=False)
