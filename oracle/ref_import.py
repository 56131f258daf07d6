"""TEST INFRASTRUCTURE ONLY - imports the real reference (ibivu/PRALINE) in THIS container.

The reference package is imported from where it lies (/root/reference, read-only); its one
native module, praline.util.cext, is the file oracle/build_ref.sh compiled from the reference's
own praline/util/cext.c.  Used only by tests/golden/make_golden.py (fixture generation) and by
CPU tests that cross-check the oracle; never by the product path, never on the GPU box
(/root/reference does not exist there).
"""
import importlib.machinery
import importlib.util
import glob
import os
import sys

REF_ROOT = os.environ.get("PRALINE_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def load_ref_cext():
    """Load oracle/_ref/cext*.so (the reference's cext.c compiled as-is) as a module."""
    cands = sorted(glob.glob(os.path.join(_HERE, "_ref", "cext*.so")))
    if not cands:
        raise ImportError("oracle/_ref/cext*.so missing - run oracle/build_ref.sh")
    loader = importlib.machinery.ExtensionFileLoader("cext", cands[0])
    spec = importlib.util.spec_from_file_location("cext", cands[0], loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "praline")) and bool(
        glob.glob(os.path.join(_HERE, "_ref", "cext*.so")))


def import_reference():
    """Import the reference `praline` package with its C extension bound to oracle/_ref."""
    if "praline" in sys.modules and getattr(sys.modules["praline"], "__file__", "").startswith(REF_ROOT):
        return sys.modules["praline"]
    if not available():
        raise ImportError("reference not available in this environment")
    cext = load_ref_cext()
    sys.modules["praline.util.cext"] = cext
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    sys.dont_write_bytecode = True  # /root/reference is read-only
    import praline  # noqa: E402
    import praline.component  # noqa: F401,E402
    return praline
