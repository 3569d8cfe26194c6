"""Import the read-only Python reference (/root/reference) in THIS container only.

Test infrastructure, never shipped, never used on the GPU box (the reference does not
travel).  The reference needs `numba` (not installed, no network); its @jit/@njit kernels are
plain Python underneath, so we import it with identity-decorator stand-ins created in a temp
dir at run time: what executes is the reference's own un-jitted NumPy/Python arithmetic
(IEEE double, no fastmath).  h5py/corner/pymultinest/... are optional third-party modules the
hot path never calls; they are stubbed as empty modules so `import archnemesis` succeeds.

Used only by oracle/gen_golden.py (fixture generator) and tests marked `needs_reference`.
"""
import os, sys, tempfile, textwrap, importlib

REFERENCE_ROOT = os.environ.get("ANSFM_REFERENCE_ROOT", "/root/reference")

_NUMBA_INIT = textwrap.dedent('''
    def _identity_factory(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        def deco(f):
            return f
        return deco
    jit = njit = vectorize = guvectorize = generated_jit = _identity_factory
    prange = range
    float64 = float; int64 = int; int32 = int; boolean = bool
    class _Types:
        float64 = float; int64 = int
        class WrapperAddressProtocol: pass
        def __getattr__(self, k): return object
    types = _Types()
    def typeof(x): return type(x)
    class config: DISABLE_JIT = True
''')

_NUMBA_EXT = textwrap.dedent('''
    def get_cython_function_address(*a, **k): return 0
    def overload(*a, **k):
        def deco(f): return f
        return deco
    register_jitable = lambda f=None, **k: (f if f is not None else (lambda g: g))
''')

_NUMBA_TYPES = textwrap.dedent('''
    class WrapperAddressProtocol: pass
    class _Sig:
        def __call__(self, *a, **k): return self
        def __getitem__(self, k): return self
    float64 = int64 = int32 = intc = double = boolean = _Sig()
    def FunctionType(*a, **k): return _Sig()
    def __getattr__(k): return _Sig()
''')

_GENERIC_STUB = textwrap.dedent('''
    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        class _Missing:
            def __init__(self,*a,**k): raise ImportError("stub module: optional dependency absent")
        return _Missing
''')


def make_shims(dirpath):
    os.makedirs(os.path.join(dirpath, "numba"), exist_ok=True)
    with open(os.path.join(dirpath, "numba", "__init__.py"), "w") as f:
        f.write(_NUMBA_INIT)
    with open(os.path.join(dirpath, "numba", "extending.py"), "w") as f:
        f.write(_NUMBA_EXT)
    with open(os.path.join(dirpath, "numba", "types.py"), "w") as f:
        f.write(_NUMBA_TYPES)
    for name in ("h5py", "corner", "pymultinest", "hapi", "bs4", "cdsapi", "pygrib", "netCDF4", "mpi4py"):
        try:
            importlib.import_module(name)
            continue
        except Exception:
            pass
        os.makedirs(os.path.join(dirpath, name), exist_ok=True)
        with open(os.path.join(dirpath, name, "__init__.py"), "w") as f:
            f.write(_GENERIC_STUB)


def import_reference():
    """Returns the imported `archnemesis` package (reference), or raises ImportError."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "archnemesis")):
        raise ImportError("reference tree not present (expected on the build container only)")
    shim_dir = tempfile.mkdtemp(prefix="ansfm_refshim_")
    make_shims(shim_dir)
    sys.path.insert(0, shim_dir)
    sys.path.insert(1, REFERENCE_ROOT)
    import archnemesis  # noqa
    # LBL: the numba cython-address binding is meaningless un-jitted; use scipy's voigt directly.
    try:
        import scipy.special
        import archnemesis.lineshape.voigt_impl.voigt_scipy as vs
        vs.voigt_profile = scipy.special.voigt_profile
    except Exception:
        pass
    return archnemesis


if __name__ == "__main__":
    ans = import_reference()
    print("imported reference archnemesis from", ans.__file__)
