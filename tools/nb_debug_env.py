"""Developer environment switches -> the library's debug setters (include/neube_hip_debug.h).

Until round 5 the conv launchers of libneube_hip.so read NB_* environment variables themselves; since round 6 the shipped library
reads NO environment variable (no global mutable state on the product path), and the A/B scripts under tools/ -- which
switch a kernel form per run with `NB_UP1_PP=0 python bench.py ...` -- get the same effect from here: bench.py and the tools call
``apply()`` right after loading the library.  Nothing in brushstroke_engine_amd/ imports this file (tests/test_abi.py polices that
the package never touches an nb_debug_ symbol)."""
import ctypes
import os

# variable -> (setter, value when the variable is unset = the setter's "automatic")
SWITCHES = {
    "NB_DEBUG": ("nb_debug_set_flags", 0),
    "NB_STAGGER": ("nb_debug_set_stagger", 0),
    "NB_UP1_SMALL": ("nb_debug_set_up1_small", -1),
    "NB_UP1_V2": ("nb_debug_set_up1_v2", -1),
    "NB_UP1_PP": ("nb_debug_set_up1_pp", -1),
    "NB_UP1_ROWS": ("nb_debug_set_up1_rows", 0),
    "NB_UP2_TQH": ("nb_debug_set_up2_tile", 0),
    "NB_UP2_PAIR": ("nb_debug_set_up2_pair", -1),
    "NB_UP2_V2": ("nb_debug_set_up2_v2", -1),
    "NB_UP2V_PERSIST": ("nb_debug_set_up2v_persistent", -1),
    "NB_UP1_PERSIST": ("nb_debug_set_up1_persistent", -1),
    "NB_PERSIST_WGS": ("nb_debug_set_persistent_wgs_per_cu", 0),
    "NB_SMALL_WAVES": ("nb_debug_set_small_waves", 0),
    "NB_SMALL_BLOCKS": ("nb_debug_set_small_blocks", 0),
    "NB_ENC_SMALL": ("nb_debug_set_enc_small", -1),
    "NB_WGRAD_WGS": ("nb_debug_set_wgrad_wgs", 0),
    "NB_UPFIRDN_GENERIC": ("nb_debug_set_upfirdn_generic", 0),
}


def apply(lib=None, verbose=False):
    """Set every switch whose variable is present in the environment; returns {variable: value} of what was applied."""
    if lib is None:
        from brushstroke_engine_amd import _lib
        lib = _lib.lib()
    done = {}
    for var, (setter, _) in SWITCHES.items():
        v = os.environ.get(var)
        if v is None or v == "":
            continue
        fn = getattr(lib, setter)
        fn.restype, fn.argtypes = None, [ctypes.c_int]
        fn(int(v))
        done[var] = int(v)
    if verbose and done:
        print(f"[nb_debug_env] {done}", flush=True)
    return done
