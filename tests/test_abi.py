"""The C-ABI library loads and exports every symbol include/svo_hip.h declares;
the ctypes mirrors have the C struct sizes.  No GPU, no compute calls."""
import ctypes as C
import os
import re
import subprocess
import tempfile

from svo_pro_universal_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "svo_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svoh_[a-z0-9_]+)\s*\(", src)))


def test_header_functions_are_exported_and_listed():
    lib = capi.load()  # raises if the extension is not built: no fallback
    decl = declared_functions()
    assert decl, "no declarations parsed"
    assert sorted(capi.EXPORTS) == decl
    for name in decl:
        assert hasattr(lib, name), name
    assert lib.svoh_abi_version() == 2


def test_struct_sizes_match_c():
    names = [n for n in dir(capi) if n.startswith("svoh_") and isinstance(getattr(capi, n), type)
             and issubclass(getattr(capi, n), C.Structure)]
    assert len(names) >= 7
    prog = '#include <stdio.h>\n#include "svo_hip.h"\nint main(void){\n'
    for n in names:
        prog += 'printf("%s %%zu\\n", sizeof(%s));\n' % (n, n)
    prog += "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"),
                               os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    sizes = dict(zip(out[::2], map(int, out[1::2])))
    for n in names:
        assert C.sizeof(getattr(capi, n)) == sizes[n], (n, C.sizeof(getattr(capi, n)), sizes[n])


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU svoh_create must fail with SVOH_ERR_NO_DEVICE (-6) or a HIP
    error; with a GPU it succeeds.  Either way nothing silently runs on the CPU."""
    lib = capi.load()
    h = C.c_void_p()
    rc = lib.svoh_create(0, C.byref(h))
    if rc == 0:
        assert h.value
        lib.svoh_destroy(h)
    else:
        assert rc in (-6, -2)
        assert not h.value
        assert b"CPU" in lib.svoh_last_error_string(None) or rc == -2


def test_lockstep_c_face_is_exported_and_refuses_null():
    """The C faces of the lock-step engines in libsvo_hip_host.so (host/svo_hip_lockstep_c.h: svohl_* mono, svohs_* stereo): every declared
    function is exported, and a call without an engine fails with an error text instead of crashing.  No GPU, no compute."""
    from svo_pro_universal_amd import lockstep as ls
    src = open(os.path.join(ROOT, "svo_pro_universal_amd", "host", "svo_hip_lockstep_c.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decl = sorted(set(re.findall(r"\b(svoh[ls]_[a-z0-9_]+)\s*\(", src)))
    assert len(decl) >= 20 and "svohl_create_streams" in decl and "svohs_run_sequence" in decl
    lib = ls.load_host()
    for name in decl:
        assert hasattr(lib, name), name
    assert lib.svohl_finish(None) != 0 and b"NULL" in lib.svohl_last_error()
    assert lib.svohs_finish(None) != 0 and b"NULL" in lib.svohs_last_error()
    out = C.c_void_p()
    assert lib.svohs_create(None, 4, None, None, None, 8, 0.5, 1, 1, C.byref(out)) != 0 and not out.value
