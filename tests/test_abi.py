"""The C-ABI library loads without a GPU and exports exactly what include/utmos_hip.h declares."""
import os
import re

import oracle_util as ou


def header_functions():
    text = open(os.path.join(ou.ROOT, "include", "utmos_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(utm_[a-z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    from utmos_amd import _native as nat
    lib = nat.lib()
    names = header_functions()
    assert "utm_run" in names and "utm_comm_init" in names
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/utmos_hip.h but not exported"


def test_ctypes_prototypes_cover_the_header():
    from utmos_amd import _native as nat
    assert set(nat.PROTOTYPES) | {"utm_last_error"} == header_functions()


def test_abi_version_and_struct_sizes():
    import ctypes
    from utmos_amd import _native as nat
    assert nat.lib().utm_abi_version() == 3
    assert ctypes.sizeof(nat.Record) == 64
    assert ctypes.sizeof(nat.Stats) == 8 * 6 + 4 * 4 + 8 * 4 + 4 * 2 + 8 * 6


def test_argument_errors_are_reported_without_a_gpu():
    import ctypes
    from utmos_amd import _native as nat
    lib = nat.lib()
    assert lib.utm_ctx_destroy(None) == 0
    rc = lib.utm_add_chunk(None, 10, None)
    assert rc == -1 and b"NULL" in lib.utm_last_error()
    h = ctypes.c_void_p()
    rc = lib.utm_ctx_create(0, 10, 8, 5, 0, ctypes.byref(h))     # 8 + 5 > 10
    assert rc == -1 and not h.value


def test_host_generator_is_deterministic_and_rows_informative():
    import numpy as np
    from utmos_amd.device import synth_host
    a, af = synth_host(7, 3000, 50)
    b, _ = synth_host(7, 3000, 50)
    assert (a == b).all()
    bits = np.unpackbits(a.view(np.uint8), axis=1, bitorder="little")[:, :3000]
    assert bits.any(axis=0).all()                                    # every variant has a carrier
    lo, _ = synth_host(7, 1000, 50, first_sample=10, n_samp=5, first_var_global=2000)
    full_bits = bits[10:15, 2000:3000]
    lo_bits = np.unpackbits(lo.view(np.uint8), axis=1, bitorder="little")[:, :1000]
    assert (lo_bits == full_bits).all()                              # shards and chunks see the same matrix
    assert af.dtype == np.float32 and (af > 0).all() and (af <= 0.5).all()
